#!/bin/bash
# round 5, GPU call 8: full mapping side after the batched put kernel / event-free pose write-back; BA_RM_FAST A/B
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fuse or put_from_frame" 2>&1 | tail -3
for i in 1 2; do
  bash tools/gb.sh r05h_full$i | cut -c1-250
  CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05h_min$i | cut -c1-250
done
for v in default rmfast default rmfast; do
  if [ "$v" = "default" ]; then L=$PWD/cubemapslam_amd/lib/libcubemapslam_hip.so; else L=$PWD/cubemapslam_amd/lib/ab_$v.so; fi
  echo "$v: $(CMS_HIP_LIB=$L timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)"
done
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmfast.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "config4_size_eight or tracked_windows or mixed_sizes or stop_flag or repeatable or device_plan" 2>&1 | tail -3
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmfast.so bash tools/gb.sh r05h_full_rmfast | cut -c1-250
timeout 600 python bench.py > $O/bench8.json 2> $O/bench8.err; tail -3 $O/bench8.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05/bench8.json') if x.startswith('{')]
j=json.loads(l[-1]); c=j['config']
print(j['value'], j['ms_per_step'], j['roofline']['ms_per_launch'], j['roofline']['frac'])
ms=c['mapping_side']; print({k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')})
for k in ('unpipelined','mapping_only','ba_window_setup','host','ba_worker_ms','one_local_ba_call','single_stream_closed_loop'):
    print(k, str(c.get(k))[:700])
PY
