#!/bin/bash
# round 5, GPU call 7: the full mapping side after the batched put / faster Fuse call / deferred set-up wait
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -x -q -k "fuse or put_from_frame or config4_size_eight or tracked_windows or stop_flag or mixed_sizes or closed_loop" 2>&1 | tail -3
for i in 1 2; do
  bash tools/gb.sh r05g_full$i | cut -c1-250
  CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05g_min$i | cut -c1-250
done
CMS_BA_SET_STREAM_WAIT=1 CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05g_min_hostwait | cut -c1-250
CMS_BA_RELAXED_WAIT=1 bash tools/gb.sh r05g_full_relaxed | cut -c1-250
bash tools/gb.sh r05g_full_wt8 --window-threads 8 | cut -c1-250
timeout 600 python bench.py > $O/bench7.json 2> $O/bench7.err; tail -3 $O/bench7.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05/bench7.json') if x.startswith('{')]
j=json.loads(l[-1]); c=j['config']
print(j['value'], j['ms_per_step'], j['roofline']['ms_per_launch'], j['roofline']['frac'])
ms=c['mapping_side']; print({k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')})
for k in ('unpipelined','mapping_only','ba_window_setup','host','ba_worker_ms','one_local_ba_call'):
    print(k, c.get(k))
PY
