#!/bin/bash
# round 5, GPU call 9: two pyramid levels per launch (bit-exact tests + frame-path timing), Fuse over shared sets, the full mapping side
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -x -q -k "extract or lut or golden or front_camera or fuse or put_from_frame or harness or closed" 2>&1 | tail -4
for i in 1 2; do
  echo "two levels per launch: $(timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)"
  echo "one level per launch:  $(CMS_RESIZE_SINGLE=1 timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)"
done
for i in 1 2; do
  bash tools/gb.sh r05i_full$i | cut -c1-250
  CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05i_min$i | cut -c1-250
done
CMS_RESIZE_SINGLE=1 CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05i_min_rz1 | cut -c1-250
python - <<'PY'
import json
l=[x for x in open('gpurun_out/gb_r05i_full2.log') if x.startswith('{')]
j=json.loads(l[-1]); c=j['config']
ms=c['mapping_side']; print({k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')})
print(c['host'], c['ba_worker_ms'])
PY
