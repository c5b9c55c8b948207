#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "put_from_frame or fuse" 2>&1 | tail -2
for i in 1 2; do
  bash tools/gb.sh r05k_full$i | cut -c1-250
  CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh r05k_full_maphi$i | cut -c1-250
done
CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05k_min | cut -c1-250
CMS_BENCH_MAP_PRIORITY=high CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05k_min_maphi | cut -c1-250
python - <<'PY'
import json
for t in ("full2", "full_maphi2"):
    l=[x for x in open('gpurun_out/gb_r05k_%s.log' % t) if x.startswith('{')]
    j=json.loads(l[-1]); c=j['config']
    ms=c['mapping_side']; print(t, {k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')}, c['ba_worker_ms'])
PY
