#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  bash tools/gb.sh r05l_full$i | cut -c1-250
  CMS_BENCH_SWITCH_INTERVAL_US=5000 bash tools/gb.sh r05l_full_sw5ms$i | cut -c1-250
done
CMS_BENCH_SWITCH_INTERVAL_US=50 bash tools/gb.sh r05l_full_sw50us | cut -c1-250
CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05l_min | cut -c1-250
CMS_BENCH_MAPPING_MINIMAL=1 CMS_BENCH_SWITCH_INTERVAL_US=5000 bash tools/gb.sh r05l_min_sw5ms | cut -c1-250
python - <<'PY'
import json
for t in ("full2", "full_sw5ms2"):
    l=[x for x in open('gpurun_out/gb_r05l_%s.log' % t) if x.startswith('{')]
    j=json.loads(l[-1]); c=j['config']
    ms=c['mapping_side']; print(t, {k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')}, c['ba_worker_ms'], c['host'])
PY
