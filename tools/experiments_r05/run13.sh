#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
bash tools/gb.sh r05m_full_g2 | cut -c1-200
bash tools/gb.sh r05m_full_g3 --ba-groups 3 | cut -c1-200
bash tools/gb.sh r05m_full_g4 --ba-groups 4 | cut -c1-200
bash tools/gb.sh r05m_full_g3b --ba-groups 3 | cut -c1-200
bash tools/gb.sh r05m_full_g2b | cut -c1-200
python - <<'PY'
import json
for t in ("full_g2b", "full_g3b"):
    l=[x for x in open('gpurun_out/gb_r05m_%s.log' % t) if x.startswith('{')]
    j=json.loads(l[-1]); c=j['config']
    print(t, c['ba_worker_ms'], c['host']['host_cores_used'], c.get('deterministic'), c['mapping_side'].get('minimal'))
PY
