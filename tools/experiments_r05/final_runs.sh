#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
bash tools/experiments_r05/exp.sh tests
for i in 1 2 3; do
  timeout 900 python bench.py > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  python - $O/bench_default_$i.json <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
j=json.loads(l[-1]); c=j['config']
print("value %.1f  step %.3f ms  roofline %s frac %.4f (%.1f us)  minimal %s  unpipelined %s  mapping_only %s  extract %s / in-step %s  det %s  host cores %s  one BA call %s  cpu %s" % (
  j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], 1e3*j['roofline']['ms_per_launch'], c['mapping_side'].get('minimal',{}).get('value'), c['unpipelined']['value'],
  c['mapping_only']['ms_per_step'], c['extract_only']['extractor_vs_survey_bytes']['frac_of_8TBps'], c['extractor_vs_survey_bytes']['frac_of_8TBps'], (c.get('deterministic') or {}).get('value'),
  c['host']['host_cores_used'], c['one_local_ba_call']['ms'], j['cpu_baseline']['value']))
PY
done
