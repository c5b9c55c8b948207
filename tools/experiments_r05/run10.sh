#!/bin/bash
# round 5, GPU call 10: full mapping side with batched write-backs; where the Fuse call's time goes (kernel stats)
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  bash tools/gb.sh r05j_full$i | cut -c1-250
  CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05j_min$i | cut -c1-250
done
python - <<'PY'
import json
l=[x for x in open('gpurun_out/gb_r05j_full2.log') if x.startswith('{')]
j=json.loads(l[-1]); c=j['config']
ms=c['mapping_side']; print({k:v for k,v in ms.items() if k not in ('note','calls_per_key_frame')})
print(c['host'], c['ba_worker_ms'])
PY
R=$PWD; OUT=$R/$O/trace_full; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 > $OUT/bench.json 2> $OUT/bench.err)
python - $OUT/t_kernel_stats.csv <<'PY' | tee $O/steptrace_full.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-34s calls %6s  avg %9.1f us  max %9.1f  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
