#!/bin/bash
set -u
R=$PWD; O=gpurun_out/r4; mkdir -p $O
bash tools/gb.sh runs
CMS_BENCH_PART=ba bash tools/gb.sh baonly
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 > $R/$O/trace_bench.json 2> $R/$O/trace_bench.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); ls -la $f; python - "$f" $O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Workgroup_Size", "Grid_Size", "LDS_Block_Size"]
keep = [k for k in keep if k in rows[0]]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $O/trace; gzip -f $O/trace_small.csv; ls -la $O
