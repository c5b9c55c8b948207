#!/bin/bash
set -u
R=$PWD; O=gpurun_out/r5; mkdir -p $O
bash tools/gb.sh shared
CMS_BENCH_WINDOW_STREAMS=1 bash tools/gb.sh ownstreams
bash tools/gb.sh shared2
bash tools/gb.sh grp3 --ba-groups 3
bash tools/gb.sh grp4 --ba-groups 4
CMS_BENCH_PART=ba bash tools/gb.sh baonly
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ba_" > $O/t_ba.log 2>&1; tail -4 $O/t_ba.log
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 > $R/$O/trace_bench.json 2> $R/$O/trace_bench.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); python - "$f" $O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Z"]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $O/trace; gzip -f $O/trace_small.csv
