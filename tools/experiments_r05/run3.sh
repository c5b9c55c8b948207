#!/bin/bash
# round 5, GPU call 3: in-step kernel durations with the device-side planner against the host planner (rocprofv3 --kernel-trace --stats over bench.py)
O=gpurun_out/r05; mkdir -p $O
R=$PWD
for v in fast host; do
  OUT=$R/$O/trace_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = host ]; then export CMS_BA_HOST_PLAN=1; else unset CMS_BA_HOST_PLAN; fi
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 > $OUT/bench.json 2> $OUT/bench.err)
  python - $OUT/t_kernel_stats.csv $v <<'PY' | tee -a $O/steptrace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("== %s" % sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print("%-34s calls %6s  avg %9.1f us  max %9.1f  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  python tools/timeline.py $OUT/t_kernel_trace.csv 2>&1 | tail -25 > $O/timeline_$v.txt
  rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
done
unset CMS_BA_HOST_PLAN
CMS_BENCH_STEP_TRACE=1 bash tools/gb.sh r05c_fast_trace | cut -c1-200
tail -60 gpurun_out/gb_r05c_fast_trace.log | cut -c1-600 > $O/steptrace_host_stamps.txt
