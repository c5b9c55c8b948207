#!/bin/bash
# round 5, GPU call 4: the device-side planner with its matcher in registers
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_plan or put_from_frame or config4_size_eight or tracked_windows" 2>&1 | tail -5 > $O/gputests4.txt
tail -3 $O/gputests4.txt
for i in 1 2; do
  bash tools/gb.sh r05d_fast$i | cut -c1-330
  CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05d_host$i | cut -c1-330
done
CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05d_fast_cpu | cut -c1-200; grep "window threads\|thread CPU" gpurun_out/gb_r05d_fast_cpu.log | cut -c1-700
CMS_BA_RELAXED_WAIT=1 CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05d_fast_relaxed | cut -c1-200; grep "window threads\|thread CPU" gpurun_out/gb_r05d_fast_relaxed.log | cut -c1-700
CMS_BA_RELAXED_WAIT=1 bash tools/gb.sh r05d_fast_relaxed_wt8 --window-threads 8 | cut -c1-200
bash tools/gb.sh r05d_fast_wt8 --window-threads 8 | cut -c1-200
R=$PWD; OUT=$R/$O/trace_fast2; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 > $OUT/bench.json 2> $OUT/bench.err)
python - $OUT/t_kernel_stats.csv <<'PY' | tee $O/steptrace2.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-34s calls %6s  avg %9.1f us  max %9.1f  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
for r in rows:
    if "expand" in r["Name"] or "unpermute" in r["Name"]:
        print("%-34s calls %6s  avg %9.1f us  max %9.1f  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
