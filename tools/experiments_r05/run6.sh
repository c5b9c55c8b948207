#!/bin/bash
# round 5, GPU call 6: full GPU suite; the step with LocalMapping's whole sequence; host waits
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/gputests6.txt; tail -3 $O/gputests6.txt
for i in 1 2; do
  bash tools/gb.sh r05f_full$i | cut -c1-250
  CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05f_min$i | cut -c1-250
done
CMS_BA_SET_STREAM_WAIT=1 CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05f_min_hostwait | cut -c1-250
CMS_BA_RELAXED_WAIT=1 bash tools/gb.sh r05f_full_relaxed | cut -c1-250
bash tools/gb.sh r05f_full_wt8 --window-threads 8 | cut -c1-250
CMS_BA_RELAXED_WAIT=1 bash tools/gb.sh r05f_full_relaxed_wt8 --window-threads 8 | cut -c1-250
CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05f_full_cpu | cut -c1-120; grep "window threads\|thread CPU" gpurun_out/gb_r05f_full_cpu.log | cut -c1-400
timeout 600 python bench.py > $O/bench6.json 2> $O/bench6.err; tail -c 1500 $O/bench6.json | head -c 600; tail -3 $O/bench6.err
