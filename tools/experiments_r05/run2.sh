#!/bin/bash
# round 5, GPU call 2: why is the step slower with the device-side planner?  (bursts of window uploads?)  + the tests that did not run in call 1
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_plan or put_from_frame or deterministic_mode or pose_direct or reference_masks or alternative_schur" 2>&1 | tail -15 > $O/gputests2.txt
tail -4 $O/gputests2.txt
for i in 1 2; do
  CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05b_host$i | cut -c1-330
  bash tools/gb.sh r05b_fast$i | cut -c1-330
done
CMS_BENCH_SPREAD_MS=6 bash tools/gb.sh r05b_fast_spread6 | cut -c1-330
CMS_BENCH_SPREAD_MS=10 bash tools/gb.sh r05b_fast_spread10 | cut -c1-330
bash tools/gb.sh r05b_fast_wt8 --window-threads 8 | cut -c1-330
bash tools/gb.sh r05b_fast_wt4 --window-threads 4 | cut -c1-330
CMS_BENCH_WINDOWS_AHEAD=1 bash tools/gb.sh r05b_fast_ahead1 | cut -c1-330
CMS_BENCH_WINDOWS_AHEAD=3 bash tools/gb.sh r05b_fast_ahead3 | cut -c1-330
CMS_BA_HOST_PLAN=1 CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05b_host_cpu | cut -c1-330; grep "window threads\|thread CPU" gpurun_out/gb_r05b_host_cpu.log | cut -c1-400
CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05b_fast_cpu | cut -c1-330; grep "window threads\|thread CPU" gpurun_out/gb_r05b_fast_cpu.log | cut -c1-400
