#!/bin/bash
# Round 3, final GPU pass: the whole -m gpu suite, the default bench line, every profile under profiles/r03_* (tools/run_profiles.sh).
set -u
R=$PWD; O=gpurun_out/final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/t_gpu.log 2>&1; tail -5 $O/t_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json
timeout 300 python tools/driver_call_times.py 40 > $O/r03_driver_call_times.txt 2>&1; tail -12 $O/r03_driver_call_times.txt
CMS_BA_CREATE_TIMING=1 python tools/prof_ba_many.py 4 track diff 2>&1 | grep 'cms_ba_create\]\|cms_ba_create:' | tail -3 > $O/create_timing.txt; cat $O/create_timing.txt
CMS_BENCH_THREAD_CPU=1 CMS_BENCH_STEP_TIMES=1 python bench.py --steps 150 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2> $O/host_cpu.txt > /dev/null; grep 'window threads\|thread CPU' $O/host_cpu.txt | cut -c1-160
timeout 1500 bash tools/run_profiles.sh r03 > $O/run_profiles.log 2>&1; tail -5 $O/run_profiles.log
ls gpurun_out/prof | head -50
