#!/bin/bash
CMS_BENCH_STEP_TIMES=1 python bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | grep "step times" | cut -c1-700
CMS_BENCH_STEP_TIMES=1 python bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | grep "step times" | cut -c1-700
