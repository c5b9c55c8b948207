#!/bin/bash
# Developer aid: bench.py at several step sizes on one box -> gpurun_out/batch_sweep.jsonl (one bench line per --batch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/batch_sweep.jsonl
for b in 64 128 256; do
  timeout 250 python $R/bench.py --cpu-frames 0 --steps 30 --warmup 5 --batch $b 2>/dev/null | tail -1 >> $R/gpurun_out/batch_sweep.jsonl
done
