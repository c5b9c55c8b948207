#!/bin/bash
# slice sum inside the solve kernel against a launch of its own, by group size
set -u
O=gpurun_out/r16; mkdir -p $O
for n in 1 2 4 8 16; do
for m in 0 1000; do
echo "n=$n CMS_BA_SOLVE_REDUCE_MAX=$m: $(CMS_BA_SOLVE_REDUCE_MAX=$m python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
done
done
