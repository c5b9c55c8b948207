#!/bin/bash
# the next set of windows handed to the pool after the step's frame path (default now) against at the start of the step
set -u
for i in 1 2 3; do
bash tools/gb.sh late_$i
CMS_BENCH_EARLY_SUBMIT=1 bash tools/gb.sh early_$i
done
