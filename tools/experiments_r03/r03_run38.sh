#!/bin/bash
# windows of a set started staggered over the step (CMS_BENCH_SPREAD_MS) against all at once
set -u
for i in 1 2; do
bash tools/gb.sh burst_$i
CMS_BENCH_SPREAD_MS=6 bash tools/gb.sh spread6_$i
CMS_BENCH_SPREAD_MS=10 bash tools/gb.sh spread10_$i
done
