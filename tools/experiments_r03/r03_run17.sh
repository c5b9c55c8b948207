#!/bin/bash
set -u
for i in 1 2 3; do
CMS_BA_SOLVE_REDUCE_MAX=24 bash tools/gb.sh fused$i
CMS_BA_SOLVE_REDUCE_MAX=0 bash tools/gb.sh sep$i
done
