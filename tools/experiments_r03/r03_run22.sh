#!/bin/bash
set -u
for i in 1 2; do
bash tools/gb.sh base_$i
CMS_BENCH_SPLIT_TRI_STREAM=1 bash tools/gb.sh split_$i
CMS_BENCH_SPLIT_TRI_STREAM=1 CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh splithigh_$i
done
