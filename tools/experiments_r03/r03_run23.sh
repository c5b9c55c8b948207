#!/bin/bash
set -u
for i in 1 2 3; do
bash tools/gb.sh base_$i
CMS_BENCH_SWITCH_INTERVAL=0.0002 bash tools/gb.sh sw200us_$i
CMS_BENCH_SWITCH_INTERVAL=0.00002 bash tools/gb.sh sw20us_$i
done
