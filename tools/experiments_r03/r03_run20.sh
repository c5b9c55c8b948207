#!/bin/bash
set -u
for i in 1 2 3; do
bash tools/gb.sh base$i
CMS_BENCH_STAGGER=1 bash tools/gb.sh stagger$i
done
CMS_BENCH_STAGGER=1 bash tools/gb.sh stagger_g4 --ba-groups 4
bash tools/gb.sh base_g4 --ba-groups 4
