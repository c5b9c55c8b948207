#!/bin/bash
# frame path's queue at low / high dispatch priority against the mapping side (A/B, three repeats each)
set -u
O=gpurun_out/r14; mkdir -p $O
for i in 1 2; do
bash tools/gb.sh base$i
CMS_BENCH_FRAME_PRIORITY=low bash tools/gb.sh flow$i
CMS_BENCH_FRAME_PRIORITY=high bash tools/gb.sh fhigh$i
done
