#!/bin/bash
# hardware queues: GPU_MAX_HW_QUEUES and queue priorities (A/B)
set -u
O=gpurun_out/r15; mkdir -p $O
for i in 1 2; do
bash tools/gb.sh base$i
GPU_MAX_HW_QUEUES=8 bash tools/gb.sh q8_$i
GPU_MAX_HW_QUEUES=16 bash tools/gb.sh q16_$i
GPU_MAX_HW_QUEUES=8 CMS_BENCH_FRAME_PRIORITY=low bash tools/gb.sh q8flow$i
CMS_BENCH_FRAME_PRIORITY=low CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh flowmhigh$i
CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh mhigh$i
CMS_BENCH_MAP_PRIORITY=low bash tools/gb.sh mlow$i
done
