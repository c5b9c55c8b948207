#!/bin/bash
set -u
for i in 1 2 3; do
CMS_BENCH_WINDOWS_AHEAD=1 bash tools/gb.sh ahead1_$i
CMS_BENCH_WINDOWS_AHEAD=2 bash tools/gb.sh ahead2_$i
done
