#!/bin/bash
# Developer aid: time the frame path for several library variants (tools/ab_build.sh), interleaved, R rounds each.
#   tools/ab_run.sh "base f2 l1" [rounds] [B] [F]
R=$(cd "$(dirname "$0")/.." && pwd)
ROUNDS=${2:-3}; B=${3:-32}; F=${4:-550}
python $R/tools/prof_frames.py $B $F 3 > /dev/null 2>&1   # warm the box up
for r in $(seq $ROUNDS); do
  for v in $1; do
    if [ "$v" = "default" ]; then L=$R/cubemapslam_amd/lib/libcubemapslam_hip.so; else L=$R/cubemapslam_amd/lib/ab_$v.so; fi
    echo "$v $(CMS_HIP_LIB=$L python $R/tools/prof_frames.py $B $F 5 | tail -1)"
  done
done
