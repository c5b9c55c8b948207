#!/bin/bash
set -u
L=$PWD/cubemapslam_amd/lib
for i in 1 2; do
echo "new: $(python tools/prof_pose.py 64 300 2>&1 | grep -v oracle | tr '\n' ' ')"
echo "old: $(CMS_HIP_LIB=$L/ab_pose_old.so python tools/prof_pose.py 64 300 2>&1 | grep -v oracle | tr '\n' ' ')"
done
timeout 600 python -m pytest tests -m gpu -q -x -k "pose or harness or driver or closed" 2>&1 | tail -2
