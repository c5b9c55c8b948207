#!/bin/bash
set -u
L=$PWD/cubemapslam_amd/lib
for i in 1 2; do
echo "refine>64 : $(python tools/prof_frames.py 256 550 10 2>&1 | tail -1)"
echo "refine>0  : $(CMS_HIP_LIB=$L/ab_refine0.so python tools/prof_frames.py 256 550 10 2>&1 | tail -1)"
echo "refine>128: $(CMS_HIP_LIB=$L/ab_refine128.so python tools/prof_frames.py 256 550 10 2>&1 | tail -1)"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "extract or fast or orb or golden or front" 2>&1 | tail -2
