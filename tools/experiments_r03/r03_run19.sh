#!/bin/bash
# trial kernel with its loads in one round trip, issued before the pose staging (A/B against the previous build)
set -u
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ba_ and not alternative" 2>&1 | tail -2
for n in 1 16; do
for k in 6 5 7; do
echo "n=$n new: $(python tools/prof_ba_many.py $n track diff $k 2>&1 | grep lock-step)"
echo "n=$n old: $(CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_te_old.so python tools/prof_ba_many.py $n track diff $k 2>&1 | grep lock-step)"
done
done
echo "random new: $(python tools/prof_ba_many.py 16 random diff 6 2>&1 | grep lock-step)"
echo "random old: $(CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_te_old.so python tools/prof_ba_many.py 16 random diff 6 2>&1 | grep lock-step)"
for i in 1 2; do
bash tools/gb.sh new$i
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_te_old.so bash tools/gb.sh old$i
done
