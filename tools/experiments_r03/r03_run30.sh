#!/bin/bash
# the window's last Schur workgroup solves the reduced system in place (CMS_BA_FUSED_SOLVE=1): parity, alone, in the step
set -u
CMS_BA_FUSED_SOLVE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ba_ and not alternative" 2>&1 | tail -3
for n in 1 16; do
echo "n=$n fused solve: $(CMS_BA_FUSED_SOLVE=1 python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
echo "n=$n separate   : $(python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
done
for i in 1 2 3; do
CMS_BA_FUSED_SOLVE=1 bash tools/gb.sh fsolve$i
bash tools/gb.sh base$i
done
