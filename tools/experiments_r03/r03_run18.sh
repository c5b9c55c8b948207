#!/bin/bash
# one global copy of the reduced system (FP64 atomics) against range slices: parity, alone, in the step
set -u
O=gpurun_out/r18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ba_ and not alternative" > $O/t_ba.log 2>&1; tail -3 $O/t_ba.log
for n in 1 4 16; do
echo "n=$n global sum: $(python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
echo "n=$n slices    : $(CMS_BA_NO_GLOBAL_SUM=1 python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
done
for i in 1 2 3; do
bash tools/gb.sh gsum$i
CMS_BA_NO_GLOBAL_SUM=1 bash tools/gb.sh slices$i
done
