#!/bin/bash
set -u
O=gpurun_out/r9; mkdir -p $O
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 16 2>&1 | tail -11
echo "== BA tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tracked_windows or signature_runs or group_of_sixteen or repeatable or ragged" > $O/t_new.log 2>&1; tail -5 $O/t_new.log
for wgt in 40 60 80 100; do CMS_BA_RM_WEIGHT=$wgt python tools/prof_ba_many.py 16 track diff > $O/ba16_mfma_w$wgt.log 2>&1; echo "mfma weight $wgt: $(tail -2 $O/ba16_mfma_w$wgt.log | head -1)"; done
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 python tools/prof_ba_many.py 16 track diff > $O/ba16_valu.log 2>&1; echo "valu w60: $(tail -2 $O/ba16_valu.log | head -1)"
python tools/prof_ba_many.py 16 random diff > $O/ba16_random.log 2>&1; echo "random: $(tail -2 $O/ba16_random.log | head -1)"
bash tools/gb.sh mfma
bash tools/gb.sh mfma2
