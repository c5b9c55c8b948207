#!/bin/bash
set -u
R=$PWD; O=gpurun_out/r6; mkdir -p $O
echo "== BA tests (MFMA runs)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tracked_windows or signature_runs or group_of_sixteen" > $O/t_new.log 2>&1; tail -12 $O/t_new.log
echo "== 16 windows alone"
for wgt in 40 60 80 100; do CMS_BA_RM_WEIGHT=$wgt python tools/prof_ba_many.py 16 track diff > $O/ba16_mfma_w$wgt.log 2>&1; echo "mfma weight $wgt: $(tail -2 $O/ba16_mfma_w$wgt.log | head -1)"; done
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 python tools/prof_ba_many.py 16 track diff > $O/ba16_valu.log 2>&1; echo "valu w60: $(tail -2 $O/ba16_valu.log | head -1)"
CMS_BA_NO_RUNS=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_noruns.log 2>&1; echo "noruns: $(tail -2 $O/ba16_noruns.log | head -1)"
echo "== counters (16 tracked windows, mfma)"
( cd /tmp && export TMPDIR=/tmp; i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1)); timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/pmc -o p$i -- python $R/tools/prof_ba_many.py 16 track diff > $R/$O/pmc_p$i.log 2>&1; tail -1 $R/$O/pmc_p$i.log | cut -c1-200
  done )
python tools/pmc_mix.py $O/pmc | grep "kb_ba_lin_schur" > $O/pmc_mix.txt; cat $O/pmc_mix.txt
rm -rf $O/pmc
bash tools/gb.sh mfma
