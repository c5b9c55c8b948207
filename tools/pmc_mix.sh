#!/bin/bash
# SQ instruction-mix counters (one --pmc pass per group, kernel trace only) of the frame-path kernels (-> gpurun_out/pmc) and of the
# local-BA kernels (-> gpurun_out/pmc_ba); summarised by tools/summarise_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for fam in frames ba; do
  if [ $fam = frames ]; then OUT=$R/gpurun_out/pmc; CMD="python $R/tools/prof_frames.py ${PROF_B:-256} 550 2"; else OUT=$R/gpurun_out/pmc_ba; CMD="python $R/tools/prof_ba_many.py 8"; fi
  mkdir -p $OUT
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p$i -- $CMD > $OUT/p$i.log 2>&1
  done
  rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
done
python $R/tools/pmc_mix.py $R/gpurun_out/pmc
python $R/tools/pmc_mix.py $R/gpurun_out/pmc_ba
