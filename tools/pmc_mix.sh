#!/bin/bash
# Developer aid: SQ instruction-mix counters of the frame-path kernels (one --pmc pass per group, kernel trace only).
#   gpurun -- 'bash tools/pmc_mix.sh'   ->  gpurun_out/pmc/*.csv, summary printed by tools/pmc_mix.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p$i -- python $R/tools/prof_frames.py ${PROF_B:-256} 550 2 > $OUT/p$i.log 2>&1
done
python $R/tools/pmc_mix.py $OUT
