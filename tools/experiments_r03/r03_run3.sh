#!/bin/bash
# round 3, GPU call 3: parity with the cascade-aware bar, run-major kernel after the prefetch change (split weights, counters), bench A/B with the thread-local host scratch
set -u
R=$PWD
O=gpurun_out/r3; mkdir -p $O
echo "== new BA tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tracked_windows or signature_runs or group_of_sixteen or repeatable" > $O/t_new.log 2>&1; tail -6 $O/t_new.log
echo "== BA alone, 16 windows"
for wgt in 60 100 140; do CMS_BA_RM_WEIGHT=$wgt python tools/prof_ba_many.py 16 track diff > $O/ba16_track_w$wgt.log 2>&1; echo "weight $wgt: $(tail -2 $O/ba16_track_w$wgt.log | head -1)"; done
CMS_BA_NO_RUNS=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_track_noruns.log 2>&1; echo "noruns: $(tail -2 $O/ba16_track_noruns.log | head -1)"
python tools/prof_ba_many.py 16 random diff > $O/ba16_random.log 2>&1; echo "random: $(tail -2 $O/ba16_random.log | head -1)"
python tools/prof_ba_many.py 8 track diff > $O/ba8_track.log 2>&1; echo "8 track: $(tail -2 $O/ba8_track.log | head -1)"
python tools/prof_ba_many.py 1 track > $O/ba1_track.log 2>&1; echo "1 track: $(tail -2 $O/ba1_track.log | head -1)"
echo "== counters (16 tracked windows)"
( cd /tmp && export TMPDIR=/tmp; i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1)); timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/pmc -o p$i -- python $R/tools/prof_ba_many.py 16 track diff > $R/$O/pmc_p$i.log 2>&1
  done
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o ba16 -- python $R/tools/prof_ba_many.py 16 track diff > $R/$O/trace_ba16.log 2>&1 )
rm -f $O/pmc/*_kernel_trace.csv $O/pmc/*_agent_info.csv $O/pmc/*/*_kernel_trace.csv $O/pmc/*/*_agent_info.csv
python tools/pmc_mix.py $O/pmc | grep "kb_ba" > $O/pmc_mix.txt; cat $O/pmc_mix.txt
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {}'
find $O/trace -name "*kernel_trace.csv" -delete
echo "== bench A/B"
bash tools/gb.sh runs
bash tools/gb.sh runs_t16 --window-threads 16
bash tools/gb.sh runs_t64 --window-threads 64
CMS_BA_NO_RUNS=1 bash tools/gb.sh noruns
bash tools/gb.sh random --ba-views random
bash tools/gb.sh grp4 --ba-groups 4
CMS_BENCH_PART=ba bash tools/gb.sh baonly
CMS_BENCH_PART=frames bash tools/gb.sh framesonly
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; tail -8 $O/t_all.log
