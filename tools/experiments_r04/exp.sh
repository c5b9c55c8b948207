#!/bin/bash
# Round-4 GPU experiments, one parametrised script (run on the GPU box: gpurun -- 'bash tools/experiments_r04/exp.sh <case> [args]').
# Output goes to gpurun_out/r04/<case>*.  Cases:
#   probe        tools/probe/f64_pipes.hip (do FP64 MFMA and FP64 vector instructions of two wavefronts of a SIMD overlap?)
#   ba16 [libs]  tools/prof_ba_many.py 16 track diff for every kernel of the round (3 Schur, 5 solve, 6 trial, 7 reduce2), per library variant
#   bench [libs] tools/gb.sh (short bench line) per library variant (default | ab_NAME)
#   tests / batests / exttests          pytest -m gpu / the BA parity subset / the extraction subset
#   ba16s [libs]                        ... the Schur kernel only
#   pmcba / pmcwait / pmcifetch [libs]  PMC passes over the local-BA kernels: instruction mix / where a wavefront's cycles go / instruction fetch
#   rmclk                               per-phase cycle stamps of the run-major body (needs tools/ab_build.sh rmclk -DBA_RM_CLK)
#   waves / rmweight / emcost / techunks   sweeps of CMS_BA_SE_WAVES / CMS_BA_RM_WEIGHT (split workgroups) / CMS_BA_EM_COST_A,B / CMS_BA_TE_CHUNKS
#   frames                              the frame path alone: describe walk against list order
#   steptrace <tag> [ENV=..]            bench.py under rocprofv3 --stats: average duration of every kernel inside the step
#   timeline <tag> [ENV=..]             bench.py under rocprofv3 --kernel-trace, per-queue busy time and gaps (tools/timeline.py)
# (profiles/r04_experiments.txt collects the outputs DESIGN.md quotes)
set -u
O=gpurun_out/r04; mkdir -p $O
libpath() { if [ "$1" = "default" ]; then echo $PWD/cubemapslam_amd/lib/libcubemapslam_hip.so; else echo $PWD/cubemapslam_amd/lib/ab_$1.so; fi; }
case "$1" in
  probe)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/f64_pipes.hip -o /tmp/f64p 2>/dev/null && /tmp/f64p | tee $O/probe_f64_pipes.txt ;;
  ba16)
    shift
    for v in ${@:-default}; do
      for k in 3 5 6 7; do
        echo "$v kernel $k: $(CMS_HIP_LIB=$(libpath $v) timeout 300 python tools/prof_ba_many.py 16 track diff $k 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/ba16.txt
      done
    done ;;
  ba16s)   # Schur kernel only
    shift
    for v in ${@:-default}; do
      echo "$v: $(CMS_HIP_LIB=$(libpath $v) timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/ba16.txt
    done ;;
  bench)
    shift
    for v in ${@:-default}; do CMS_HIP_LIB=$(libpath $v) bash tools/gb.sh r04_$v | tee -a $O/bench.txt; done ;;
  pmcba)   # instruction counters of the local-BA kernels, 16 tracked windows per launch, per library variant
    shift
    R=$PWD
    for v in ${@:-default}; do
      OUT=$R/$O/pmc_$v; rm -rf $OUT; mkdir -p $OUT
      i=0
      for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM"; do
        i=$((i+1))
        (cd /tmp && TMPDIR=/tmp CMS_HIP_LIB=$(cd $R && libpath $v) timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p$i -- python $R/tools/prof_ba_many.py 16 track diff > $OUT/p$i.log 2>&1)
      done
      rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
      echo "== $v"; python tools/pmc_mix.py $OUT | grep "schur\|trial\|reduce2" | tee -a $O/pmcba.txt
    done ;;
  rmclk)   # cycle stamps of one wavefront of the run-major body (needs tools/ab_build.sh rmclk -DBA_RM_CLK)
    CMS_HIP_LIB=$(libpath rmclk) timeout 300 python tools/prof_rm_clk.py 16 2>&1 | tail -12 | tee $O/rmclk.txt ;;
  waves)   # the Schur kernel with fewer wavefronts per workgroup: how much of its time is latency?
    for w in 8 6 4 2; do echo "CMS_BA_SE_WAVES=$w: $(CMS_BA_SE_WAVES=$w timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/waves.txt; done ;;
  frames)  # the frame path alone, 256 frames per launch: stage times, list order of k_describe against the spatial walk
    for i in 1 2; do
      echo "walk:       $(CMS_DESC_SPATIAL_ORDER=1 timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)" | tee -a $O/frames.txt
      echo "list order: $(timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)" | tee -a $O/frames.txt
    done ;;
  exttests) # the extraction parity tests only
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "extract or lut or golden or front_camera" 2>&1 | tail -8 | tee $O/exttests.txt ;;
  steptrace)  # bench.py under rocprofv3 --kernel-trace --stats: in-step average duration of every kernel.  usage: steptrace <tag> [ENV=VAL ...]
    shift; TAG=$1; shift
    R=$PWD; OUT=$R/$O/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 > $OUT/bench.json 2> $OUT/bench.err)
    python - $OUT/t_kernel_stats.csv $TAG <<'PY' | tee -a $O/steptrace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("== %s" % sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    print("%-34s calls %6s  avg %9.1f us  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
    tail -c 400 $OUT/bench.json | cut -c1-300; rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv ;;
  pmcwait)  # where a wavefront's cycles go (disjoint: WAIT_ANY parked in s_waitcnt / barrier, WAIT_INST_ANY issue stall, ACTIVE_INST_ANY issuing)
    shift
    R=$PWD
    for v in ${@:-default}; do
      OUT=$R/$O/pmcw_$v; rm -rf $OUT; mkdir -p $OUT
      i=0
      for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
        i=$((i+1))
        (cd /tmp && TMPDIR=/tmp CMS_HIP_LIB=$(cd $R && libpath $v) timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p$i -- python $R/tools/prof_ba_many.py 16 track diff > $OUT/p$i.log 2>&1)
      done
      rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
      echo "== $v"; python tools/pmc_mix.py $OUT | grep "schur\|trial_edges" | tr ' ' '\n' | tee -a $O/pmcwait.txt
    done ;;
  rmweight)  # split of a window's workgroups between the run-major and the edge-major body (CMS_BA_RM_WEIGHT)
    for w in 100 60 50 40 30; do echo "CMS_BA_RM_WEIGHT=$w: $(CMS_BA_RM_WEIGHT=$w timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/rmweight.txt; done ;;
  emcost)  # run chunks and left-over chunks in the same workgroups: cost model of a left-over chunk (a + b x steps), against separate workgroups
    run1() { echo "$1: $(env $1 timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/emcost.txt; }
    run1 "CMS_BA_SPLIT_WORKGROUPS=1"
    for ab in ${EMCOST_SET:-60:45 40:30 80:60 110:80 60:20 100:40 140:100}; do run1 "CMS_BA_EM_COST_A=${ab%:*} CMS_BA_EM_COST_B=${ab#*:}"; done ;;
  timeline)  # kernel trace of a few steps (CSV kept: analysed by tools/timeline.py).  usage: timeline <tag> [ENV=VAL ...]
    shift; TAG=$1; shift
    R=$PWD; OUT=$R/$O/tl_$TAG; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 4 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 > $OUT/bench.json 2> $OUT/bench.err)
    python tools/timeline.py $OUT/t_kernel_trace.csv | tee $O/timeline_$TAG.txt
    rm -f $OUT/*_agent_info.csv; gzip -f $OUT/t_kernel_trace.csv ;;
  techunks)  # chunks per wavefront of the trial kernel
    for w in ${TE_SET:-1 2 3 4 6 8}; do echo "CMS_BA_TE_CHUNKS=$w: $(CMS_BA_TE_CHUNKS=$w timeout 300 python tools/prof_ba_many.py 16 track diff 6 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/techunks.txt; done ;;
  pmcifetch)  # instruction fetch of the local-BA kernels: requests, hits / misses of the shared instruction cache, its busy cycles
    shift
    R=$PWD
    for v in ${@:-default}; do
      OUT=$R/$O/pmci_$v; rm -rf $OUT; mkdir -p $OUT
      i=0
      for grp in "SQ_WAVES SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_INPUT_VALID_READYB SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "InstrFetchLatency"; do
        i=$((i+1))
        (cd /tmp && TMPDIR=/tmp CMS_HIP_LIB=$(cd $R && libpath $v) timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p$i -- python $R/tools/prof_ba_many.py 16 track diff > $OUT/p$i.log 2>&1)
      done
      rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
      echo "== $v"; python tools/pmc_mix.py $OUT | grep "schur\|trial_edges\|solve" | tr ' ' '\n' | tee -a $O/pmcifetch.txt
    done ;;
  batests)  # the BA parity tests only
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ba_" 2>&1 | tail -8 | tee $O/batests.txt ;;
  tests)
    timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt ;;
  *) echo "unknown case $1"; exit 2 ;;
esac
