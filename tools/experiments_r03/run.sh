#!/bin/bash
# Round-3 GPU experiments, one parametrised script (was: 36 one-off scripts).  Start from the repository root on the GPU box:
#   gpurun -- 'bash tools/experiments_r03/run.sh <n>'      n = the experiment number of README.md's table
set -u
case "${1:-}" in
1)
# round 3, GPU call 1: the new local-BA path -- parity first, then A/B timings (run through gpurun from the repo root)
O=gpurun_out/r1; mkdir -p $O
echo "== new BA tests" ; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tracked_windows or signature_runs or group_of_sixteen or repeatable" > $O/t_new.log 2>&1; tail -15 $O/t_new.log
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; tail -12 $O/t_all.log
echo "== BA alone, 16 windows"
for v in "track" "random"; do
  python tools/prof_ba_many.py 16 $v diff > $O/ba16_$v.log 2>&1; tail -3 $O/ba16_$v.log
done
CMS_BA_NO_RUNS=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_track_noruns.log 2>&1; tail -2 $O/ba16_track_noruns.log
CMS_BA_SEPARATE_REDUCE=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_track_sepred.log 2>&1; tail -2 $O/ba16_track_sepred.log
CMS_BA_CREATE_TIMING=1 python tools/prof_ba_many.py 2 track > $O/ba_create_timing.log 2>&1; grep cms_ba_create $O/ba_create_timing.log | tail -3
echo "== bench A/B"
bash tools/gb.sh runs
CMS_BA_NO_RUNS=1 bash tools/gb.sh noruns
CMS_BA_SEPARATE_REDUCE=1 bash tools/gb.sh sepred
bash tools/gb.sh random --ba-views random
bash tools/gb.sh runs2
bash tools/gb.sh grp1 --ba-groups 1
bash tools/gb.sh grp4 --ba-groups 4
CMS_BENCH_PART=ba bash tools/gb.sh baonly
echo "== probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomics.hip -o /tmp/lds_atomics 2>/dev/null && /tmp/lds_atomics > $O/probe_lds_atomics.txt 2>&1; tail -5 $O/probe_lds_atomics.txt
echo "== full bench line"
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json; tail -5 $O/bench_full.err
;;
2)
O=gpurun_out/r2; mkdir -p $O
L=$PWD/cubemapslam_amd/lib/ab_r02.so
for args in "312 track" "142 track" "204 track 20 12000 2 0.0" "313 random"; do
  echo "== $args"
  python tools/diag_ba_window.py $args 2>&1 | tail -14
  CMS_BA_NO_RUNS=1 python tools/diag_ba_window.py $args 2>&1 | tail -8
  CMS_BA_RUNS_AS_EDGES=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_BA_SEPARATE_REDUCE=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_BA_DETERMINISTIC=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_HIP_LIB=$L python tools/diag_ba_window.py $args 2>&1 | head -1
done > $O/diag.log 2>&1
cat $O/diag.log
;;
3)
# round 3, GPU call 3: parity with the cascade-aware bar, run-major kernel after the prefetch change (split weights, counters), bench A/B with the thread-local host scratch
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
;;
4)
R=$PWD; O=gpurun_out/r4; mkdir -p $O
bash tools/gb.sh runs
CMS_BENCH_PART=ba bash tools/gb.sh baonly
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 > $R/$O/trace_bench.json 2> $R/$O/trace_bench.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); ls -la $f; python - "$f" $O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Workgroup_Size", "Grid_Size", "LDS_Block_Size"]
keep = [k for k in keep if k in rows[0]]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $O/trace; gzip -f $O/trace_small.csv; ls -la $O
;;
5)
R=$PWD; O=gpurun_out/r5; mkdir -p $O
bash tools/gb.sh shared
CMS_BENCH_WINDOW_STREAMS=1 bash tools/gb.sh ownstreams
bash tools/gb.sh shared2
bash tools/gb.sh grp3 --ba-groups 3
bash tools/gb.sh grp4 --ba-groups 4
CMS_BENCH_PART=ba bash tools/gb.sh baonly
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ba_" > $O/t_ba.log 2>&1; tail -4 $O/t_ba.log
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 > $R/$O/trace_bench.json 2> $R/$O/trace_bench.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); python - "$f" $O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Z"]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $O/trace; gzip -f $O/trace_small.csv
;;
6)
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
;;
7)
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 16 2>&1 | tail -12
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 1 2>&1 | tail -12
;;
8)
O=gpurun_out/r8; mkdir -p $O
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 16 2>&1 | tail -11
echo "== BA tests (MFMA runs)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tracked_windows or signature_runs or group_of_sixteen" > $O/t_new.log 2>&1; tail -5 $O/t_new.log
for wgt in 50 70 100; do CMS_BA_RM_WEIGHT=$wgt python tools/prof_ba_many.py 16 track diff > $O/ba16_mfma_w$wgt.log 2>&1; echo "mfma weight $wgt: $(tail -2 $O/ba16_mfma_w$wgt.log | head -1)"; done
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 python tools/prof_ba_many.py 16 track diff > $O/ba16_valu.log 2>&1; echo "valu w60: $(tail -2 $O/ba16_valu.log | head -1)"
;;
9)
O=gpurun_out/r9; mkdir -p $O
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 16 2>&1 | tail -11
echo "== BA tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tracked_windows or signature_runs or group_of_sixteen or repeatable or ragged" > $O/t_new.log 2>&1; tail -5 $O/t_new.log
for wgt in 40 60 80 100; do CMS_BA_RM_WEIGHT=$wgt python tools/prof_ba_many.py 16 track diff > $O/ba16_mfma_w$wgt.log 2>&1; echo "mfma weight $wgt: $(tail -2 $O/ba16_mfma_w$wgt.log | head -1)"; done
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 python tools/prof_ba_many.py 16 track diff > $O/ba16_valu.log 2>&1; echo "valu w60: $(tail -2 $O/ba16_valu.log | head -1)"
python tools/prof_ba_many.py 16 random diff > $O/ba16_random.log 2>&1; echo "random: $(tail -2 $O/ba16_random.log | head -1)"
bash tools/gb.sh mfma
bash tools/gb.sh mfma2
;;
10)
bash tools/gb.sh base
CMS_BENCH_BA_FIRST=0 bash tools/gb.sh bafirst0
CMS_BENCH_BA_FIRST=300 bash tools/gb.sh bafirst300
CMS_BENCH_BA_FIRST=800 bash tools/gb.sh bafirst800
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 bash tools/gb.sh valu
CMS_BA_NO_RUNS=1 bash tools/gb.sh noruns
bash tools/gb.sh base2
;;
11)
O=gpurun_out/r11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_harness.py -m gpu -q -x > $O/t_harness.log 2>&1; tail -25 $O/t_harness.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "distance_bounds" > $O/t_bounds.log 2>&1; tail -5 $O/t_bounds.log
python - <<'PY' 2>&1 | tail -20
import sys, tempfile, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cubemapslam_amd import harness, synth
camd = synth.camera("lafida", 550)
mask = synth.cubemap_valid_mask(camd)
frames, gts = harness.render_sequence(camd, 40)
d = tempfile.mkdtemp()
harness.export_sequence(d, camd, frames, gts, mask)
rc, recs, out = harness.run_driver(d)
print("rc", rc); print(out[-400:])
tr = [r["ms"] for r in recs if r.get("stage") == "track" and "ba_iterations" not in r and "n_inliers" in r]
kf = [r["ms"] for r in recs if "ba_iterations" in r]
print("driver: tracked median %.3f ms (min %.3f), key frame median %.3f ms, frames %d" % (np.median(tr), min(tr), np.median(kf) if kf else -1, len(recs)))
for r in recs[:8]: print(r)
PY
;;
12)
bash tools/gb.sh sdma
CMS_BA_STAGE_COHERENT=1 bash tools/gb.sh coherent
bash tools/gb.sh sdma2
CMS_BA_STAGE_COHERENT=1 bash tools/gb.sh coherent2
python tools/prof_ba_many.py 16 track diff 2>&1 | grep -E "create|read"
;;
13)
bash tools/gb.sh base
CMS_BENCH_TRI_LAST=1 bash tools/gb.sh trilast
bash tools/gb.sh base2
CMS_BENCH_TRI_LAST=1 bash tools/gb.sh trilast2
;;
14)
# frame path's queue at low / high dispatch priority against the mapping side (A/B, three repeats each)
O=gpurun_out/r14; mkdir -p $O
for i in 1 2; do
bash tools/gb.sh base$i
CMS_BENCH_FRAME_PRIORITY=low bash tools/gb.sh flow$i
CMS_BENCH_FRAME_PRIORITY=high bash tools/gb.sh fhigh$i
done
;;
15)
# hardware queues: GPU_MAX_HW_QUEUES and queue priorities (A/B)
O=gpurun_out/r15; mkdir -p $O
for i in 1 2; do
bash tools/gb.sh base$i
GPU_MAX_HW_QUEUES=8 bash tools/gb.sh q8_$i
GPU_MAX_HW_QUEUES=16 bash tools/gb.sh q16_$i
GPU_MAX_HW_QUEUES=8 CMS_BENCH_FRAME_PRIORITY=low bash tools/gb.sh q8flow$i
CMS_BENCH_FRAME_PRIORITY=low CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh flowmhigh$i
CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh mhigh$i
CMS_BENCH_MAP_PRIORITY=low bash tools/gb.sh mlow$i
done
;;
16)
# slice sum inside the solve kernel against a launch of its own, by group size
O=gpurun_out/r16; mkdir -p $O
for n in 1 2 4 8 16; do
for m in 0 1000; do
echo "n=$n CMS_BA_SOLVE_REDUCE_MAX=$m: $(CMS_BA_SOLVE_REDUCE_MAX=$m python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
done
done
;;
17)
for i in 1 2 3; do
CMS_BA_SOLVE_REDUCE_MAX=24 bash tools/gb.sh fused$i
CMS_BA_SOLVE_REDUCE_MAX=0 bash tools/gb.sh sep$i
done
;;
18)
# one global copy of the reduced system (FP64 atomics) against range slices: parity, alone, in the step
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
;;
20)
for i in 1 2 3; do
bash tools/gb.sh base$i
CMS_BENCH_STAGGER=1 bash tools/gb.sh stagger$i
done
CMS_BENCH_STAGGER=1 bash tools/gb.sh stagger_g4 --ba-groups 4
bash tools/gb.sh base_g4 --ba-groups 4
;;
21)
for i in 1 2 3; do
CMS_BENCH_WINDOWS_AHEAD=1 bash tools/gb.sh ahead1_$i
CMS_BENCH_WINDOWS_AHEAD=2 bash tools/gb.sh ahead2_$i
done
;;
22)
for i in 1 2; do
bash tools/gb.sh base_$i
CMS_BENCH_SPLIT_TRI_STREAM=1 bash tools/gb.sh split_$i
CMS_BENCH_SPLIT_TRI_STREAM=1 CMS_BENCH_MAP_PRIORITY=high bash tools/gb.sh splithigh_$i
done
;;
25)
# k_fast_cells: time up to each phase boundary (CMS_DBG_FAST_STOP: 9 launch only, 1 staging, 2 compass pre-test, 3 refinement, 4 ring score, 0 all)
for st in 9 1 2 3 4 0; do
echo "stop=$st: $(CMS_DBG_FAST_STOP=$st python tools/prof_frames.py 256 550 6 2>&1 | tail -1 | cut -c1-120)"
done
;;
27)
CMS_BENCH_STEP_TIMES=1 python bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | grep "step times" | cut -c1-700
CMS_BENCH_STEP_TIMES=1 python bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | grep "step times" | cut -c1-700
;;
28)
timeout 600 python -m pytest tests -m gpu -q -x -k "pose or harness or driver or extract or smoke or golden" 2>&1 | tail -2
python tools/driver_call_times.py 40 2>&1 | tail -12
python tools/prof_pose.py 64 300 2>&1 | grep -v oracle
;;
30)
# the window's last Schur workgroup solves the reduced system in place (CMS_BA_FUSED_SOLVE=1): parity, alone, in the step
CMS_BA_FUSED_SOLVE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ba_ and not alternative" 2>&1 | tail -3
for n in 1 16; do
echo "n=$n fused solve: $(CMS_BA_FUSED_SOLVE=1 python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
echo "n=$n separate   : $(python tools/prof_ba_many.py $n track diff 2>&1 | grep lock-step)"
done
for i in 1 2 3; do
CMS_BA_FUSED_SOLVE=1 bash tools/gb.sh fsolve$i
bash tools/gb.sh base$i
done
;;
31)
# window groups per step, again, with the frame path at low priority and the global copy of the reduced system
for i in 1 2; do
bash tools/gb.sh g2_$i
bash tools/gb.sh g3_$i --ba-groups 3
bash tools/gb.sh g1_$i --ba-groups 1
done
;;
33)
# host cost of a window against the Schur kernel's time: composition of the left-over chunks on / off / with shorter look-ahead
for v in "" "CMS_BA_NO_PERMUTE=1" "CMS_BA_LOOKAHEAD=12" "CMS_BA_LOOKAHEAD=6"; do
  echo "== ${v:-default}"
  env $v python tools/prof_ba_many.py 16 track diff 2>&1 | grep "lock-step\|cms_ba_create" | tail -2
done
for i in 1 2; do
bash tools/gb.sh base_$i
CMS_BA_NO_PERMUTE=1 bash tools/gb.sh noperm_$i
CMS_BA_LOOKAHEAD=6 bash tools/gb.sh la6_$i
done
;;
34)
# left-over points composed with a look-ahead of 8 (before) against the caller's order (now, when they are a minority): host CPU, step, kernel
for v in "CMS_BA_LEFTOVER_LOOKAHEAD=8" "CMS_BA_LEFTOVER_LOOKAHEAD=0"; do
for i in 1 2; do
echo "== $v"
env $v CMS_BENCH_STEP_TIMES=1 CMS_BENCH_THREAD_CPU=1 python bench.py --steps 150 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | python -c "
import sys, re, json, statistics
txt = sys.stdin.read()
t = [float(x) for x in re.search(r'step times \(ms\): \[(.*?)\]', txt).group(1).split(',')]
print('steps: mean %.2f median %.2f max %.2f, >16 ms: %d of %d' % (statistics.mean(t), statistics.median(t), max(t), sum(1 for x in t if x > 16), len(t)))
for l in txt.splitlines():
    if l.startswith('window threads') or l.startswith('thread CPU'): print(l[:120])
    if l.startswith('{'):
        d = json.loads(l); print('schur us', 1e3 * d['roofline']['ms_per_launch'])
"
done
env $v python tools/prof_ba_many.py 16 track diff 2>&1 | grep "lock-step\|cms_ba_create" | tail -2
done
;;
35)
# measurements / informations / point positions gathered into the internal order on the device (k_ba_gather) against on the host (before)
L=$PWD/cubemapslam_amd/lib
for i in 1 2 3; do
bash tools/gb.sh gather_$i
CMS_HIP_LIB=$L/ab_pre.so bash tools/gb.sh hostgather_$i
done
;;
36)
# the windows' own (creation) streams at low priority like the frame path's, the group streams that carry the Levenberg rounds at normal priority
for i in 1 2 3; do
bash tools/gb.sh prepnormal_$i
CMS_BA_STREAM_PRIORITY=low bash tools/gb.sh preplow_$i
done
;;
37)
# the next set of windows handed to the pool after the step's frame path (default now) against at the start of the step
for i in 1 2 3; do
bash tools/gb.sh late_$i
CMS_BENCH_EARLY_SUBMIT=1 bash tools/gb.sh early_$i
done
;;
38)
# windows of a set started staggered over the step (CMS_BENCH_SPREAD_MS) against all at once
for i in 1 2; do
bash tools/gb.sh burst_$i
CMS_BENCH_SPREAD_MS=6 bash tools/gb.sh spread6_$i
CMS_BENCH_SPREAD_MS=10 bash tools/gb.sh spread10_$i
done
;;
*) echo "usage: $0 <experiment number>; see tools/experiments_r03/README.md"; exit 2 ;;
esac
