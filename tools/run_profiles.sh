#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the same command bench.py is judged on          -> gpurun_out/prof/<tag>_bench_*
#   2. FETCH_SIZE and WRITE_SIZE, one --pmc pass each: frame-path kernels (tools/prof_frames.py, 256 frames per dispatch) and the
#      local-BA kernels (tools/prof_ba_many.py, 8 windows per dispatch)           -> gpurun_out/prof/<tag>_pmc_{fetch,write}[_ba]_*
#   3. --kernel-trace --stats of CreateNewMapPoints (8 key frames x 20 neighbours) and of 8 local-BA windows in lock-step
#   4. SQ instruction mix of both kernel families (three --pmc passes each)       -> gpurun_out/pmc/, gpurun_out/pmc_ba/
# Summaries are then copied into profiles/ by tools/summarise_profiles.py.  Counter passes never carry --stats / trace domains beyond
# --kernel-trace.
set -u
TAG=${1:-r05}
PB=${PROF_B:-256}          # frames per dispatch of the PMC passes = bench.py's default --batch
export PROF_B=$PB
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 --confined-steps 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# the same trace, condensed to (kernel, start, end, queue, stream): input of tools/step_table.py (per-step chip-time table)
python - "$OUT/${TAG}_bench_kernel_trace.csv" "$OUT/${TAG}_trace_small.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
gzip -f $OUT/${TAG}_trace_small.csv
python $R/tools/step_table.py $OUT/${TAG}_trace_small.csv.gz 6 $OUT/${TAG}_bench_kernel_stats_timed_steps.csv > $OUT/${TAG}_step_table.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_fetch -- python $R/tools/prof_frames.py $PB 550 3 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_write -- python $R/tools/prof_frames.py $PB 550 3 > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_fetch_ba -- python $R/tools/prof_ba_many.py 8 > $OUT/${TAG}_pmc_fetch_ba.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_write_ba -- python $R/tools/prof_ba_many.py 8 > $OUT/${TAG}_pmc_write_ba.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_mapping -- python $R/tools/prof_tri.py 8 20 5 > $OUT/${TAG}_mapping.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_ba8 -- python $R/tools/prof_ba_many.py 8 > $OUT/${TAG}_ba8.log 2>&1
rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv
# probes and stamps behind DESIGN.md's statements: LDS atomic rates, per-phase cycles of the run-major MFMA body, the Python-free driver
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probe/lds_atomics.hip -o /tmp/lds_atomics 2>/dev/null && /tmp/lds_atomics > $OUT/${TAG}_probe_lds_atomics.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics $R/tools/probe/global_atomics.hip -o /tmp/global_atomics 2>/dev/null && /tmp/global_atomics > $OUT/${TAG}_probe_global_atomics.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probe/f64_pipes.hip -o /tmp/f64_pipes 2>/dev/null && /tmp/f64_pipes > $OUT/${TAG}_probe_f64_pipes.txt 2>&1
[ -f $R/cubemapslam_amd/lib/ab_rmclk.so ] && CMS_HIP_LIB=$R/cubemapslam_amd/lib/ab_rmclk.so python $R/tools/prof_rm_clk.py 16 > $OUT/${TAG}_rm_phase_cycles.txt 2>&1
python $R/tools/prof_ba_many.py 16 track diff > $OUT/${TAG}_ba16_track.txt 2>&1
CMS_BA_RM_VALU=1 python $R/tools/prof_ba_many.py 16 track diff > $OUT/${TAG}_ba16_track_valu.txt 2>&1
CMS_BA_NO_RUNS=1 python $R/tools/prof_ba_many.py 16 track diff > $OUT/${TAG}_ba16_track_edges_only.txt 2>&1
python $R/tools/prof_ba_many.py 16 random diff > $OUT/${TAG}_ba16_random.txt 2>&1
python $R/tools/prof_ba_many.py 1 track > $OUT/${TAG}_ba1_track.txt 2>&1
bash $R/tools/pmc_mix.sh > $OUT/${TAG}_pmc_mix.log 2>&1
ls $OUT
