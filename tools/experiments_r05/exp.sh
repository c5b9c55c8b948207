#!/bin/bash
# Round-5 GPU experiments, one parametrised script (run on the GPU box: gpurun -- 'bash tools/experiments_r05/exp.sh <case> [args]').  Output: gpurun_out/r05/.
# tools/gb.sh <tag> [bench args] prints one line of key numbers of a bench.py run; environment knobs are inherited.  Cases:
#   tests [-k expr]        pytest -m gpu (all, or a subset)
#   create                 host cost of cms_ba_create, device-side planner against CMS_BA_HOST_PLAN=1 (tools/prof_ba_create.py, CMS_BA_CREATE_TIMING)
#   create_in_step         the same phases inside bench.py's step (last 400 windows averaged)
#   planners               bench step with the device-side planner / the host planner, twice each
#   mapping                bench step with LocalMapping's whole sequence / CMS_BENCH_MAPPING_MINIMAL=1, twice each, + config.mapping_side of the last run
#   knobs <tag> [ENV=..]   one bench run under the given environment (e.g. CMS_BA_SET_STREAM_WAIT=1, CMS_BENCH_MAP_PRIORITY=high, CMS_BENCH_BA_PRIORITY=high,
#                          CMS_BA_RELAXED_WAIT=1, CMS_BENCH_SWITCH_INTERVAL_US=5000, CMS_RESIZE_FUSED=1); further args go to bench.py (--window-threads 8, --ba-groups 3)
#   steptrace <tag> [ENV=..]  bench.py under rocprofv3 --kernel-trace --stats: in-step average duration of every kernel
#   steptable <tag>        bench.py under rocprofv3 --kernel-trace -> tools/step_table.py (per-step launches, summed and average durations)
#   frames                 the frame path alone: one pyramid level per launch (default) against CMS_RESIZE_FUSED=1 (two levels per launch)
#   ba16 [libs]            tools/prof_ba_many.py 16 track diff 3 (the Schur kernel alone) per library variant (default | ab_NAME from tools/ab_build.sh)
# (profiles/r05_experiments.txt collects the outputs DESIGN.md quotes)
set -u
O=gpurun_out/r05; mkdir -p $O
libpath() { if [ "$1" = "default" ]; then echo $PWD/cubemapslam_amd/lib/libcubemapslam_hip.so; else echo $PWD/cubemapslam_amd/lib/ab_$1.so; fi; }
case "$1" in
  tests) shift; timeout 1200 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -8 | tee $O/tests.txt ;;
  create)
    CMS_BA_CREATE_TIMING=1 timeout 120 python tools/prof_ba_create.py 8 2>&1 | tail -3 | tee $O/create_device_planner.txt
    CMS_BA_HOST_PLAN=1 CMS_BA_CREATE_TIMING=1 timeout 120 python tools/prof_ba_create.py 8 2>&1 | tail -3 | tee $O/create_host_planner.txt ;;
  create_in_step)
    CMS_BA_CREATE_TIMING=1 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 2>&1 | grep "cms_ba_create\]" | tail -400 > $O/create_timing.txt
    python - $O/create_timing.txt <<'PY' | tee $O/create_in_step.txt
import re, collections, sys
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    for k, v in re.findall(r" ([a-z\-]+) ([0-9.]+)", l.split("ms:")[1]):
        acc[k].append(float(v))
print({k: round(sum(v) / len(v), 3) for k, v in acc.items()}, "windows", len(next(iter(acc.values()))), "sum", round(sum(sum(v) / len(v) for v in acc.values()), 2), "ms")
PY
    ;;
  planners) for i in 1 2; do bash tools/gb.sh r05_device_plan$i | cut -c1-260 | tee -a $O/planners.txt; CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05_host_plan$i | cut -c1-260 | tee -a $O/planners.txt; done ;;
  mapping)
    for i in 1 2; do bash tools/gb.sh r05_map_full$i | cut -c1-260 | tee -a $O/mapping.txt; CMS_BENCH_MAPPING_MINIMAL=1 bash tools/gb.sh r05_map_min$i | cut -c1-260 | tee -a $O/mapping.txt; done
    python - <<'PY' | tee -a $O/mapping.txt
import json
l = [x for x in open('gpurun_out/gb_r05_map_full2.log') if x.startswith('{')]
c = json.loads(l[-1])['config']
print({k: v for k, v in c['mapping_side'].items() if k not in ('note', 'calls_per_key_frame')}, c['ba_worker_ms'], c['host'])
PY
    ;;
  knobs) shift; TAG=$1; shift; ENVS=(); while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do ENVS+=("$1"); shift; done; env "${ENVS[@]}" bash tools/gb.sh r05_$TAG "$@" | cut -c1-260 | tee -a $O/knobs.txt ;;
  steptrace)
    shift; TAG=$1; shift
    R=$PWD; OUT=$R/$O/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 > $OUT/bench.json 2> $OUT/bench.err)
    python - $OUT/t_kernel_stats.csv $TAG <<'PY' | tee -a $O/steptrace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("== %s (whole run: set-up launches included)" % sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-34s calls %6s  avg %9.1f us  max %9.1f  total %8.2f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
    rm -f $OUT/*_kernel_trace.csv $OUT/*_agent_info.csv ;;
  steptable)              # per-step chip-time table of the bench step as it is (tools/step_table.py) -> gpurun_out/r05/step_table_<tag>.md
    shift; TAG=${1:-now}; R=$PWD; OUT=$R/$O/st_$TAG; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 > $OUT/bench.json 2> $OUT/bench.err)
    python - "$OUT/t_kernel_trace.csv" "$OUT/trace_small.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0][:40] if k == "Kernel_Name" else r[k] for k in keep])
PY
    gzip -f $OUT/trace_small.csv; rm -f $OUT/t_kernel_trace.csv $OUT/t_agent_info.csv
    python tools/step_table.py $OUT/trace_small.csv.gz 6 > $O/step_table_$TAG.md; head -60 $O/step_table_$TAG.md | cut -c1-150 ;;
  frames)
    for i in 1 2; do
      echo "two levels per launch: $(CMS_RESIZE_FUSED=1 timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)" | tee -a $O/frames.txt
      echo "one level per launch:  $(timeout 300 python tools/prof_frames.py 256 550 5 2>&1 | tail -1)" | tee -a $O/frames.txt
    done ;;
  ba16) shift; for v in ${@:-default}; do echo "$v: $(CMS_HIP_LIB=$(libpath $v) timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-200)" | tee -a $O/ba16.txt; done ;;
  *) echo "unknown case $1"; exit 2 ;;
esac
