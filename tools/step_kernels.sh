#!/bin/bash
# developer helper (GPU box): in-step durations of the Levenberg round's kernels under rocprofv3 for the given environment (tag first)
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/sk_$tag
env "$@" CMS_BENCH_NO_PY_LOOP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sk_$tag -o t -- python $R/bench.py --steps ${STEPS:-10} --warmup 3 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass --optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 --confined-steps 0 > $R/gpurun_out/sk_$tag.json 2> $R/gpurun_out/sk_$tag.err
python - $R/gpurun_out/sk_$tag/t_kernel_stats.csv $R/gpurun_out/sk_$tag.json $tag <<'PY'
import csv, sys, json
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
l = [x for x in open(sys.argv[2]) if x.startswith("{")]
v = json.loads(l[-1])["value"] if l else None
print("== %s: value under the profiler %s" % (sys.argv[3], v))
for k in ("kb_ba_lin_schur_runs", "kb_ba_trial_solve3r", "kb_ba_trial_edges", "k_fast_cells", "k_resize", "k_ba_expand_edges_many", "k_ba_results_to_host_many", "k_kf_update_poses", "k_area_query"):
    if k in rows:
        r = rows[k]
        print("   %-28s calls %5s  avg %8.1f us  total %8.2f ms" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
