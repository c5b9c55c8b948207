"""Per-step chip-time table of bench.py's step from one rocprofv3 --kernel-trace (tools/r03 scripts write gpurun_out/*/trace_small.csv.gz:
kernel name, start, end, queue, stream): which kernels the step is made of, how often they run, their summed and average duration, how
much of the wall time the GPU has 1 / 2 / 3+ kernels in flight.  Steps are delimited by k_remap launches (one per step).

    python tools/step_table.py gpurun_out/r5/trace_small.csv.gz [steps to average, default 4] [stats.csv] > profiles/r03_step_table.md

With a third argument the per-kernel statistics of THOSE steps only (calls, total, average, share, minimum, maximum: the columns of rocprofv3's
--stats file, which covers the whole process and therefore the set-up's launches too) are written there as CSV.
"""
import collections, csv, gzip, sys

rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt")))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t = [r["s"] for r in rows if r["Kernel_Name"].startswith("k_remap")]
a, b = t[-nst - 1], t[-1]
sel = [r for r in rows if a <= r["s"] < b]
agg = collections.defaultdict(lambda: [0, 0])
mn = {}; mx = {}
for r in sel:
    agg[r["Kernel_Name"]][0] += 1; agg[r["Kernel_Name"]][1] += r["e"] - r["s"]
    mn[r["Kernel_Name"]] = min(mn.get(r["Kernel_Name"], 1 << 62), r["e"] - r["s"]); mx[r["Kernel_Name"]] = max(mx.get(r["Kernel_Name"], 0), r["e"] - r["s"])
if len(sys.argv) > 3:
    tot_all = sum(v[1] for v in agg.values()) or 1
    with open(sys.argv[3], "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for k in sorted(agg, key=lambda k: -agg[k][1]):
            f.write("%s,%d,%d,%.1f,%.2f,%d,%d\n" % (k, agg[k][0], agg[k][1], agg[k][1] / agg[k][0], 100.0 * agg[k][1] / tot_all, mn[k], mx[k]))
ev = sorted([(r["s"], 1) for r in sel] + [(r["e"], -1) for r in sel])
busy = 0; depth = 0; last = None; conc = collections.Counter()
for tt, d in ev:
    if depth > 0:
        busy += tt - last; conc[min(depth, 3)] += tt - last
    depth += d; last = tt
wall = (b - a) / nst / 1e6
print("# bench.py step under rocprofv3 --kernel-trace: %d steps averaged, %.2f ms wall per step (profiled), %.0f kernels per step" % (nst, wall, len(sel) / nst))
print()
print("GPU busy (at least one kernel in flight) %.2f ms per step = %.0f %% of the wall time; exactly one kernel %.2f ms, two %.2f ms, three or more %.2f ms; "
      "summed kernel time %.2f ms." % (busy / nst / 1e6, 100 * busy / nst / 1e6 / wall, conc[1] / nst / 1e6, conc[2] / nst / 1e6, conc[3] / nst / 1e6,
                                      sum(v[1] for v in agg.values()) / nst / 1e6))
print()
groups = [("local BA: Levenberg rounds", ("kb_ba_lin_schur", "kb_ba_trial_", "kb_ba_reduce2", "kb_ba_schur_edges_reduce")),
          ("local BA: per stage / per window", ("kb_ba_first_pass", "k_copy16", "kb_ba_lin", "kb_ba_maxdiag", "kb_ba_errors", "kb_ba_reduce", "kb_ba_classify", "kb_ba_lm_load", "kb_ba_counts", "k_ba_reset", "k_ba_gather", "k_ba_expand", "k_ba_unpermute", "k_ba_results")),
          ("CreateNewMapPoints", ("k_tri_",)),
          ("key-frame insertion, SearchInNeighbors (Fuse), pose write-back", ("k_kf_", "k_fuse_")),
          ("frame path: remap + ORB extraction", ("k_remap", "k_resize", "k_fast_cells", "k_quadtree", "k_cull", "k_describe")),
          ("frame path: grids, searches, pose optimisation", ("k_area_", "k_search_local", "k_project_last", "k_rot_filter", "k_in_frustum", "k_pose_optimize")),
          ("copies (runtime blit kernels) and fills", ("__amd_rocclr",)), ("other", ("",))]
done = set()
print("| kernel | launches / step | ms / step (summed) | average us |")
print("|---|---|---|---|")
for title, prefixes in groups:
    ks = [k for k in agg if k not in done and any(k.startswith(p) for p in prefixes)]
    if title.startswith("local BA: per stage"):
        ks = [k for k in ks if not k.startswith(("kb_ba_lin_schur", "kb_ba_reduce2"))]
    if not ks:
        continue
    tot = sum(agg[k][1] for k in ks) / nst / 1e6
    print("| **%s** | | **%.3f** | |" % (title, tot))
    for k in sorted(ks, key=lambda k: -agg[k][1]):
        done.add(k)
        print("| `%s` | %.1f | %.3f | %.1f |" % (k, agg[k][0] / nst, agg[k][1] / nst / 1e6, agg[k][1] / agg[k][0] / 1e3))
