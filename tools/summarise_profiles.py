#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag>_* (rocprofv3 csv output) into the small, tracked summaries under profiles/."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof"); dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = os.path.join(src, tag + "_bench_kernel_stats.csv")
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --cpu-frames 0   (MI355X, 1 GPU)\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(",".join([r["Name"].split("(")[0][:60], r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                              "%.2f" % float(r["Percentage"]), r["MinNs"], r["MaxNs"]]) + "\n")
    bj = os.path.join(src, tag + "_bench.json")
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith("{")]
        if line:
            open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line[-1])
for extra, cmd in (("mapping", "python tools/prof_tri.py 8 20 5"), ("ba8", "python tools/prof_ba_many.py 8")):
    st = os.path.join(src, "%s_%s_kernel_stats.csv" % (tag, extra))
    if os.path.exists(st):
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, extra)), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- %s   (MI355X, 1 GPU)\n" % cmd)
            lg = os.path.join(src, "%s_%s.log" % (tag, extra))
            if os.path.exists(lg):
                for l in open(lg):
                    if "ms" in l and not l.startswith(("W2", "E2", "I2")):
                        f.write("# " + l.strip() + "\n")
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in csv.DictReader(open(st)):
                f.write(",".join([r["Name"].split("(")[0][:60], r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                                  "%.2f" % float(r["Percentage"]), r["MinNs"], r["MaxNs"]]) + "\n")
pmc = {}
for kind in ("fetch", "write"):
    p = os.path.join(src, "%s_pmc_%s_counter_collection.csv" % (tag, kind))
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        a = agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in agg.items():
        pmc.setdefault(k, {})[c] = {"sum": v, "dispatches": n, "per_dispatch": v / n}
if pmc:
    with open(os.path.join(dst, tag + "_pmc_hbm_traffic.json"), "w") as f:
        json.dump({"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/prof_frames.py 64 550 3",
                   "note": "counter unit: KB (rocprofv3 derived metric); separate passes for FETCH_SIZE and WRITE_SIZE; "
                           "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): bench.py doubles it; "
                           "calibration on this path: k_fast_cells with the XCD-striped cell list reports 183 MB against >= 300 MB of "
                           "non-zero cell bytes it must read (x2 = 366 MB incl. the 3-px halos); the plain row-major list reported 854 MB",
                   "kernels": {k: v for k, v in pmc.items() if k.startswith("k_")}}, f, indent=1)
print("profiles/:", sorted(os.listdir(dst)))
