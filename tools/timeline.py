"""developer helper: what a window group's queue does during one step, from a rocprofv3 --kernel-trace CSV.
usage: python tools/timeline.py <kernel_trace.csv[.gz]>
Prints, for the LAST complete step (delimited by k_remap launches), per queue: busy time, and for the queues that carry kb_ba_* kernels the
sequence of kernels with the idle gap in front of each (gaps > 15 us only, the rest summed)."""
import csv, gzip, sys, collections
fn = sys.argv[1]
op = gzip.open if fn.endswith(".gz") else open
rows = list(csv.DictReader(op(fn, "rt")))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0]
rows.sort(key=lambda r: r["s"])
remaps = [r["s"] for r in rows if r["n"] == "k_remap"]
if len(remaps) >= 3:
    t0, t1 = remaps[-2], remaps[-1]
else:
    bas = [r["s"] for r in rows if r["n"].startswith("kb_ba_lm_load")]
    t0, t1 = bas[-5], bas[-1]
step = [r for r in rows if t0 <= r["s"] < t1]
print("step window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
# union busy time of the whole chip
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
busy = 0; depth = 0; last = None; conc = 0
for t, d in ev:
    if depth > 0: busy += t - last; conc += depth * (t - last)
    depth += d; last = t
print("chip: some kernel running %.3f ms, mean concurrency while busy %.2f, sum of kernel time %.3f ms" % (busy / 1e6, conc / max(busy, 1), sum(r["e"] - r["s"] for r in step) / 1e6))
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    tot = sum(r["e"] - r["s"] for r in rs)
    names = collections.Counter(r["n"] for r in rs)
    print("queue %s: %d kernels, busy %.3f ms, span %.3f ms: %s" % (q, len(rs), tot / 1e6, (rs[-1]["e"] - rs[0]["s"]) / 1e6, ", ".join("%s x%d" % kv for kv in names.most_common(6))))
    if any(r["n"].startswith("kb_ba_lin_schur") for r in rs):
        prev = None; small = 0.0
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])      # kernel -> count, time, gap in front
        for r in rs:
            gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
            a = agg[r["n"]]; a[0] += 1; a[1] += (r["e"] - r["s"]) / 1e3; a[2] += max(gap, 0.0)
            if gap > 40: print("    gap %.0f us in front of %s (at %.3f ms)" % (gap, r["n"], (r["s"] - t0) / 1e6))
            prev = max(prev or 0, r["e"])
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("    %-28s x%-4d %8.1f us total  %6.1f us avg   idle in front: %7.1f us total" % (n, a[0], a[1], a[1] / a[0], a[2]))
