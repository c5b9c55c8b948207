"""developer helper: which HIP streams of a profiled run shared which hardware queue (rocprofv3 --kernel-trace csv), with their kernels"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t_end = max(int(r["End_Timestamp"]) for r in rows); lo = t_end - int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 0
by = collections.defaultdict(collections.Counter); q = collections.defaultdict(set); busy = collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) < lo: continue
    by[r["Stream_Id"]][r["Kernel_Name"].split("(")[0][:24]] += 1; q[r["Stream_Id"]].add(r["Queue_Id"])
    busy[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for s, c in sorted(by.items(), key=lambda x: (sorted(q[x[0]]), -sum(x[1].values()))):
    print("queue %-6s stream %4s  kernels %5d  busy %7.1f ms : %s" % (",".join(sorted(q[s])), s, sum(c.values()), busy[s], ", ".join("%s x%d" % (k, n) for k, n in c.most_common(4))))
