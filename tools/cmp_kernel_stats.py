import csv, sys
a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
tot = [0, 0]
rows = []
for k in set(a) | set(b):
    ta = float(a[k]["TotalDurationNs"]) / 1e6 if k in a else 0; tb = float(b[k]["TotalDurationNs"]) / 1e6 if k in b else 0
    ca = int(a[k]["Calls"]) if k in a else 0; cb = int(b[k]["Calls"]) if k in b else 0
    tot[0] += ta; tot[1] += tb
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort(key=lambda r: -abs(r[0]))
print("total kernel ms: %.1f vs %.1f" % tuple(tot))
for r in rows[:22]:
    print("%+9.2f ms  %-44s calls %6d %9.2f ms | %6d %9.2f ms" % (r[0], r[1][:44], r[2], r[3], r[4], r[5]))
