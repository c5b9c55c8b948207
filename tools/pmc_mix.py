"""Summarise the counter_collection CSVs written by tools/pmc_mix.sh: per kernel, per counter, the mean over its dispatches."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not k.startswith(("k_", "kb_")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    w = m.get("SQ_WAVES", 0) or 1
    print(k, "waves=%d" % w, " ".join("%s/wave=%.1f" % (c.replace("SQ_", ""), v / w) for c, v in sorted(m.items()) if c != "SQ_WAVES"))
