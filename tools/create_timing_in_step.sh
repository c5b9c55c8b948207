#!/bin/bash
# developer helper (GPU box): cms_ba_create's host phases inside the bench's step, averaged over the last 400 windows
mkdir -p gpurun_out/r04
CMS_BA_CREATE_TIMING=1 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 2>&1 | grep "cms_ba_create\]" | tail -400 > gpurun_out/r04/create_timing.txt
python - <<'PY'
import re, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/r04/create_timing.txt"):
    for k, v in re.findall(r" ([a-z\-]+) ([0-9.]+)", l.split("ms:")[1]):
        acc[k].append(float(v))
print({k: round(sum(v) / len(v), 3) for k, v in acc.items()}, len(acc["runs"]), "windows; sum", round(sum(sum(v) / len(v) for v in acc.values()), 2), "ms")
PY
