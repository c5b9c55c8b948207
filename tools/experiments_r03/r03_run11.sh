#!/bin/bash
set -u
O=gpurun_out/r11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_harness.py -m gpu -q -x > $O/t_harness.log 2>&1; tail -25 $O/t_harness.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "distance_bounds" > $O/t_bounds.log 2>&1; tail -5 $O/t_bounds.log
python - <<'PY' 2>&1 | tail -20
import sys, tempfile, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cubemapslam_amd import harness, synth
camd = synth.camera("lafida", 550)
mask = synth.cubemap_valid_mask(camd)
frames, gts = harness.render_sequence(camd, 40)
d = tempfile.mkdtemp()
harness.export_sequence(d, camd, frames, gts, mask)
rc, recs, out = harness.run_driver(d)
print("rc", rc); print(out[-400:])
tr = [r["ms"] for r in recs if r.get("stage") == "track" and "ba_iterations" not in r and "n_inliers" in r]
kf = [r["ms"] for r in recs if "ba_iterations" in r]
print("driver: tracked median %.3f ms (min %.3f), key frame median %.3f ms, frames %d" % (np.median(tr), min(tr), np.median(kf) if kf else -1, len(recs)))
for r in recs[:8]: print(r)
PY
