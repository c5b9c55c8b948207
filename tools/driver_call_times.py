"""Where a tracked frame's time goes in the Python-free driver (cubemapslam_amd/host/closed_loop_driver.cpp): per boundary call, wall time.

    python tools/driver_call_times.py [frames, default 40] > profiles/r03_driver_call_times.txt
"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import harness, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
camd = synth.camera("lafida", 550)
mask = synth.cubemap_valid_mask(camd)
frames, gts = harness.render_sequence(camd, n)
d = tempfile.mkdtemp()
harness.export_sequence(d, camd, frames, gts, mask)
os.environ["CMS_DRIVER_CALL_TIMES"] = "1"
rc, recs, out = harness.run_driver(d, kf_every=5, ba_window=8, new_points_per_kf=400, warmup=n)      # a throw-away tracker goes through the images once first
print("exit code %d, %d frames (Lafida cam0 geometry, face 550; key frame every 5 frames, local BA over 8 key frames)" % (rc, len(recs)))
print(out)
ms = {}
for r in recs:
    ms.setdefault(r["stage"], []).append(r["ms"])
for k, v in ms.items():
    print("%-10s %3d frames: median %.3f ms, mean %.3f ms, min %.3f ms, max %.3f ms" % (k, len(v), float(np.median(v)), float(np.mean(v)), min(v), max(v)))
