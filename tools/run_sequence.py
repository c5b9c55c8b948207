"""Single-stream figure next to bench.py's batch figure: one rendered Lafida-cam0-like stream (ray-cast box room, F = 550) through the
closed-loop harness with the product backend, frame after frame like the reference's main loop (cubemap_lafida.cpp:128-154).

    python tools/run_sequence.py [frames] [face]        (on the GPU box)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import harness, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
F = int(sys.argv[2]) if len(sys.argv) > 2 else 550
camd = synth.camera("lafida", F)
mask = synth.cubemap_valid_mask(camd)
frames, gts = harness.render_sequence(camd, n)
be = harness.GpuBackend(camd, mask)
harness.run_sequence(camd, be, frames[:6], gts[:6])                 # warm-up: allocations, first launches
trk, secs = harness.run_sequence(camd, be, frames, gts)
secs = np.array(secs)
track = np.array([s for s, r in zip(secs, trk.log) if r["stage"] == "track" and "ba_iterations" not in r])
kf = np.array([s for s, r in zip(secs, trk.log) if "ba_iterations" in r])
err = []
for r in trk.log:
    if "pose" in r:
        T = r["pose"].astype(np.float64); Rg, tg = gts[r["frame"]]
        err.append(float(np.linalg.norm((-T[:3, :3].T @ T[:3, 3]) - (-Rg.T @ tg))))
print(json.dumps({"single_stream_closed_loop": True, "frames": n, "face": F, "state": trk.state, "frames_per_s": round(len(secs) / secs.sum(), 1),
                  "median_ms_tracked_frame": round(1e3 * float(np.median(track)), 3) if len(track) else None,
                  "median_ms_key_frame_incl_local_ba": round(1e3 * float(np.median(kf)), 3) if len(kf) else None,
                  "key_frames": len(trk.kfs), "map_points": int(len(trk.mp_pos)), "mean_inliers": round(float(np.mean([r["n_inliers"] for r in trk.log if "n_inliers" in r])), 1),
                  "max_position_error_m": round(max(err), 4) if err else None,
                  "note": "host glue in Python (numpy problem assembly) is inside these times"}))
