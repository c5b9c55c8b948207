"""Developer aid: the K = 40 window of test_ba_plan_kernel_on_small_and_odd_windows (seed 303) on the default path, in the deterministic mode (39 free key frames: the
pair-owner kernel on the host's full plan) and in the oracle, plus the oracle against itself under a 1e-12 m perturbation -- how sensitive the window is at rounding level.

    python tools/diag_k40_det.py        (on the GPU box)
"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from cubemapslam_amd import api, synth
import test_gpu_parity as T
orc = T.orc
prob = synth.ba_problem(K=40, P=1500, obs_per_point=4, F=550, seed=303, views="track")
w = orc.ba_run(prob)
g0 = api.ba_run(prob)
api.ba_set_deterministic(True)
g1 = api.ba_run(prob); g2 = api.ba_run(prob)
api.ba_set_deterministic(False)
def rel(a, b):
    du = np.linalg.norm(w["poses"][:, :3] - prob["poses"][:, :3], axis=1); e = np.linalg.norm(a["poses"][:, :3] - b["poses"][:, :3], axis=1)
    return float((e / np.maximum(du, 0.01 * np.median(du[du > 0]))).max())
print("iterations default/det/oracle", list(g0["stats"].iterations_done), list(g1["stats"].iterations_done), list(w["stats"].iterations_done))
print("default vs oracle %.3g, det vs oracle %.3g, det vs default %.3g, det repeat equal %s" % (rel(g0, w), rel(g1, w), rel(g1, g0), np.array_equal(g1["poses"], g2["poses"]) and np.array_equal(g1["points"], g2["points"])))
try:
    T._ba_updates_close_or_cascade(prob, g1["poses"], g1["points"], w, tag="K40 det")
    print("cascade-aware bar: ok")
except AssertionError as e:
    print("cascade-aware bar FAILED", str(e)[:300])
# the oracle against itself under a 1e-12 perturbation
p2 = dict(prob, points=prob["points"] + 1e-12)
w2 = orc.ba_run(p2)
print("oracle vs oracle(+1e-12 m) %.3g" % rel(w2, w))
