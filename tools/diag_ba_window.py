"""Developer diagnostic: one synthetic window through the product (current library, or CMS_HIP_LIB) against the oracle; which points miss
the 1e-4 bar, and what they look like.   python tools/diag_ba_window.py seed views [K P obs dropout]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import orc
from cubemapslam_amd import api, synth
seed = int(sys.argv[1]); views = sys.argv[2]
K, P, obs = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (20, 22150, 4)
drop = float(sys.argv[6]) if len(sys.argv) > 6 else 0.07
prob = synth.ba_problem(K=K, P=P, obs_per_point=obs, F=550, seed=seed, views=views, dropout=drop)
w = orc.ba_run(prob)
g = api.ba_run(prob)
nrm = np.linalg.norm(w["points"] - prob["points"], axis=1); err = np.linalg.norm(g["points"] - w["points"], axis=1)
floor = 0.01 * np.median(nrm[nrm > 0]); r = err / np.maximum(nrm, floor)
bad = np.flatnonzero(r > 1e-4)
tn = np.linalg.norm(w["poses"][:, :3] - prob["poses"][:, :3], axis=1); te = np.linalg.norm(g["poses"][:, :3] - w["poses"][:, :3], axis=1)
print("lib", os.environ.get("CMS_HIP_LIB", "default"), {k: v for k, v in os.environ.items() if k.startswith("CMS_BA")},
      "its", list(g["stats"].iterations_done), list(w["stats"].iterations_done), "flags differ", int((g["outliers"] != w["outliers"]).sum()),
      "bad points", len(bad), "max rel point %.3g" % r.max(), "max rel pose t %.3g" % (te / np.maximum(tn, 1e-12)).max())
cnt = np.bincount(prob["e_point"], minlength=P)
nout = np.bincount(prob["e_point"], weights=w["outliers"], minlength=P)
try:
    pl = api.ba_plan(prob["fixed"], P, prob["e_pose"], prob["e_point"])
    prank = np.empty(P, np.int64); prank[pl["pinv"]] = np.arange(P)
except Exception:
    pl = None
for p in bad[:12]:
    es = np.flatnonzero(prob["e_point"] == p)
    print("  point %d rel %.2e err %.2e upd %.2e obs %d outliers %d poses %s depth %.1f %s" % (
        p, r[p], err[p], nrm[p], cnt[p], nout[p], sorted(prob["e_pose"][es].tolist()), np.linalg.norm(prob["points"][p]),
        "" if pl is None else ("internal %d (%s)" % (prank[p], "run" if prank[p] < pl["rm_points"] else "left-over"))))
