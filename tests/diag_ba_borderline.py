"""Developer diagnostic (not collected by pytest): closed loop on the product; every local-BA problem is also given to the oracle, and
where the outlier flags differ the robust chi2 of the edges in question is printed for both estimates (is it a borderline edge?).

    python tests/diag_ba_borderline.py [frames]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc
from cubemapslam_amd import synth, harness


def chi2_of(prob, poses, points):
    out = np.zeros(len(prob["e_pose"]))
    for e in range(len(out)):
        k, p = prob["e_pose"][e], prob["e_point"][e]
        q = poses[k]
        x, y, z, w = q[3:7]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        Xc = (R @ points[p] + q[:3]).astype(np.float32).astype(np.float64)
        f = prob["e_face"][e]
        l = [Xc, np.array([Xc[2], Xc[1], -Xc[0]]), np.array([-Xc[2], Xc[1], Xc[0]]), np.array([Xc[0], Xc[2], -Xc[1]]), np.array([Xc[0], -Xc[2], Xc[1]])][f]
        u = np.float32(l[0] * prob["fx"] / l[2] + prob["cx"]); v = np.float32(l[1] * prob["fy"] / l[2] + prob["cy"])
        r = prob["e_obs"][e] - np.array([u, v], np.float64)
        out[e] = prob["e_invsig2"][e] * (r @ r)
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
camd = synth.camera("lafida", 550)
mask = synth.cubemap_valid_mask(camd)
frames, gts = harness.render_sequence(camd, n)
gpu = harness.GpuBackend(camd, mask)
orig = gpu.local_ba


def both(prob):
    r = orig(prob)
    o = orc.ba_run(prob)
    fo = np.asarray(o["outliers"]) != 0; fg = np.asarray(r[2]) != 0
    dp = np.abs(r[0] - o["poses"]).max(); dx = np.abs(r[1] - o["points"]).max()
    print("BA: E=%d its gpu %s oracle %s  outliers %d / %d  max|dpose| %.3e max|dpoint| %.3e" % (
        len(fg), r[3], list(o["stats"].iterations_done), fg.sum(), fo.sum(), dp, dx))
    for e in np.flatnonzero(fo != fg):
        cg = chi2_of(prob, r[0], r[1])[e]; co = chi2_of(prob, o["poses"], o["points"])[e]
        print("   edge %d: chi2 product %.6f oracle %.6f (threshold 5.991)" % (e, cg, co))
    return r


gpu.local_ba = both
harness.run_sequence(camd, gpu, frames, gts, kf_every=4, ba_window=6, new_points_per_kf=400)
