"""Optimizer::PoseOptimization (Optimizer.cpp:48-190) written from the reference's text in numpy -- the independent check of the oracle's
(and through it the product's) pose-only optimisation.  The edge is EdgeSE3ProjectXYZMultiPinholeOnlyPose (g2o_cubemap_vertices_edges.h:42-88,
.cpp:61-124): residual = measurement - multipinhole_project(T.map(Xw)) with the float cast, Jacobian of linearizeOplus; one 6x6 system,
solved densely; Levenberg iteration, Huber kernel, pose update and the nBad rule as in tests/npref_ba.py (same g2o files).  The four rounds
(:135-175): estimate reset to the frame's pose, optimize(10) on the level-0 edges, then every edge re-classified by chi2 > 5.991 evaluated in
FLOAT -- edges that were outliers get a fresh error at the round's final pose first, the others keep the error of the round's last trial --,
the kernel dropped after the third round, early exit for frames with fewer than ten edges; < 3 edges: nothing happens, 0 returned."""
import numpy as np
import npref_ba as nb


def pose_optimize(pr):
    N = len(pr["Xw"])
    pose0 = pr["pose0"].copy(); pose0[3:] = nb.normalize_rot(pose0[3:])
    if N < 3:
        return 0, pose0, np.zeros(N, np.uint8), []
    prob = dict(fx=pr["fx"], fy=pr["fy"], cx=pr["cx"], cy=pr["cy"])
    err = np.zeros((N, 2)); level = np.zeros(N, int); outlier = np.zeros(N, np.uint8)
    use_kernel = True
    delta = np.sqrt(5.991)
    pose = pose0.copy()
    its = []

    def compute(edges, T):
        for e in edges:
            uv, _ = nb.project(prob, T, pr["Xw"][e], pr["face"][e])
            err[e] = pr["obs"][e] - uv

    def chi2(e):
        return pr["invsig2"][e] * float(err[e] @ err[e])

    def rchi(edges):
        s = 0.0
        for e in edges:
            c = chi2(e)
            s += c if (not use_kernel or np.sqrt(c) <= delta) else 2 * np.sqrt(c) * delta - delta * delta
        return s

    n_bad = 0
    for rnd in range(4):
        pose = pose0.copy()
        edges = [e for e in range(N) if level[e] == 0]
        done = 0
        if edges:
            lam, ni, nbad_it = -1.0, 2.0, 0
            for it in range(10):
                compute(edges, pose)
                cur = rchi(edges); ini = cur
                H = np.zeros((6, 6)); b = np.zeros(6)
                for e in edges:
                    Jp, _ = nb.jacobians(prob, pose, pr["Xw"][e], pr["face"][e])
                    c = chi2(e)
                    w = 1.0 if (not use_kernel or np.sqrt(c) <= delta) else delta / np.sqrt(c)
                    om = w * pr["invsig2"][e]
                    H += om * Jp.T @ Jp; b -= om * Jp.T @ err[e]
                if it == 0:
                    lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0; nbad_it = 0
                rho, qmax = 0.0, 0
                while True:
                    x = np.linalg.solve(H + lam * np.eye(6), b)
                    trial = nb.pose_update(pose, x)
                    compute(edges, trial)
                    temp = rchi(edges)
                    rho = (cur - temp) / (float(x @ (lam * x + b)) + 1e-3)
                    if rho > 0 and np.isfinite(temp):
                        lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)); ni = 2.0; cur = temp; pose = trial
                    else:
                        lam *= ni; ni *= 2
                    qmax += 1
                    if not (rho < 0 and qmax < 10):
                        break
                done += 1
                if qmax == 10 or rho == 0:
                    break
                nbad_it = nbad_it + 1 if (ini - cur) * 1e3 < ini else 0
                if nbad_it >= 3:
                    break
        its.append(done)
        n_bad = 0
        for e in range(N):
            if outlier[e]:
                compute([e], pose)
            if np.float32(chi2(e)) > np.float32(5.991):
                outlier[e] = 1; level[e] = 1; n_bad += 1
            else:
                outlier[e] = 0; level[e] = 0
        if rnd == 2:
            use_kernel = False
        if N < 10:
            break
    return N - n_bad, pose, outlier, its
