"""Local bundle adjustment written from the reference's text alone, in numpy, with DENSE normal equations -- an independent check of the
oracle's Levenberg loop, Schur solve and classification (tests/test_oracle_ba.py::test_levenberg_loop_against_a_dense_numpy_restatement).

What is restated, and from where (none of it from oracle/):
  * the edge: _error = measurement - multipinhole_project(T.map(X)), the camera-frame point cast to float, rotated into its face, projected
    in float with double intrinsics (g2o_cubemap_vertices_edges.h:100-112, .cpp:225-233, CamModelGeneral.cpp:228-263, CamModelGeneral.h:418-443);
    Jacobians of linearizeOplus (.cpp:164-223); isDepthPositive (.h:114-118)
  * the robust kernel: Huber, rho = e2 | 2 sqrt(e2) delta - delta^2, weight rho' (ThirdParty/g2o/g2o/core/robust_kernel_impl.cpp:78-91), applied
    as information *= rho', b uses the weighted information (base_binary_edge.hpp:54-120)
  * one Levenberg iteration (ThirdParty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189): chi2 and system at the current estimate,
    lambda_0 = 1e-5 max diag(H) on the first iteration of an optimize() call, trials with rho = (chi - chi_trial) / (dx . (lambda dx + b) + 1e-3),
    accept: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2; reject: lambda *= ni, ni *= 2, estimate restored; at most 10 trials;
    stop when the trials ran out or rho == 0, or after three iterations in a row that gained less than 1e-3 of the chi2 (the nBad rule of the
    reference's own copy of g2o)
  * the errors an edge KEEPS are those of the last trial, accepted or not (the stack restores estimates, not errors): Optimizer.cpp:377, :399 read
    e->chi2() from them
  * the vertex updates: exp(dx) * T for a key frame (types_six_dof_expmap.h:73-76, se3quat.h:217-285), X + dx for a point
  * Optimizer::LocalBundleAdjustment's two stages and classifications (Optimizer.cpp:359-412): 5 iterations with Huber(sqrt 5.991), edges with
    chi2 > 5.991 or a point behind the key frame go to level 1, 10 iterations without the kernel on the level-0 edges, the same test again.
The linear algebra differs on purpose: the full (6 K_free + 3 P_active) system is solved densely (numpy), no Schur complement."""
import numpy as np

RF = {0: np.eye(3), 1: np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0.]]), 2: np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0.]]),
      3: np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.]]), 4: np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.]])}      # cvtRigToFaces: FRONT LEFT RIGHT UPPER LOWER


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_quat(R):
    """Eigen's Quaterniond(Matrix3d)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]: i = 1
    if R[2, 2] > R[i, i]: i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    return q


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def normalize_rot(q):
    """SE3Quat::normalizeRotation: w >= 0, unit norm"""
    q = q.copy()
    if q[3] < 0: q = -q
    return q / np.linalg.norm(q)


def se3_exp(u):
    om, up = u[:3], u[3:]
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    Om2 = Om @ Om
    if th < 0.00001:
        R = np.eye(3) + Om + Om2; V = R
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om2
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om2
    return normalize_rot(R_quat(R)), V @ up


def pose_update(pose, u):
    """exp(u) * T"""
    q, t = se3_exp(u)
    tn = quat_R(q) @ pose[:3] + t
    return np.concatenate([tn, normalize_rot(qmul(q, pose[3:]))])


def project(prob, pose, X, face):
    Xc = quat_R(pose[3:]) @ X + pose[:3]
    l = (RF[int(face)] @ Xc.astype(np.float32).astype(np.float64)).astype(np.float32)          # cv::Vec3f rigPt, cvtRigToFaces<float>
    u = np.float32(np.float64(l[0]) * prob["fx"] / np.float64(l[2]) + prob["cx"])             # float _x * double fx / float _z + double cx -> float
    v = np.float32(np.float64(l[1]) * prob["fy"] / np.float64(l[2]) + prob["cy"])
    return np.array([np.float64(u), np.float64(v)]), Xc


def jacobians(prob, pose, X, face):
    R = quat_R(pose[3:])
    Xc = R @ X + pose[:3]
    Rl = RF[int(face)]
    l = Rl @ Xc
    d = np.array([[prob["fx"] / l[2], 0, -prob["fx"] * l[0] / l[2] ** 2], [0, prob["fy"] / l[2], -prob["fy"] * l[1] / l[2] ** 2]])
    dR = -1.0 * d @ Rl
    neg_skew = np.array([[0, Xc[2], -Xc[1]], [-Xc[2], 0, Xc[0]], [Xc[1], -Xc[0], 0]])
    return dR @ np.hstack([neg_skew, np.eye(3)]), dR @ R          # d error / d pose (rotation first), d error / d point


class Window:
    def __init__(self, prob):
        self.p = prob
        self.poses = prob["poses"].copy(); self.pts = prob["points"].copy()
        for k in range(len(self.poses)):
            self.poses[k, 3:] = normalize_rot(self.poses[k, 3:])                              # SE3Quat constructor
        self.E = len(prob["e_pose"])
        self.level = np.zeros(self.E, int)
        self.err = np.zeros((self.E, 2))                                                       # every edge's stored _error
        self.free = [k for k in range(len(self.poses)) if not prob["fixed"][k]]

    def active(self):
        return [e for e in range(self.E) if self.level[e] == 0]

    def compute_errors(self, edges):
        for e in edges:
            k, j = self.p["e_pose"][e], self.p["e_point"][e]
            uv, _ = project(self.p, self.poses[k], self.pts[j], self.p["e_face"][e])
            self.err[e] = self.p["e_obs"][e] - uv

    def chi2(self, e):
        return self.p["e_invsig2"][e] * float(self.err[e] @ self.err[e])

    def robust_chi2(self, edges, robust, delta):
        s = 0.0
        for e in edges:
            c = self.chi2(e)
            s += (c if (not robust or np.sqrt(c) <= delta) else 2 * np.sqrt(c) * delta - delta * delta)
        return s

    def optimize(self, iterations, robust):
        delta = np.sqrt(5.991)
        edges = self.active()
        pts_active = sorted({int(self.p["e_point"][e]) for e in edges})
        col_p = {k: 6 * i for i, k in enumerate(self.free)}
        col_l = {j: 6 * len(self.free) + 3 * i for i, j in enumerate(pts_active)}
        n = 6 * len(self.free) + 3 * len(pts_active)
        lam, ni, n_bad, done = -1.0, 2.0, 0, 0
        if not edges:
            return 0
        for it in range(iterations):
            self.compute_errors(edges)
            cur = self.robust_chi2(edges, robust, delta); ini = cur
            H = np.zeros((n, n)); b = np.zeros(n)
            for e in edges:
                k, j = int(self.p["e_pose"][e]), int(self.p["e_point"][e])
                Jp, Jl = jacobians(self.p, self.poses[k], self.pts[j], self.p["e_face"][e])
                c = self.chi2(e)
                w = 1.0 if (not robust or np.sqrt(c) <= delta) else delta / np.sqrt(c)
                om = w * self.p["e_invsig2"][e]
                J = np.zeros((2, n))
                if k in col_p: J[:, col_p[k]:col_p[k] + 6] = Jp
                J[:, col_l[j]:col_l[j] + 3] = Jl
                H += om * J.T @ J; b -= om * J.T @ self.err[e]
            if it == 0:
                lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0; n_bad = 0
            rho, qmax = 0.0, 0
            while True:
                try:
                    x = np.linalg.solve(H + lam * np.eye(n), b); ok = bool(np.all(np.isfinite(x)))
                except np.linalg.LinAlgError:
                    x = np.zeros(n); ok = False
                saved = (self.poses.copy(), self.pts.copy())
                for k in self.free: self.poses[k] = pose_update(self.poses[k], x[col_p[k]:col_p[k] + 6])
                for j in pts_active: self.pts[j] = self.pts[j] + x[col_l[j]:col_l[j] + 3]
                self.compute_errors(edges)
                temp = self.robust_chi2(edges, robust, delta) if ok else np.finfo(float).max
                rho = (cur - temp) / (float(x @ (lam * x + b)) + 1e-3)
                if rho > 0 and np.isfinite(temp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)); ni = 2.0; cur = temp
                else:
                    lam *= ni; ni *= 2
                    self.poses, self.pts = saved
                qmax += 1
                if not (rho < 0 and qmax < 10):
                    break
            done += 1
            if qmax == 10 or rho == 0:
                break
            n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
            if n_bad >= 3:
                break
        return done

    def depth_positive(self, e):
        k, j = self.p["e_pose"][e], self.p["e_point"][e]
        return (quat_R(self.poses[k, 3:]) @ self.pts[j] + self.poses[k, :3])[2] > 0.0

    def run(self):
        it1 = self.optimize(5, True)
        for e in range(self.E):
            if self.chi2(e) > 5.991 or not self.depth_positive(e):
                self.level[e] = 1
        mid = int(self.level.sum())
        it2 = self.optimize(10, False)
        out = np.array([1 if (self.chi2(e) > 5.991 or not self.depth_positive(e)) else 0 for e in range(self.E)], np.uint8)
        return dict(poses=self.poses, points=self.pts, outliers=out, iterations=[it1, it2], n_outliers_mid=mid)
