"""Deterministic synthetic inputs for the hot path (numpy only; harness code, not the product compute path).

Camera intrinsics are the values of the reference's Config/*.yaml (lafida_cam0_params.yaml, front_cam_params.yaml);
datasets are not available, so images / masks / BA problems are generated here.
"""
import numpy as np

LAFIDA = dict(
    c=0.999626131079017, d=-0.0034775192597376, e=0.00385134991673147, u0=392.219508388648, v0=243.494438476351,
    invpol=[293.667187375663, 149.982043337335, -10.448650568161, 28.2295300683376, 7.13365723186292,
            0.056303218962532, 10.4144677485333, 0.166354960773665, -5.86858687381081, 1.18165998645705,
            3.1108311354746, 0.810799620714366],
    pol=[-209.200757992065, 0.0, 0.00213741670953883, -4.2203617319086e-06, 1.77146086919594e-08],
    Iw=754, Ih=480, fov_deg=190.0, nfeatures=2000)

FRONT = dict(
    c=0.999896, d=0.000052, e=0.000010, u0=653.580142, v0=359.303805,
    invpol=[583.462246, 454.202253, 13.957207, -19.762352, 80.087582, 42.643599, -90.924921, -114.523166,
            -50.405891, -8.090297],
    pol=[-3.120719e+02, 0.0, 1.007745e-03, -9.430929e-07, 1.348974e-09],
    Iw=1280, Ih=720, fov_deg=190.0, nfeatures=3000)


def camera(name="lafida", face=450, Ih=None):
    base = dict(LAFIDA if name == "lafida" else FRONT)
    base["invpol"] = list(base["invpol"])
    base["face"] = int(face)
    if Ih is not None and Ih != base["Ih"]:  # config 2: 1280x1024 synthetic variant of front_cam (v0 shifted with the centre)
        base["v0"] = base["v0"] + (Ih - base["Ih"]) / 2.0
        base["Ih"] = int(Ih)
    return base


def texture(h, w, seed):
    """Seeded gray texture with corners at many scales: value-noise octaves + random rectangles."""
    rs = np.random.RandomState(seed)
    img = np.zeros((h, w), np.float64)
    for cell, amp in ((64, 60.0), (24, 45.0), (9, 30.0), (4, 18.0)):
        gh, gw = h // cell + 2, w // cell + 2
        g = rs.randint(0, 256, size=(gh, gw)).astype(np.float64) / 255.0 - 0.5
        yy = (np.arange(h) / cell)
        xx = (np.arange(w) / cell)
        y0 = yy.astype(int); x0 = xx.astype(int)
        fy = (yy - y0)[:, None]; fx = (xx - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    img += 128.0
    nrect = (h * w) // 1500
    ys = rs.randint(0, h, nrect); xs = rs.randint(0, w, nrect)
    hs = rs.randint(3, 40, nrect); ws = rs.randint(3, 40, nrect)
    vs = rs.randint(-70, 71, nrect)
    for y, x, hh, ww, v in zip(ys, xs, hs, ws, vs):
        img[y:y + hh, x:x + ww] += v
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _horner(coefs, x):
    r = np.zeros_like(x)
    for c in coefs[::-1]:
        r = r * x + c
    return r


def world_to_img(cam, X):
    """Vectorised CamModelGeneral::WorldToImg (harness use only, not bit-critical)."""
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    n = np.sqrt(x * x + y * y)
    n = np.where(n == 0, 1e-14, n)
    theta = np.arctan(-z / n)
    inv = list(cam["invpol"]) + [0.0] * (12 - len(cam["invpol"]))
    rho = _horner(inv, theta)
    uu = x / n * rho
    vv = y / n * rho
    return uu * cam["c"] + vv * cam["d"] + cam["u0"], uu * cam["e"] + vv + cam["v0"]


_F2R = {0: lambda x, y, z: (x, y, z), 1: lambda x, y, z: (-z, y, x), 2: lambda x, y, z: (z, y, -x),
        3: lambda x, y, z: (x, -z, y), 4: lambda x, y, z: (x, z, -y)}  # face -> rig (FRONT, LEFT, RIGHT, UPPER, LOWER)
_FACE_ORIGIN = {0: (1, 1), 1: (0, 1), 2: (2, 1), 3: (1, 0), 4: (1, 2)}  # (col, row) of each face on the 3x3 cross


def cubemap_valid_mask(cam, erode=20, band=70):
    """Model-derived cubemap mask: pixels whose LUT entry lands inside the fisheye image, eroded, outer band zeroed."""
    from scipy import ndimage
    F = cam["face"]; W = 3 * F
    m = np.zeros((W, W), bool)
    jj, ii = np.meshgrid(np.arange(F, dtype=np.float64), np.arange(F, dtype=np.float64))
    for f, (cx_, cy_) in _FACE_ORIGIN.items():
        x = (jj - F / 2.0) / (F / 2.0); y = (ii - F / 2.0) / (F / 2.0); z = np.ones_like(x)
        rx, ry, rz = _F2R[f](x, y, z)
        u, v = world_to_img(cam, np.stack([rx, ry, rz], -1))
        ok = (u >= 0) & (u < cam["Iw"]) & (v >= 0) & (v < cam["Ih"])
        m[cy_ * F:(cy_ + 1) * F, cx_ * F:(cx_ + 1) * F] = ok
    if erode > 0:
        m = ndimage.binary_erosion(m, iterations=erode)
    if band > 0:
        m[:band] = False; m[-band:] = False; m[:, :band] = False; m[:, -band:] = False
    return (m.astype(np.uint8) * 255)


def rays_to_cubemap(F, X):
    """Vectorised TransformRaysToCubemap in float64 (harness use: building synthetic observations)."""
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    face = np.full(x.shape, -1, np.int32)
    up = np.full(x.shape, -1.0); vp = np.full(x.shape, -1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        conds = [
            (0, (z > 0) & (np.abs(x / z) <= 1) & (np.abs(y / z) <= 1), (x, y, z), (1, 1)),
            (2, (x > 0) & (np.abs(y / x) <= 1) & (np.abs(z / x) <= 1), (-z, y, x), (2, 1)),
            (1, (x < 0) & (np.abs(y / x) <= 1) & (np.abs(z / x) <= 1), (z, y, -x), (0, 1)),
            (4, (y > 0) & (np.abs(x / y) <= 1) & (np.abs(z / y) <= 1), (x, -z, y), (1, 2)),
            (3, (y < 0) & (np.abs(x / y) <= 1) & (np.abs(z / y) <= 1), (x, z, -y), (1, 0)),
        ]
        todo = np.ones(x.shape, bool)
        for fid, c, (lx, ly, lz), (ox, oy) in conds:
            sel = todo & c
            u = lx * (F / 2.0) / lz + F / 2.0
            v = ly * (F / 2.0) / lz + F / 2.0
            inb = sel & (u >= 0) & (u < F) & (v >= 0) & (v < F)
            face[inb] = fid; up[inb] = u[inb] + ox * F; vp[inb] = v[inb] + oy * F
            todo &= ~sel
    return face, up, vp


def face_of_pixel(F, px, py):
    i = np.floor(px / F).astype(int); j = np.floor(py / F).astype(int)
    face = np.full(px.shape, -1, np.int32)
    for f, (cx_, cy_) in _FACE_ORIGIN.items():
        face[(i == cx_) & (j == cy_)] = f
    return face


def _rot(axis, ang):
    axis = np.asarray(axis, float); axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = [(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w]
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0, 0, 0, 0]; q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    q = np.array(q)
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def ba_problem(K=20, P=20000, obs_per_point=4, F=650, seed=42, outlier_frac=0.05, nlevels=8, scale=1.2, views="random", dropout=0.07):
    """Config 4 of BASELINE.json: K keyframes on a 2 m arc (KF 0 fixed), P points, obs_per_point views each.

    views = "random": every point is seen by obs_per_point key frames drawn at random (no structure at all: ~C(K, 4) different observation
    sets).  views = "track": a map point is TRACKED -- it is seen by a stretch of consecutive key frames along the arc (obs_per_point on
    average, 2 .. obs_per_point + 3), and every observation of the stretch is missing with probability `dropout` (a failed match).  That is
    how Tracking / LocalMapping create observations (a point enters the map at a key frame and is re-observed by the following ones until it
    leaves the view or its scale range), and it is what gives a local window points that share their set of observing key frames."""
    rs = np.random.RandomState(seed)
    Rt, tt = [], []
    for k in range(K):
        a = (k / max(K - 1, 1) - 0.5) * 0.6
        Rwc = _rot([0, 1, 0], a) @ _rot([1, 0, 0], 0.05 * np.sin(3 * a))
        cw = np.array([2.0 * np.sin(a) / 0.6, 0.05 * np.cos(5 * a), 0.3 * (1 - np.cos(a))])
        Rcw = Rwc.T
        Rt.append(Rcw); tt.append(-Rcw @ cw)
    Rt = np.array(Rt); tt = np.array(tt)
    pts = np.stack([rs.uniform(-6, 6, P), rs.uniform(-3, 3, P), rs.uniform(-2, 9, P)], -1)
    near = np.linalg.norm(pts, axis=1) < 1.5
    pts[near] += np.array([0, 0, 4.0])
    sig2 = (np.float32(scale) ** np.arange(nlevels, dtype=np.float32)) ** 2
    inv_sig2_tab = (np.float32(1.0) / sig2.astype(np.float32)).astype(np.float32)
    # all (point, keyframe) projections at once, then per point the first `obs_per_point` valid views in a random order
    Xc = np.einsum("kij,pj->pki", Rt, pts) + tt[None, :, :]                   # P x K x 3
    ray_z = Xc[..., 2] / np.linalg.norm(Xc, axis=-1)
    face, up, vp = rays_to_cubemap(F, Xc)
    octave = rs.randint(0, nlevels, size=(P, K))
    sd = scale ** octave
    gross = rs.uniform(size=(P, K)) < outlier_frac
    noise = np.where(gross[..., None], rs.uniform(-30, 30, size=(P, K, 2)), rs.normal(0, 1, size=(P, K, 2)) * sd[..., None])
    px = (up + noise[..., 0]).astype(np.float32).astype(np.float64)
    py = (vp + noise[..., 1]).astype(np.float32).astype(np.float64)
    f2 = face_of_pixel(F, px, py)
    ok = (face >= 0) & (f2 >= 0) & (ray_z >= np.cos(np.deg2rad(190.0 / 2)))
    if views == "track":
        nvalid = ok.sum(1)                                      # key frames that see the point at all (in front of the camera, on a face)
        rank = np.cumsum(ok, axis=1) - 1                        # position of a key frame among those
        ln = np.minimum(np.clip(obs_per_point + rs.choice([-2, -1, -1, 0, 0, 0, 1, 1, 2, 3], size=P), 2, K), nvalid)
        start = (rs.uniform(size=P) * (nvalid - ln + 1)).astype(int)
        vis = ok & (rank >= start[:, None]) & (rank < (start + ln)[:, None])
        keep = vis & ~(rs.uniform(size=(P, K)) < dropout)
        few = keep.sum(1) < 2                                   # a map point has at least two observations: such a point keeps its whole stretch
        keep[few] = vis[few]
        pp, kk = np.nonzero(keep)
    else:
        order = np.argsort(rs.uniform(size=(P, K)), axis=1)
        ok_o = np.take_along_axis(ok, order, 1)
        take_o = ok_o & (np.cumsum(ok_o, axis=1) <= obs_per_point)
        pp, jj = np.nonzero(take_o)
        kk = order[pp, jj]
    u = px[pp, kk] - np.floor(px[pp, kk] / F) * F
    v = py[pp, kk] - np.floor(py[pp, kk] / F) * F
    e_pose, e_point = kk, pp
    e_obs = np.stack([u, v], 1)
    e_inv = inv_sig2_tab[octave[pp, kk]].astype(np.float64)
    e_face = f2[pp, kk]
    # perturb the initial estimate (1 deg / 2 cm poses, 2 cm points); inputs are float-representable doubles
    poses = np.zeros((K, 7))
    for k in range(K):
        R = Rt[k]; t = tt[k]
        if k > 0:
            ax = rs.normal(size=3)
            R = _rot(ax, np.deg2rad(1.0) * rs.uniform(0.3, 1.0)) @ R
            t = t + rs.normal(0, 0.02, 3)
        R32 = R.astype(np.float32).astype(np.float64)
        poses[k, :3] = t.astype(np.float32).astype(np.float64)
        poses[k, 3:] = _quat_from_R(R32)
    points = (pts + rs.normal(0, 0.02, pts.shape)).astype(np.float32).astype(np.float64)
    fixed = np.zeros(K, np.uint8); fixed[0] = 1
    return dict(poses=poses, fixed=fixed, points=points, e_pose=np.array(e_pose, np.int32),
                e_point=np.array(e_point, np.int32), e_obs=np.ascontiguousarray(e_obs, np.float64).reshape(-1, 2),
                e_invsig2=np.ascontiguousarray(e_inv, np.float64), e_face=np.array(e_face, np.int8),
                fx=F / 2.0, fy=F / 2.0, cx=F / 2.0, cy=F / 2.0)


def pose_problem(N=600, F=550, seed=7, outlier_frac=0.1, nlevels=8, scale=1.2, rot_deg=1.5, trans=0.05):
    """One frame of Optimizer::PoseOptimization: N map points seen on the five faces, pixel noise growing
    with the octave, `outlier_frac` gross mismatches, initial pose = ground truth perturbed by rot_deg / trans.
    Returns the edge arrays the C-ABI takes plus the ground-truth pose."""
    rs = np.random.RandomState(seed)
    a = 0.3 * rs.uniform(-1, 1)
    Rcw = (_rot([0, 1, 0], a) @ _rot([1, 0, 0], 0.1 * rs.uniform(-1, 1))).T
    tcw = -Rcw @ np.array([0.4 * rs.uniform(-1, 1), 0.1 * rs.uniform(-1, 1), 0.3 * rs.uniform(-1, 1)])
    M = 4 * N + 64
    pts = np.stack([rs.uniform(-6, 6, M), rs.uniform(-3, 3, M), rs.uniform(-2, 9, M)], -1)
    pts = pts.astype(np.float32).astype(np.float64)           # map points are float cv::Mat in the reference
    Xc = pts @ Rcw.T + tcw
    ray_z = Xc[:, 2] / np.linalg.norm(Xc, axis=1)
    face, up, vp = rays_to_cubemap(F, Xc)
    octave = rs.randint(0, nlevels, size=M)
    gross = rs.uniform(size=M) < outlier_frac
    noise = np.where(gross[:, None], rs.uniform(-40, 40, size=(M, 2)), rs.normal(0, 1, size=(M, 2)) * (scale ** octave)[:, None])
    px = (up + noise[:, 0]).astype(np.float32).astype(np.float64)
    py = (vp + noise[:, 1]).astype(np.float32).astype(np.float64)
    f2 = face_of_pixel(F, px, py)
    ok = (face >= 0) & (f2 >= 0) & (ray_z >= np.cos(np.deg2rad(190.0 / 2))) & (np.linalg.norm(Xc, axis=1) > 1.0)
    sel = np.nonzero(ok)[0][:N]
    sig2 = (np.float32(scale) ** np.arange(nlevels, dtype=np.float32)) ** 2
    inv_tab = (np.float32(1.0) / sig2).astype(np.float32)
    u = px[sel] - np.floor(px[sel] / F) * F
    v = py[sel] - np.floor(py[sel] / F) * F
    R0 = _rot(rs.normal(size=3), np.deg2rad(rot_deg) * rs.uniform(0.5, 1.0)) @ Rcw
    t0 = tcw + rs.normal(0, trans, 3)
    pose0 = np.concatenate([t0.astype(np.float32).astype(np.float64), _quat_from_R(R0.astype(np.float32).astype(np.float64))])
    pose_gt = np.concatenate([tcw, _quat_from_R(Rcw)])
    return dict(Xw=np.ascontiguousarray(pts[sel]), obs=np.ascontiguousarray(np.stack([u, v], 1)),
                invsig2=inv_tab[octave[sel]].astype(np.float64), face=f2[sel].astype(np.int8), gross=gross[sel],
                pose0=pose0, pose_gt=pose_gt, fx=F / 2.0, fy=F / 2.0, cx=F / 2.0, cy=F / 2.0)


def pixel_to_ray(F, px, py):
    """rig-frame ray (z_local = 1) through canvas pixel (px, py): inverse of rays_to_cubemap on the five faces (harness use)."""
    px = np.asarray(px, np.float64); py = np.asarray(py, np.float64)
    face = face_of_pixel(F, px, py)
    lx = (px - np.floor(px / F) * F - F / 2.0) / (F / 2.0); ly = (py - np.floor(py / F) * F - F / 2.0) / (F / 2.0); lz = np.ones_like(lx)
    X = np.zeros(px.shape + (3,))
    for fid, (rx, ry, rz) in {0: (lx, ly, lz), 1: (-lz, ly, lx), 2: (lz, ly, -lx), 4: (lx, lz, -ly), 3: (lx, -lz, ly)}.items():
        m = face == fid
        X[m, 0] = rx[m]; X[m, 1] = ry[m]; X[m, 2] = rz[m]
    return face, X


def local_map_problem(F, kx, ky, koct, kdesc, seed=3, nlevels=8, scale=1.2, extra=0.6, dup=0.25, order="random"):
    """A local map seen from one frame (Tracking::SearchLocalPoints' inputs): float pose (Rcw | tcw | Ow), map points behind a share of
    the frame's key points (position noise of a few pixels, descriptor = key point's with a few flipped bits, scale range around the key
    point's level), `dup` of them doubled (two map points competing for one key point), plus `extra` x as many that are out of view,
    too near / far, or seen from behind.  order: "random" or "spatial" (map points sorted along the image -- long claim chains)."""
    rng = np.random.default_rng(seed)
    kx = np.asarray(kx, np.float32); ky = np.asarray(ky, np.float32); koct = np.asarray(koct, np.int32)
    n = len(kx)
    ang = rng.normal(0, 0.3, 3)
    R = (_rot((1, 0, 0), ang[0]) @ _rot((0, 1, 0), ang[1]) @ _rot((0, 0, 1), ang[2])).astype(np.float32)
    t = rng.normal(0, 1.0, 3).astype(np.float32)
    Ow = (-(R.T.astype(np.float32) @ t)).astype(np.float32)
    sf = np.float32(scale) ** np.arange(nlevels, dtype=np.float32)
    pick = np.flatnonzero(rng.random(n) < 0.7)
    pick = np.concatenate([pick, rng.choice(pick, int(dup * len(pick)), replace=False)]) if len(pick) else pick
    fc, ray = pixel_to_ray(F, kx[pick] + rng.normal(0, 1.5, len(pick)), ky[pick] + rng.normal(0, 1.5, len(pick)))
    ray[fc < 0] = (0.0, 0.0, 1.0)                                       # jittered into a corner block of the cross
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    depth = rng.uniform(2.0, 12.0, len(pick))
    Xc = ray * depth[:, None]
    Xw = (Xc - t.astype(np.float64)) @ R.astype(np.float64)           # R^T (Xc - t)
    lvl = np.clip(koct[pick] + rng.integers(-1, 2, len(pick)), 0, nlevels - 1)
    max_d = depth * sf[lvl] * rng.uniform(0.97, 1.03, len(pick))
    min_d = max_d / sf[nlevels - 1]
    nrm = (Xw - Ow.astype(np.float64)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm += rng.normal(0, 0.25, nrm.shape); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    desc = np.asarray(kdesc, np.uint8)[pick].copy()
    flips = rng.integers(0, 256, (len(pick), 12))
    nflip = rng.integers(0, 13, len(pick))
    for j in range(12):
        m = nflip > j
        desc[m, flips[m, j] >> 3] ^= (1 << (flips[m, j] & 7)).astype(np.uint8)
    ne = int(extra * len(pick))
    Xe = rng.normal(0, 8.0, (ne, 3)) + Ow
    ne_n = rng.normal(0, 1, (ne, 3)); ne_n /= np.linalg.norm(ne_n, axis=1, keepdims=True)
    de = np.linalg.norm(Xe - Ow, axis=1)
    max_e = de * rng.uniform(0.3, 3.0, ne); min_e = max_e / sf[nlevels - 1]
    P = np.concatenate([Xw, Xe]).astype(np.float32); N = np.concatenate([nrm, ne_n]).astype(np.float32)
    mind = np.concatenate([min_d, min_e]).astype(np.float32); maxd = np.concatenate([max_d, max_e]).astype(np.float32)
    D = np.concatenate([desc, rng.integers(0, 256, (ne, 32), dtype=np.uint8)])
    if order == "spatial":
        key = np.concatenate([ky[pick] * 4096.0 + kx[pick], rng.uniform(0, 4096.0 * 3 * F, ne)])
        perm = np.argsort(key, kind="stable")
    else:
        perm = rng.permutation(len(P))
    pose15 = np.concatenate([R.reshape(-1), t, Ow]).astype(np.float32)
    return dict(pose15=pose15, pos=np.ascontiguousarray(P[perm]), normal=np.ascontiguousarray(N[perm]), min_dist=mind[perm].copy(),
                max_dist=maxd[perm].copy(), desc=np.ascontiguousarray(D[perm]), scale_factors=sf)


def motion_model_problem(F, kx, ky, koct, kangle, kdesc, seed=3, nlevels=8, scale=1.2, dup=0.2):
    """Inputs of ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono): the current frame's float pose (Rcw | tcw) and, per
    key point of a synthetic last frame, whether it holds a usable map point (`valid`), that point's world position (behind a current
    key point, a few pixels of noise; some behind the camera), descriptor (the current key point's + flipped bits), the last key point's
    octave (current +-1) and angle (current + a common rotation + noise; 12 % random, so the rotation histogram has something to reject).
    `dup` of the points are doubled (two map points competing for one key point)."""
    rng = np.random.default_rng(seed)
    kx = np.asarray(kx, np.float32); ky = np.asarray(ky, np.float32); koct = np.asarray(koct, np.int32); kangle = np.asarray(kangle, np.float32)
    n = len(kx)
    ang = rng.normal(0, 0.3, 3)
    R = (_rot((1, 0, 0), ang[0]) @ _rot((0, 1, 0), ang[1]) @ _rot((0, 0, 1), ang[2])).astype(np.float32)
    t = rng.normal(0, 1.0, 3).astype(np.float32)
    pick = np.flatnonzero(rng.random(n) < 0.75)
    pick = np.concatenate([pick, rng.choice(pick, int(dup * len(pick)), replace=False)])
    pick = pick[rng.permutation(len(pick))]
    m = len(pick)
    fc, ray = pixel_to_ray(F, kx[pick] + rng.normal(0, 2.0, m), ky[pick] + rng.normal(0, 2.0, m))
    ray[fc < 0] = (0.0, 0.0, 1.0)
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    depth = rng.uniform(2.0, 12.0, m)
    depth[rng.random(m) < 0.05] *= -1.0                                   # behind the camera
    Xw = ((ray * depth[:, None]) - t.astype(np.float64)) @ R.astype(np.float64)
    desc = np.asarray(kdesc, np.uint8)[pick].copy()
    flips = rng.integers(0, 256, (m, 14)); nflip = rng.integers(0, 15, m)
    for j in range(14):
        sel = nflip > j
        desc[sel, flips[sel, j] >> 3] ^= (1 << (flips[sel, j] & 7)).astype(np.uint8)
    octave = np.clip(koct[pick] + rng.integers(-1, 2, m), 0, nlevels - 1).astype(np.int32)
    angle = (kangle[pick] + np.float32(17.0) + rng.normal(0, 4.0, m)).astype(np.float32) % np.float32(360.0)
    wild = rng.random(m) < 0.12
    angle[wild] = rng.uniform(0, 360, wild.sum()).astype(np.float32)
    valid = (rng.random(m) < 0.85).astype(np.uint8)
    sf = np.float32(scale) ** np.arange(nlevels, dtype=np.float32)
    return dict(pose12=np.concatenate([R.reshape(-1), t]).astype(np.float32), valid=valid, Xw=Xw.astype(np.float32), octave=octave, angle=angle.astype(np.float32),
                desc=desc, scale_factors=sf)


def keyframe_set(F, n_kf=4, n_pts=1600, seed=5, nlevels=8, scale=1.2, node_size=6, with_mp=0.45):
    """A current key frame (index 0) and n_kf - 1 covisible neighbours looking at one random scene (LocalMapping::CreateNewMapPoints'
    inputs): per key frame float pose (Rcw, tcw, Ow), key points (projection + noise, random level), descriptors (the scene point's +
    a few flipped bits), `mp` (>= 0 where the feature already holds a map point) and a DBoW2-style FeatureVector: node id ->
    feature indices (scene points are binned into nodes of ~node_size; 8 % of the features land in a wrong node)."""
    rng = np.random.default_rng(seed)
    X = rng.normal(0, 1, (n_pts, 3)); X *= (rng.uniform(3.0, 9.0, n_pts) / np.linalg.norm(X, axis=1))[:, None]
    pdesc = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    pnode = (rng.permutation(n_pts) // node_size).astype(np.int32)
    has_mp = rng.random(n_pts) < with_mp
    sf = np.float32(scale) ** np.arange(nlevels, dtype=np.float32)
    kfs = []
    for k in range(n_kf):
        ang = rng.normal(0, 0.15, 3)
        R = (_rot((1, 0, 0), ang[0]) @ _rot((0, 1, 0), ang[1]) @ _rot((0, 0, 1), ang[2])).astype(np.float32)
        t = (rng.normal(0, 0.35, 3) if k else np.zeros(3)).astype(np.float32)
        Ow = (-(R.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32)
        Xc = X @ R.astype(np.float64).T + t
        face, up, vp = rays_to_cubemap(F, Xc)
        vis = np.flatnonzero((face >= 0) & (rng.random(n_pts) < 0.85))
        u = (up[vis] + rng.normal(0, 0.4, len(vis))).astype(np.float32); v = (vp[vis] + rng.normal(0, 0.4, len(vis))).astype(np.float32)
        ok = face_of_pixel(F, u.astype(np.float64), v.astype(np.float64)) == face[vis]
        vis, u, v = vis[ok], u[ok], v[ok]
        perm = rng.permutation(len(vis)); vis, u, v = vis[perm], u[perm], v[perm]
        n = len(vis)
        depth = np.linalg.norm(Xc[vis], axis=1)
        octave = np.clip(np.round(np.log(6.0 / depth) / np.log(scale) + 3 + rng.normal(0, 0.4, n)), 0, nlevels - 1).astype(np.int32)
        d = pdesc[vis].copy()
        flips = rng.integers(0, 256, (n, 10)); nflip = rng.integers(0, 11, n)
        for j in range(10):
            m = nflip > j
            d[m, flips[m, j] >> 3] ^= (1 << (flips[m, j] & 7)).astype(np.uint8)
        node = pnode[vis].copy()
        wrong = rng.random(n) < 0.08
        node[wrong] = rng.integers(0, pnode.max() + 1, wrong.sum())
        order = np.lexsort((np.arange(n), node))
        ids, starts = np.unique(node[order], return_index=True)
        mp = np.where(has_mp[vis] & (rng.random(n) < 0.9), vis, -1).astype(np.int32)
        kfs.append(dict(R=R, t=t, Ow=Ow, x=u, y=v, octave=octave, angle=rng.uniform(0, 360, n).astype(np.float32), desc=d, point=vis, mp=mp,
                        node_id=ids.astype(np.int32), node_off=np.concatenate([starts, [n]]).astype(np.int32), node_feat=order.astype(np.int32),
                        median_depth=np.float32(np.median(Xc[vis, 2]) if n else 1.0)))
    return dict(kfs=kfs, X=X, scale_factors=sf, level_sigma2=(sf * sf).astype(np.float32), inv_level_sigma2=(np.float32(1.0) / (sf * sf)).astype(np.float32))


def descriptors(n, seed):
    return np.random.RandomState(seed).randint(0, 256, size=(n, 32)).astype(np.uint8)


def candidate_lists(nq, nt, mean_cand, seed):
    rs = np.random.RandomState(seed)
    counts = rs.poisson(mean_cand, nq).astype(np.int64)
    counts[rs.uniform(size=nq) < 0.05] = 0  # some empty lists
    off = np.zeros(nq + 1, np.int32)
    off[1:] = np.cumsum(counts)
    idx = rs.randint(0, nt, size=int(off[-1])).astype(np.int32)
    return off, idx


# ---------------------------------------------------------------------------------------------------------------------------------
# Ray-cast box-room renderer: a 6 x 3 x 4 m room with a seeded procedural texture on every wall, seen by
# the fisheye camera model from a smooth trajectory.  Every pixel's ray comes from CamModelGeneral::ImgToWorld (include/CamModelGeneral.h:
# 262-280), is intersected with the six walls and samples the wall texture bilinearly.  Ground truth (poses, the 3-D point behind any
# cubemap pixel) comes with it: the closed-loop harness seeds its map points from it instead of running the Initializer / triangulation.
def img_to_world(cam, u, v):
    """CamModelGeneral::ImgToWorld: unit ray (camera frame, z forward) through fisheye pixel (u, v)."""
    inv_aff = cam["c"] - cam["d"] * cam["e"]
    ut = u - cam["u0"]; vt = v - cam["v0"]
    x = (ut - cam["d"] * vt) / inv_aff
    y = (-cam["e"] * ut + cam["c"] * vt) / inv_aff
    z = -_horner(list(cam["pol"]), np.sqrt(x * x + y * y))
    n = np.sqrt(x * x + y * y + z * z)
    return np.stack([x / n, y / n, z / n], -1)


ROOM_HALF = np.array([3.0, 1.5, 2.0])          # x (right), y (down), z (forward)
_ROOM_PPM = 150.0                              # wall texture resolution, pixels per metre


def room_scene(seed=0xC0FFEE):
    """wall textures of the box room: for axis a and side s the wall spans the two other axes"""
    walls = {}
    for a in range(3):
        o = [i for i in range(3) if i != a]
        h = int(2 * ROOM_HALF[o[1]] * _ROOM_PPM) + 2; w = int(2 * ROOM_HALF[o[0]] * _ROOM_PPM) + 2
        for s in (0, 1):
            walls[(a, s)] = texture(h, w, (seed + 17 * a + 5 * s) & 0x7FFFFFFF).astype(np.float32)
    return dict(walls=walls, seed=seed)


def room_raycast(origin, dirs):
    """first wall hit of rays origin + t * dirs (origin inside the room): (points, axis, side, t)"""
    o = np.asarray(origin, np.float64)
    d = np.asarray(dirs, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        bound = np.where(d > 0, ROOM_HALF, -ROOM_HALF)
        t = np.where(d != 0, (bound - o) / d, np.inf)
    axis = np.argmin(t, -1)
    tt = np.take_along_axis(t, axis[..., None], -1)[..., 0]
    P = o + d * tt[..., None]
    side = (np.take_along_axis(d, axis[..., None], -1)[..., 0] > 0).astype(np.int64)
    return P, axis, side, tt


def room_pose(i, n=300):
    """camera i of a smooth n-frame loop inside the room: (Rcw, tcw) world -> camera, float64"""
    a = 2 * np.pi * i / n
    cw = np.array([1.2 * np.cos(a), 0.15 * np.sin(2 * a), 0.7 * np.sin(a)])
    yaw = a + np.pi / 2 + 0.15 * np.sin(3 * a)             # looking along the direction of travel, swaying a little
    Rwc = _rot([0, 1, 0], -yaw) @ _rot([1, 0, 0], 0.08 * np.sin(2 * a)) @ _rot([0, 0, 1], 0.05 * np.sin(a))
    Rcw = Rwc.T
    return Rcw, -Rcw @ cw


def render_fisheye(cam, scene, Rcw, tcw):
    """the fisheye image (Ih x Iw, uint8) the camera at pose (Rcw, tcw) sees in the room"""
    Ih, Iw = cam["Ih"], cam["Iw"]
    vv, uu = np.meshgrid(np.arange(Ih, dtype=np.float64), np.arange(Iw, dtype=np.float64), indexing="ij")
    rays_c = img_to_world(cam, uu, vv)
    Rwc = np.asarray(Rcw, np.float64).T
    cw = -Rwc @ np.asarray(tcw, np.float64)
    rays_w = rays_c @ Rwc.T
    P, axis, side, _ = room_raycast(cw, rays_w)
    img = np.zeros((Ih, Iw), np.float32)
    for a in range(3):
        o = [i for i in range(3) if i != a]
        for s in (0, 1):
            m = (axis == a) & (side == s)
            if not m.any():
                continue
            tex = scene["walls"][(a, s)]
            x = (P[m][:, o[0]] + ROOM_HALF[o[0]]) * _ROOM_PPM; y = (P[m][:, o[1]] + ROOM_HALF[o[1]]) * _ROOM_PPM
            x = np.clip(x, 0, tex.shape[1] - 1.001); y = np.clip(y, 0, tex.shape[0] - 1.001)
            x0 = x.astype(np.int64); y0 = y.astype(np.int64); fx = (x - x0).astype(np.float32); fy = (y - y0).astype(np.float32)
            img[m] = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def room_points_behind_pixels(F, Rcw, tcw, px, py):
    """ground truth: the room point seen at cubemap pixel (px, py) from pose (Rcw, tcw) -> (valid, Xw float64)"""
    face, ray = pixel_to_ray(F, px, py)
    Rwc = np.asarray(Rcw, np.float64).T
    cw = -Rwc @ np.asarray(tcw, np.float64)
    ok = face >= 0
    ray = np.where(ok[:, None], ray, np.array([0.0, 0.0, 1.0]))
    P, _, _, _ = room_raycast(cw, ray @ Rwc.T)
    return ok, P
