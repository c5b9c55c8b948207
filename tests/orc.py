"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


class Camera(C.Structure):
    _fields_ = [("c", C.c_double), ("d", C.c_double), ("e", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("invpol", C.c_double * 12), ("pol", C.c_double * 5), ("Iw", C.c_int), ("Ih", C.c_int),
                ("face", C.c_int), ("fov_deg", C.c_double)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4")])


class BaStats(C.Structure):
    _fields_ = [("iterations_done", C.c_int * 2), ("chi2_initial", C.c_double * 2), ("chi2_final", C.c_double * 2),
                ("lambda_final", C.c_double * 2), ("n_outliers_mid", C.c_int), ("n_outliers_final", C.c_int)]


def build(force=False):
    so = os.path.join(_ORC_DIR, "liborc.so")
    srcs = [os.path.join(_ORC_DIR, f) for f in os.listdir(_ORC_DIR) if f.endswith((".cpp", ".h", ".inc"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _ORC_DIR, "-s"])
    return so


_lib = None


_lib_lock = __import__("threading").Lock()


def lib():
    """the oracle library with every prototype set; published only once it is complete (bench.py's first use is eight threads at once: a thread that found
    the handle before its prototypes were set passed a Python float to an untyped function)"""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is None:
            _lib = _load()
    return _lib


def _load():
    if True:
        L = C.CDLL(build())
        L.orc_orb_create.restype = C.c_void_p
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cos_fov_th.restype = C.c_float
        L.orc_cv_round.argtypes = [C.c_double]
        for name in ("orc_world_to_img",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_img_to_world.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_cubemap_to_fisheye.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_face_in_cubemap.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_rays_to_cubemap.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_rays_to_target_face.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_cubemap_to_rays.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_ba_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ba_linearize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_int, C.c_double] + [C.c_void_p] * 10
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int]
        L.orc_pose_optimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int]
        for n in ("orc_orb_destroy", "orc_orb_nlevels"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_orb_tables.argtypes = [C.c_void_p] * 7
        L.orc_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_orb_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_level_distributed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_camera(d):
    cam = Camera()
    for k in ("c", "d", "e", "u0", "v0", "Iw", "Ih", "face", "fov_deg"):
        setattr(cam, k, d[k])
    for i in range(12):
        cam.invpol[i] = d["invpol"][i] if i < len(d["invpol"]) else 0.0
    for i in range(5):
        cam.pol[i] = d["pol"][i] if i < len(d["pol"]) else 0.0
    return cam


def build_lut(cam):
    W = 3 * cam.face
    m1 = np.zeros((W, W), np.float32)
    m2 = np.zeros((W, W), np.float32)
    lib().orc_build_lut(C.byref(cam), _p(m1), _p(m2))
    return m1, m2


def fisheye_to_cubemap(cam, m1, m2, fisheye):
    W = 3 * cam.face
    out = np.zeros((W, W), np.uint8)
    fisheye = np.ascontiguousarray(fisheye)
    lib().orc_fisheye_to_cubemap(C.byref(cam), _p(m1), _p(m2), _p(fisheye), C.c_int(fisheye.strides[0]), _p(out), C.c_int(W))
    return out


def remap(src, m1, m2):
    src = np.ascontiguousarray(src)
    m1 = np.ascontiguousarray(m1, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    dst = np.zeros(m1.shape, np.uint8)
    lib().orc_remap_bilinear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(m1), _p(m2), m1.shape[1], _p(dst),
                             m1.shape[1], m1.shape[0], dst.strides[0])
    return dst


def resize(src, dw, dh):
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def blur7(src, column_mode=0):
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    if column_mode:
        f = lib().orc_gaussian_blur7_mode
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        f(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0], int(column_mode))
        return dst
    lib().orc_gaussian_blur7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast(img, threshold):
    img = np.ascontiguousarray(img)
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, _p(out), cap)
    return out[:n].copy()


def distribute_octree(xys, min_x, max_x, min_y, max_y, N):
    xys = np.ascontiguousarray(xys, np.int32)
    out = np.zeros((max(N + 8, 8), 3), np.int32)
    n = lib().orc_distribute_octree(_p(xys), len(xys), min_x, max_x, min_y, max_y, N, _p(out), len(out))
    assert n <= len(out)
    return out[:n].copy()


class Orb:
    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, gaussian_column_mode=0):
        self.params = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.h = C.c_void_p(lib().orc_orb_create(C.byref(self.params)))
        if gaussian_column_mode:
            f = lib().orc_orb_set_gaussian_mode
            f.argtypes = [C.c_void_p, C.c_int]
            f(self.h, int(gaussian_column_mode))
        self.nlevels = nlevels

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_orb_destroy(self.h)
            self.h = None

    def tables(self):
        L = self.nlevels
        sc, isc, s2, is2 = (np.zeros(L, np.float32) for _ in range(4))
        q = np.zeros(L, np.int32)
        um = np.zeros(16, np.int32)
        lib().orc_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(q), _p(um))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=q, umax=um)

    def extract(self, cam, image, mask, cap=None):
        image = np.ascontiguousarray(image)
        mask = np.ascontiguousarray(mask)
        cap = cap or 4 * self.params.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_orb_extract(self.h, C.byref(cam), _p(image), image.shape[1], image.shape[0], image.strides[0],
                                  _p(mask), mask.strides[0], _p(kps), _p(desc), cap)
        assert 0 <= n <= cap, n
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l):
        w, h = C.c_int(), C.c_int()
        lib().orc_orb_level_size(self.h, l, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        lib().orc_orb_level_copy(self.h, l, _p(out), w.value)
        return out

    def candidates(self, l):
        n = lib().orc_orb_level_candidates(self.h, l, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        lib().orc_orb_level_candidates(self.h, l, _p(out), n)
        return out[:n]

    def distributed(self, l):
        n = lib().orc_orb_level_distributed(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().orc_orb_level_distributed(self.h, l, _p(out), n)
        return out[:n]


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_best2(qdesc, tdesc, cand_off, cand_idx, tlevel=None):
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    tdesc = np.ascontiguousarray(tdesc, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32)
    cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    nq = len(qdesc)
    outs = [np.zeros(nq, np.int32) for _ in range(5)]
    tl = np.ascontiguousarray(tlevel, np.int32) if tlevel is not None else None
    lib().orc_hamming_best2(_p(qdesc), nq, _p(tdesc), _p(cand_off), _p(cand_idx), _p(tl), *[_p(o) for o in outs])
    return dict(best_idx=outs[0], best_dist=outs[1], best_level=outs[2], second_dist=outs[3], second_level=outs[4])


def hamming_matrix(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.zeros((len(a), len(b)), np.uint16)
    lib().orc_hamming_matrix(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def ba_run(prob, its=(5, 10), stop=None):
    poses = np.array(prob["poses"], np.float64, copy=True)
    pts = np.array(prob["points"], np.float64, copy=True)
    E = len(prob["e_pose"])
    flags = np.zeros(E, np.uint8)
    st = BaStats()
    stop_arr = np.array([1 if stop else 0], np.uint8)
    rc = lib().orc_ba_run(len(poses), _p(poses), _p(prob["fixed"]), len(pts), _p(pts), E, _p(prob["e_pose"]),
                          _p(prob["e_point"]), _p(prob["e_obs"]), _p(prob["e_invsig2"]), _p(prob["e_face"]),
                          prob["fx"], prob["fy"], prob["cx"], prob["cy"], its[0], its[1], _p(stop_arr), _p(flags), C.byref(st))
    return dict(rc=rc, poses=poses, points=pts, outliers=flags, stats=st)


def ba_run_stop_after(prob, stop_after_trials, its=(5, 10)):
    """ba_run with the stop flag raised while trial number `stop_after_trials` (1-based, counted over both stages) is running"""
    poses = np.array(prob["poses"], np.float64, copy=True)
    pts = np.array(prob["points"], np.float64, copy=True)
    E = len(prob["e_pose"])
    flags = np.zeros(E, np.uint8)
    st = BaStats()
    f = lib().orc_ba_run_stop_after
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_double] * 4 + [C.c_int] * 3 + [C.c_void_p] * 2
    rc = f(len(poses), _p(poses), _p(prob["fixed"]), len(pts), _p(pts), E, _p(prob["e_pose"]), _p(prob["e_point"]), _p(prob["e_obs"]),
           _p(prob["e_invsig2"]), _p(prob["e_face"]), prob["fx"], prob["fy"], prob["cx"], prob["cy"], its[0], its[1], int(stop_after_trials),
           _p(flags), C.byref(st))
    return dict(rc=rc, poses=poses, points=pts, outliers=flags, stats=st)


class PoseStats(C.Structure):
    _fields_ = [("rounds", C.c_int), ("n_bad", C.c_int), ("iterations_done", C.c_int * 4), ("chi2_final", C.c_double * 4)]


def pose_optimize(prob, n=None):
    """Optimizer::PoseOptimization on synth.pose_problem-style arrays; returns (n_inliers, pose7, outlier flags, stats)."""
    n = len(prob["Xw"]) if n is None else n
    pose = np.array(prob["pose0"], np.float64, copy=True)
    out = np.zeros(max(n, 1), np.uint8)
    st = PoseStats()
    Xw = np.ascontiguousarray(prob["Xw"][:n], np.float64); obs = np.ascontiguousarray(prob["obs"][:n], np.float64)
    inv = np.ascontiguousarray(prob["invsig2"][:n], np.float64); face = np.ascontiguousarray(prob["face"][:n], np.int8)
    r = lib().orc_pose_optimize(n, _p(Xw), _p(obs), _p(inv), _p(face), prob["fx"], prob["fy"], prob["cx"], prob["cy"],
                                _p(pose), _p(out), C.byref(st))
    return r, pose, out[:n], st


def ba_linearize(prob, robust=True, delta=np.sqrt(5.991)):
    K, P, E = len(prob["poses"]), len(prob["points"]), len(prob["e_pose"])
    poses = np.ascontiguousarray(prob["poses"], np.float64)
    pts = np.ascontiguousarray(prob["points"], np.float64)
    o = dict(err=np.zeros((E, 2)), chi2=np.zeros(E), Jpose=np.zeros((E, 2, 6)), Jpoint=np.zeros((E, 2, 3)),
             Hpp=np.zeros((K, 6, 6)), bp=np.zeros((K, 6)), Hll=np.zeros((P, 3, 3)), bl=np.zeros((P, 3)),
             Hpl=np.zeros((E, 6, 3)), chi=np.zeros(1))
    lib().orc_ba_linearize(K, _p(poses), _p(prob["fixed"]), P, _p(pts), E, _p(prob["e_pose"]), _p(prob["e_point"]),
                           _p(prob["e_obs"]), _p(prob["e_invsig2"]), _p(prob["e_face"]), prob["fx"], prob["fy"],
                           prob["cx"], prob["cy"], 1 if robust else 0, float(delta), _p(o["err"]), _p(o["chi2"]),
                           _p(o["Jpose"]), _p(o["Jpoint"]), _p(o["Hpp"]), _p(o["bp"]), _p(o["Hll"]), _p(o["bl"]),
                           _p(o["Hpl"]), _p(o["chi"]))
    return o


def features_in_area(cam, kx, ky, koct, qx, qy, qr, qmin, qmax, cap=None):
    """Frame::GetFeaturesInArea for a batch of queries; returns (off[nq+1], idx[total]) in the reference's candidate order."""
    kx = np.ascontiguousarray(kx, np.float32); ky = np.ascontiguousarray(ky, np.float32); koct = np.ascontiguousarray(koct, np.int32)
    qx = np.ascontiguousarray(qx, np.float32); qy = np.ascontiguousarray(qy, np.float32); qr = np.ascontiguousarray(qr, np.float32)
    qmin = np.ascontiguousarray(qmin, np.int32); qmax = np.ascontiguousarray(qmax, np.int32)
    nq = len(qx)
    cap = cap or max(1, 64 * nq + 1024)
    off = np.zeros(nq + 1, np.int32); idx = np.zeros(cap, np.int32)
    tot = lib().orc_features_in_area(C.byref(cam), len(kx), _p(kx), _p(ky), _p(koct), nq, _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax),
                                     _p(off), _p(idx), cap)
    assert tot <= cap, (tot, cap)
    return off, idx[:tot]


def is_in_frustum(cam, pose15, pos, normal, min_dist, max_dist, viewing_cos_limit=0.5, scale_factor=1.2, nlevels=8):
    """Frame::isInFrustum over a list of map points -> dict(in_view, proj_x, proj_y, level, view_cos)"""
    L = lib()
    L.orc_is_in_frustum.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 5
    pose15 = np.ascontiguousarray(pose15, np.float32).reshape(15)
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3); normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
    min_dist = np.ascontiguousarray(min_dist, np.float32); max_dist = np.ascontiguousarray(max_dist, np.float32)
    n = len(pos)
    vis = np.zeros(n, np.uint8); px = np.zeros(n, np.float32); py = np.zeros(n, np.float32); lvl = np.zeros(n, np.int32); vc = np.zeros(n, np.float32)
    R, t, O = pose15[:9].copy(), pose15[9:12].copy(), pose15[12:15].copy()
    L.orc_is_in_frustum(C.byref(cam), _p(R), _p(t), _p(O), n, _p(pos), _p(normal), _p(min_dist), _p(max_dist), viewing_cos_limit,
                        scale_factor, nlevels, _p(vis), _p(px), _p(py), _p(lvl), _p(vc))
    return dict(in_view=vis, proj_x=px, proj_y=py, level=lvl, view_cos=vc)


def search_local_points(cam, kx, ky, koct, kdesc, scale_factors, fr, mp_desc, kp_mp, th=1.0, nnratio=0.8, th_high=100):
    """ORBMatcher::SearchByProjection(F, vpMapPoints, th); fr = is_in_frustum's dict; kp_mp int32 in/out.  Returns (match, nmatches)."""
    L = lib()
    L.orc_search_local_points.argtypes = ([C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_int] +
                                          [C.c_void_p] * 2)
    kx = np.ascontiguousarray(kx, np.float32); ky = np.ascontiguousarray(ky, np.float32); koct = np.ascontiguousarray(koct, np.int32)
    kdesc = np.ascontiguousarray(kdesc, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    n = len(fr["in_view"])
    assert kp_mp.dtype == np.int32 and len(kp_mp) == len(kx)
    match = np.full(n, -1, np.int32)
    nm = L.orc_search_local_points(C.byref(cam), len(kx), _p(kx), _p(ky), _p(koct), _p(kdesc), _p(sf), n, _p(fr["in_view"]), _p(fr["proj_x"]),
                                   _p(fr["proj_y"]), _p(fr["level"]), _p(fr["view_cos"]), _p(mp_desc), th, nnratio, th_high, _p(kp_mp), _p(match))
    return match, nm


class Keyframe(C.Structure):
    _fields_ = [("n", C.c_int), ("kps", C.c_void_p), ("desc", C.c_void_p), ("rays", C.c_void_p), ("mp", C.c_void_p),
                ("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
                ("nnodes", C.c_int), ("node_id", C.c_void_p), ("node_off", C.c_void_p), ("node_feat", C.c_void_p), ("median_depth", C.c_float)]


def keyframe_rays(cam, x, y):
    """mvKeyRays: CamModelGeneral::TransformCubemapToRays of every key point"""
    L = lib()
    L.orc_cubemap_to_rays.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    out = np.zeros((len(x), 3), np.float32); r = (C.c_float * 3)()
    for i in range(len(x)):
        L.orc_cubemap_to_rays(C.byref(cam), float(x[i]), float(y[i]), r)
        out[i] = r[:]
    return out


def make_keyframe(cam, kf):
    """kf: one entry of synth.keyframe_set(...)['kfs']; returns (Keyframe struct, keep-alive list)"""
    n = len(kf["x"])
    kps = np.zeros(n, KP_DTYPE); kps["x"] = kf["x"]; kps["y"] = kf["y"]; kps["octave"] = kf["octave"]; kps["angle"] = kf["angle"]
    if "rays" not in kf:
        kf["rays"] = keyframe_rays(cam, kf["x"], kf["y"])
    keep = [kps, np.ascontiguousarray(kf["desc"]), np.ascontiguousarray(kf["rays"], np.float32), np.ascontiguousarray(kf["mp"], np.int32),
            np.ascontiguousarray(kf["node_id"], np.int32), np.ascontiguousarray(kf["node_off"], np.int32), np.ascontiguousarray(kf["node_feat"], np.int32)]
    K = Keyframe()
    K.n = n; K.kps = keep[0].ctypes.data; K.desc = keep[1].ctypes.data; K.rays = keep[2].ctypes.data; K.mp = keep[3].ctypes.data
    K.Rcw[:] = [float(v) for v in np.asarray(kf["R"], np.float32).reshape(9)]; K.tcw[:] = [float(v) for v in kf["t"]]; K.Ow[:] = [float(v) for v in kf["Ow"]]
    K.nnodes = len(keep[4]); K.node_id = keep[4].ctypes.data; K.node_off = keep[5].ctypes.data; K.node_feat = keep[6].ctypes.data
    K.median_depth = float(kf["median_depth"])
    return K, keep


def compute_e12(k1, k2):
    L = lib()
    E = np.zeros(9, np.float32)
    a = [np.ascontiguousarray(v, np.float32) for v in (k1["R"], k1["t"], k2["R"], k2["t"])]
    L.orc_compute_e12.argtypes = [C.c_void_p] * 5
    L.orc_compute_e12(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(E))
    return E


def search_for_triangulation(cam, K1, K2, E12, sf, sigma2, check_ori=False):
    L = lib()
    L.orc_search_for_triangulation.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
    m = np.full(K1.n, -1, np.int32)
    n = L.orc_search_for_triangulation(C.byref(cam), C.byref(K1), C.byref(K2), _p(E12), _p(sf), _p(sigma2), int(check_ori), _p(m))
    return m, n


def create_new_map_points(cam, K1, Kn, sf, sigma2, cur_mp):
    """Kn: list of Keyframe structs (neighbours, covisibility order); cur_mp int32 in/out -> (neigh, idx1, idx2, x3d)"""
    L = lib()
    L.orc_create_new_map_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int]
    arr = (Keyframe * len(Kn))(*Kn)
    cap = K1.n * max(len(Kn), 1) + 1
    on = np.zeros(cap, np.int32); o1 = np.zeros(cap, np.int32); o2 = np.zeros(cap, np.int32); ox = np.zeros((cap, 3), np.float32)
    n = L.orc_create_new_map_points(C.byref(cam), C.byref(K1), len(Kn), arr, _p(sf), _p(sigma2), _p(cur_mp), _p(on), _p(o1), _p(o2), _p(ox), cap)
    return on[:n], o1[:n], o2[:n], ox[:n]


def fuse_search(cam, K, skip, pos, normal, min_dist, max_dist, desc, th, sf, inv_sigma2):
    L = lib()
    L.orc_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = len(pos)
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
    a = [np.ascontiguousarray(skip, np.uint8), np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(normal, np.float32),
         np.ascontiguousarray(min_dist, np.float32), np.ascontiguousarray(max_dist, np.float32), np.ascontiguousarray(desc, np.uint8)]
    L.orc_fuse_search(C.byref(cam), C.byref(K), n, *[_p(v) for v in a], th, _p(sf), _p(inv_sigma2), len(sf), _p(bi), _p(bd))
    return bi, bd


def distinctive_descriptors(obs_off, desc):
    L = lib()
    L.orc_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    obs_off = np.ascontiguousarray(obs_off, np.int32); desc = np.ascontiguousarray(desc, np.uint8)
    out = np.zeros(len(obs_off) - 1, np.int32)
    L.orc_distinctive_descriptors(len(out), _p(obs_off), _p(desc), _p(out))
    return out


def update_normal_and_depth(obs_off, pos, obs_Ow, ref_Ow, ref_level, sf):
    L = lib()
    L.orc_update_normal_and_depth.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3
    a = [np.ascontiguousarray(obs_off, np.int32), np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(obs_Ow, np.float32),
         np.ascontiguousarray(ref_Ow, np.float32), np.ascontiguousarray(ref_level, np.int32), np.ascontiguousarray(sf, np.float32)]
    n = len(a[0]) - 1
    nrm = np.zeros((n, 3), np.float32); mn = np.zeros(n, np.float32); mx = np.zeros(n, np.float32)
    L.orc_update_normal_and_depth(n, *[_p(v) for v in a], len(a[5]), _p(nrm), _p(mn), _p(mx))
    return nrm, mn, mx


def search_by_projection_frames(cam, Rcw, tcw, kx, ky, koct, kangle, kdesc, sf, valid, Xw, loct, langle, mp_desc, kp_mp, th=15.0, check_ori=True, th_high=100):
    L = lib()
    L.orc_search_by_projection_frames.argtypes = ([C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int] +
                                                  [C.c_void_p] * 2)
    f32 = lambda a: np.ascontiguousarray(a, np.float32); i32 = lambda a: np.ascontiguousarray(a, np.int32); u8 = lambda a: np.ascontiguousarray(a, np.uint8)
    a = [f32(Rcw), f32(tcw)]; k = [f32(kx), f32(ky), i32(koct), f32(kangle), u8(kdesc), f32(sf)]
    l = [u8(valid), f32(Xw), i32(loct), f32(langle), u8(mp_desc)]
    match = np.full(len(l[0]), -1, np.int32)
    assert kp_mp.dtype == np.int32
    n = L.orc_search_by_projection_frames(C.byref(cam), _p(a[0]), _p(a[1]), len(k[0]), *[_p(v) for v in k], len(l[0]), *[_p(v) for v in l], th, int(check_ori),
                                          th_high, _p(kp_mp), _p(match))
    return match, n


def search_for_initialization(cam, k1, d1, k2, d2, prev_matched, window=100, nnratio=0.9, check_ori=True):
    """ORBMatcher::SearchForInitialization (ORBMatcher.cpp:676-794); k1 / k2: key-point records (x, y, octave, angle), prev_matched (n1, 2) float32
    updated in place -> (matches12, nmatches)"""
    L = lib()
    L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
    f32 = lambda a: np.ascontiguousarray(a, np.float32); i32 = lambda a: np.ascontiguousarray(a, np.int32); u8 = lambda a: np.ascontiguousarray(a, np.uint8)
    a1 = [f32(k1["x"]), f32(k1["y"]), i32(k1["octave"]), f32(k1["angle"]), u8(d1)]
    a2 = [f32(k2["x"]), f32(k2["y"]), i32(k2["octave"]), f32(k2["angle"]), u8(d2)]
    assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(k1), 2)
    m12 = np.full(len(k1), -1, np.int32)
    n = L.orc_search_for_initialization(C.byref(cam), len(k1), *[_p(v) for v in a1], len(k2), *[_p(v) for v in a2], _p(prev_matched), int(window), float(nnratio),
                                        int(check_ori), _p(m12))
    return m12, n
