"""Closed-loop "extract + match + track" harness of BASELINE.json configs[2].

One camera stream through the hot path in the order the reference's Tracking thread runs it (src/Tracking.cpp):

    frame 0 / 1   3 x nFeatures extractor (Tracking.cpp:95-96, 145-148); MonocularInitialization (:391-465): the first frame needs > 100 key
                  points, then ORBMatcher(0.9, true).SearchForInitialization(F0, F1, window 100) needs >= 100 matches.
                  The Initializer (two-view RANSAC) + GlobalBA that follow in the reference are out of scope: the matched key
                  points' 3-D positions and both poses are seeded from the renderer's ground truth instead.
    frame t >= 2  TrackWithMotionModel (:620-677): pose = velocity * last pose; SearchByProjection(Cur, Last, th = 15, then 30 if < 20
                  matches); PoseOptimization; outliers dropped.  TrackLocalMap (:679-719, 794-843): isInFrustum + SearchByProjection(F,
                  local map points, th = 1) over the points not matched yet; PoseOptimization again; the frame is tracked with >= 30
                  inliers.  velocity = Tcw * Twc_last (:360-368).
    every k frames a key frame: new map points behind unmatched key points (ground-truth seeded in place of CreateNewMapPoints'
                  triangulation, whose BoW pairing needs the vocabulary), then Optimizer::LocalBundleAdjustment over the window of the last
                  key frames and every map point they see, poses and points written back through float like Converter::toCvMat.

Every numeric step goes through a `backend`: GpuBackend below is the product (C-ABI of libcubemapslam_hip.so through api.py); the
tests plug the CPU oracle in behind the same interface and compare the two runs frame by frame (match lists, inliers, BA iterations).
All glue arithmetic (pose composition, problem assembly) lives here and is shared, so a difference can only come from a backend.
"""
import time

import numpy as np

from . import synth

TH_HIGH = 100


def _tcw(R, t):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.asarray(R, np.float32); T[:3, 3] = np.asarray(t, np.float32)
    return T


def _pose7_from_T(T):
    """float cv::Mat pose -> (t, q) doubles like Converter::toSE3Quat (Converter.cpp:41-51)"""
    R = T[:3, :3].astype(np.float64)
    return np.concatenate([T[:3, 3].astype(np.float64), synth._quat_from_R(R)])


def _T_from_pose7(p):
    """SE3Quat -> float cv::Mat (Converter::toCvMat, Converter.cpp:53-104)"""
    x, y, z, w = p[3:] / np.linalg.norm(p[3:])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return _tcw(R.astype(np.float32), p[:3].astype(np.float32))


class GpuBackend:
    """the product: every call is a C-ABI entry of libcubemapslam_hip.so"""
    name = "hip"

    def __init__(self, camd, mask, device=0, gaussian_mode=0):
        """gaussian_mode: cms_set_gaussian_mode -- 0 the integer definition of the 8-bit GaussianBlur, 1 what an x86 (SSE2) build of OpenCV <= 3.2
        computes (the definition the reference's maintainers get: integration/CubemapHipBridge.cpp selects it)"""
        from . import api
        self.api = api
        self.camd = camd
        nf = camd["nfeatures"]
        self.ctx_ini = api.Context(camd, nfeatures=3 * nf, max_batch=1, device=device)
        self.ctx_trk = api.Context(camd, nfeatures=nf, max_batch=1, device=device)
        for c in (self.ctx_ini, self.ctx_trk):
            c.set_mask(mask)
            if gaussian_mode:
                c.set_gaussian_mode(gaussian_mode)
        self.cur = None
        self.device = device

    def scale_factors(self):
        g = self.ctx_trk.geom
        return np.array([g.scale[l] for l in range(g.nlevels)], np.float32), np.array([g.inv_sigma2[l] for l in range(g.nlevels)], np.float32)

    def extract(self, fisheye, init):
        """System::CvtFisheyeToCubeMap + ORBextractor::operator(); the frame stays on the device as the current frame"""
        self.cur = self.ctx_ini if init else self.ctx_trk
        k, d, self.last_rays = self.cur.remap_extract_rays(fisheye)      # + Frame::ComputeKeyPointRays (Frame.cpp:746-760) from the device
        self.cur.area_grid(1)                                  # Frame::AssignFeaturesToGrid
        return k, d

    def search_for_initialization(self, k1, d1, k2, d2, prev_matched):
        return self.cur.search_for_initialization(0, k1, d1, prev_matched, 100, 0.9, True)

    def search_by_projection(self, kcur, dcur, pose12, valid, Xw, octave, angle, mp_desc, kp_mp, th):
        return self.cur.search_by_projection(0, pose12, valid, Xw, octave, angle, mp_desc, kp_mp, th=th, check_ori=True, th_high=TH_HIGH)

    def search_local_points(self, kcur, dcur, pose15, pos, normal, min_dist, max_dist, mp_desc, kp_mp, th):
        r = self.cur.search_local_points(0, pose15, pos, normal, min_dist, max_dist, mp_desc, kp_mp, viewing_cos_limit=0.5, th=th, nnratio=0.8, th_high=TH_HIGH)
        return r["match"], r["n_matches"], r["in_view"]

    def pose_optimize(self, prob):
        n, pose, out, st = self.api.pose_optimize(prob, device=self.device)
        return n, pose, out

    def local_ba(self, prob):
        r = self.api.ba_run(prob, device=self.device)
        return r["poses"], r["points"], r["outliers"], list(r["stats"].iterations_done)

    def close(self):
        self.ctx_ini.close(); self.ctx_trk.close()


class Tracker:
    def __init__(self, camd, backend, kf_every=5, ba_window=8, new_points_per_kf=400):
        self.camd, self.be = camd, backend
        self.F = camd["face"]
        self.sf, self.inv_sigma2 = backend.scale_factors()
        self.kf_every, self.ba_window, self.new_pts = kf_every, ba_window, new_points_per_kf
        self.cos_fov = np.cos(np.float32(camd["fov_deg"]) / 2 * (np.float32(np.pi) / 180))
        # map: parallel arrays
        self.mp_pos = np.zeros((0, 3), np.float32); self.mp_desc = np.zeros((0, 32), np.uint8); self.mp_normal = np.zeros((0, 3), np.float32)
        self.mp_min = np.zeros(0, np.float32); self.mp_max = np.zeros(0, np.float32)
        self.kfs = []            # key frames: dict(T, kps, kp_mp)
        self.state = "no_images"
        self.log = []
        self.velocity = None
        self.last = None         # dict(T, kps, desc, kp_mp, outlier)
        self.ini = None

    # ---- map points
    def _add_points(self, Xw, desc, octave, Ow):
        """new MapPoints with one observation: MapPoint::UpdateNormalAndDepth (MapPoint.cpp:332-373) for n = 1"""
        Xw = np.asarray(Xw, np.float32)
        PO = Xw - np.asarray(Ow, np.float32)
        dist = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(np.float32)
        base = len(self.mp_pos)
        self.mp_pos = np.concatenate([self.mp_pos, Xw]); self.mp_desc = np.concatenate([self.mp_desc, np.asarray(desc, np.uint8)])
        self.mp_normal = np.concatenate([self.mp_normal, (PO / dist[:, None]).astype(np.float32)])
        mx = (dist * self.sf[octave]).astype(np.float32)
        self.mp_max = np.concatenate([self.mp_max, mx]); self.mp_min = np.concatenate([self.mp_min, (mx / self.sf[-1]).astype(np.float32)])
        return np.arange(base, base + len(Xw), dtype=np.int32)

    def _pose_problem(self, T, kps, kp_mp, rays):
        """Optimizer::PoseOptimization's edges (Optimizer.cpp:78-129): key points holding a map point whose ray is inside the field of view.
        rays: the frame's mvKeyRays as the backend's extraction delivered them (unit vectors, CamModelGeneral.h:494-513)"""
        F = self.F
        idx = np.flatnonzero(kp_mp >= 0)
        idx = idx[rays[idx, 2] >= self.cos_fov]
        px = kps["x"][idx].astype(np.float64); py = kps["y"][idx].astype(np.float64)
        face = synth.face_of_pixel(F, px, py)
        idx = idx[face >= 0]; px = px[face >= 0]; py = py[face >= 0]; face = face[face >= 0]
        obs = np.stack([px - np.floor(px / F) * F, py - np.floor(py / F) * F], 1)
        return idx, dict(Xw=np.ascontiguousarray(self.mp_pos[kp_mp[idx]].astype(np.float64)), obs=np.ascontiguousarray(obs),
                         invsig2=self.inv_sigma2[kps["octave"][idx]].astype(np.float64), face=face.astype(np.int8), pose0=_pose7_from_T(T),
                         fx=F / 2.0, fy=F / 2.0, cx=F / 2.0, cy=F / 2.0)

    def _optimize_pose(self, fr):
        idx, prob = self._pose_problem(fr["T"], fr["kps"], fr["kp_mp"], fr["rays"])
        if len(idx) < 3:
            return 0
        n, pose, out = self.be.pose_optimize(prob)
        fr["T"] = _T_from_pose7(pose)
        fr["outlier"][:] = False
        fr["outlier"][idx[np.asarray(out[:len(idx)]).astype(bool)]] = True
        return int(n)

    # ---- initialisation (Tracking::MonocularInitialization, Tracking.cpp:391-465)
    def _initialize(self, i, fisheye, gt):
        k, d = self.be.extract(fisheye, init=True)
        rec = dict(frame=i, stage="init", nkp=len(k))
        if self.ini is None:
            if len(k) > 100:
                self.ini = dict(i=i, kps=k, desc=d, gt=gt, rays=self.be.last_rays, prev=np.stack([k["x"], k["y"]], 1).astype(np.float32))
            self.log.append(rec)
            return
        if len(k) <= 100:
            self.ini = None; self.log.append(rec); return
        ini = self.ini
        m12, nm = self.be.search_for_initialization(ini["kps"], ini["desc"], k, d, ini["prev"])
        rec.update(init_matches=m12.copy(), n_init=int(nm))
        if nm < 100:
            self.ini = None; self.log.append(rec); return
        # ground-truth stand-in for Initializer + GlobalBA: both poses and the matched points' positions
        R0, t0 = ini["gt"]; R1, t1 = gt
        sel = np.flatnonzero(m12 >= 0)
        ok, Xw = synth.room_points_behind_pixels(self.F, R0, t0, ini["kps"]["x"][sel].astype(np.float64), ini["kps"]["y"][sel].astype(np.float64))
        sel = sel[ok]; Xw = Xw[ok]
        T0, T1 = _tcw(R0, t0), _tcw(R1, t1)
        Ow1 = (-(T1[:3, :3].T @ T1[:3, 3])).astype(np.float32)
        ids = self._add_points(Xw, d[m12[sel]], k["octave"][m12[sel]], Ow1)
        kp_mp0 = np.full(len(ini["kps"]), -1, np.int32); kp_mp0[sel] = ids
        kp_mp1 = np.full(len(k), -1, np.int32); kp_mp1[m12[sel]] = ids
        self.kfs = [dict(T=T0, kps=ini["kps"], kp_mp=kp_mp0, frame=ini["i"], rays=ini["rays"]), dict(T=T1, kps=k, kp_mp=kp_mp1, frame=i, rays=self.be.last_rays)]
        self.last = dict(T=T1, kps=k, desc=d, kp_mp=kp_mp1, outlier=np.zeros(len(k), bool), rays=self.be.last_rays)
        self.velocity = None
        self.state = "ok"
        rec.update(n_map=len(ids))
        self.log.append(rec)

    # ---- tracking
    def _track(self, i, fisheye, gt):
        be = self.be
        k, d = be.extract(fisheye, init=False)
        rec = dict(frame=i, stage="track", nkp=len(k))
        last = self.last
        cur = dict(kps=k, desc=d, kp_mp=np.full(len(k), -1, np.int32), outlier=np.zeros(len(k), bool), rays=be.last_rays)
        # TrackWithMotionModel (Tracking.cpp:620-677); the very first tracked frame has no velocity yet: TrackReferenceKeyFrame needs BoW, the
        # harness starts the motion model from a standing camera instead
        cur["T"] = (self.velocity @ last["T"]).astype(np.float32) if self.velocity is not None else last["T"].copy()
        pose12 = np.concatenate([cur["T"][:3, :3].reshape(-1), cur["T"][:3, 3]]).astype(np.float32)
        valid = ((last["kp_mp"] >= 0) & ~last["outlier"]).astype(np.uint8)
        lm = np.maximum(last["kp_mp"], 0)
        Xw = self.mp_pos[lm] if len(self.mp_pos) else np.zeros((len(lm), 3), np.float32)
        mdesc = self.mp_desc[lm] if len(self.mp_desc) else np.zeros((len(lm), 32), np.uint8)
        kp_slot = np.full(len(k), -1, np.int32)
        match, nm = be.search_by_projection(k, d, pose12, valid, Xw, last["kps"]["octave"], last["kps"]["angle"], mdesc, kp_slot, 15.0)
        if nm < 20:
            kp_slot[:] = -1
            match, nm = be.search_by_projection(k, d, pose12, valid, Xw, last["kps"]["octave"], last["kps"]["angle"], mdesc, kp_slot, 30.0)
        rec.update(mm_match=match.copy(), n_mm=int(nm))
        if nm < 20:
            self.state = "lost"; self.log.append(rec); return
        got = kp_slot >= 0
        cur["kp_mp"][got] = last["kp_mp"][kp_slot[got]]
        n_inl = self._optimize_pose(cur)
        cur["kp_mp"][cur["outlier"]] = -1                      # discard outliers (:655-671)
        cur["outlier"][:] = False
        rec.update(n_mm_inliers=int((cur["kp_mp"] >= 0).sum()))
        if (cur["kp_mp"] >= 0).sum() < 10:
            self.state = "lost"; self.log.append(rec); return
        # TrackLocalMap (:679-719): the local map of this small scene is the whole map; points already matched are skipped (:806-822)
        taken = np.zeros(len(self.mp_pos), bool); taken[cur["kp_mp"][cur["kp_mp"] >= 0]] = True
        cand = np.flatnonzero(~taken)
        T = cur["T"]
        Ow = (-(T[:3, :3].T @ T[:3, 3])).astype(np.float32)
        pose15 = np.concatenate([T[:3, :3].reshape(-1), T[:3, 3], Ow]).astype(np.float32)
        kp_lm = np.where(cur["kp_mp"] >= 0, 1 << 20, -1).astype(np.int32)
        lmatch, nl, in_view = be.search_local_points(k, d, pose15, self.mp_pos[cand], self.mp_normal[cand], self.mp_min[cand], self.mp_max[cand], self.mp_desc[cand],
                                                     kp_lm, 1.0)
        new = (kp_lm >= 0) & (kp_lm < (1 << 20))
        cur["kp_mp"][new] = cand[kp_lm[new]]
        rec.update(lm_match=lmatch.copy(), n_lm=int(nl), n_in_view=int(np.asarray(in_view).sum()))
        n_inl = self._optimize_pose(cur)
        n_track = int(((cur["kp_mp"] >= 0) & ~cur["outlier"]).sum())
        rec.update(n_inliers=n_track, pose=cur["T"].copy())
        if n_track < 30:
            self.state = "lost"; self.log.append(rec); return
        # motion model update (:360-368): mVelocity = mCurrentFrame.mTcw * LastTwc
        Twc_last = np.eye(4, dtype=np.float32)
        Twc_last[:3, :3] = last["T"][:3, :3].T; Twc_last[:3, 3] = -(last["T"][:3, :3].T @ last["T"][:3, 3])
        self.velocity = (cur["T"] @ Twc_last).astype(np.float32)
        cur["kp_mp"][cur["outlier"]] = -1                      # (:377-384)
        cur["outlier"][:] = False
        self.last = cur
        if (i - self.kfs[-1].get("frame", 0)) >= self.kf_every:
            self._new_keyframe(i, cur, gt, rec)
        self.log.append(rec)

    def _new_keyframe(self, i, cur, gt, rec):
        """LocalMapping for this key frame: new points (ground-truth seeded stand-in for CreateNewMapPoints), then local BA"""
        k = cur["kps"]
        free = np.flatnonzero((cur["kp_mp"] < 0) & (k["octave"] <= 3))[:self.new_pts]
        Rg, tg = gt
        ok, Xw = synth.room_points_behind_pixels(self.F, Rg, tg, k["x"][free].astype(np.float64), k["y"][free].astype(np.float64))
        free = free[ok]; Xw = Xw[ok]
        T = cur["T"]
        # the seeds live in the ground-truth world; bring them into the estimated one through this frame: Xw_est = Twc_est * Tcw_gt * Xw
        Xc = Xw @ np.asarray(Rg, np.float64).T + np.asarray(tg, np.float64)
        Xw_est = (Xc - T[:3, 3].astype(np.float64)) @ T[:3, :3].astype(np.float64)
        Ow = (-(T[:3, :3].T @ T[:3, 3])).astype(np.float32)
        ids = self._add_points(Xw_est, cur["desc"][free], k["octave"][free], Ow)
        cur["kp_mp"][free] = ids
        self.kfs.append(dict(T=cur["T"].copy(), kps=k, kp_mp=cur["kp_mp"].copy(), frame=i, rays=cur["rays"]))
        rec.update(new_points=len(ids))
        # Optimizer::LocalBundleAdjustment (Optimizer.cpp:192-451): the last `ba_window` key frames are free, older ones that see the same
        # points are fixed; edges = every observation of the window's points whose ray is inside the field of view
        win = self.kfs[-self.ba_window:]
        pts = np.unique(np.concatenate([kf["kp_mp"][kf["kp_mp"] >= 0] for kf in win]))
        pt_index = np.full(len(self.mp_pos), -1, np.int64); pt_index[pts] = np.arange(len(pts))
        kf_ids = [j for j, kf in enumerate(self.kfs) if (pt_index[np.maximum(kf["kp_mp"], 0)][kf["kp_mp"] >= 0] >= 0).any()]
        kfs = [self.kfs[j] for j in kf_ids]
        first_free = len(self.kfs) - len(win)
        e_pose, e_point, e_obs, e_inv, e_face = [], [], [], [], []
        F = self.F
        for kj, kf in enumerate(kfs):
            idx = np.flatnonzero(kf["kp_mp"] >= 0)
            idx = idx[pt_index[kf["kp_mp"][idx]] >= 0]
            idx = idx[kf["rays"][idx, 2] >= self.cos_fov]                 # Optimizer.cpp:323-325 on the key frame's mvKeyRays
            px = kf["kps"]["x"][idx].astype(np.float64); py = kf["kps"]["y"][idx].astype(np.float64)
            face = synth.face_of_pixel(F, px, py)
            keep = face >= 0
            idx, px, py, face = idx[keep], px[keep], py[keep], face[keep]
            e_pose.append(np.full(len(idx), kj, np.int32)); e_point.append(pt_index[kf["kp_mp"][idx]].astype(np.int32))
            e_obs.append(np.stack([px - np.floor(px / F) * F, py - np.floor(py / F) * F], 1)); e_inv.append(self.inv_sigma2[kf["kps"]["octave"][idx]].astype(np.float64))
            e_face.append(face.astype(np.int8))
        fixed = np.array([1 if (j < first_free or j == 0) else 0 for j in kf_ids], np.uint8)
        if fixed.all() or len(pts) < 10:
            return
        prob = dict(poses=np.stack([_pose7_from_T(kf["T"]) for kf in kfs]), fixed=fixed, points=self.mp_pos[pts].astype(np.float64),
                    e_pose=np.concatenate(e_pose), e_point=np.concatenate(e_point), e_obs=np.ascontiguousarray(np.concatenate(e_obs)),
                    e_invsig2=np.concatenate(e_inv), e_face=np.concatenate(e_face), fx=F / 2.0, fy=F / 2.0, cx=F / 2.0, cy=F / 2.0)
        poses, points, outliers, its = self.be.local_ba(prob)
        rec.update(ba_edges=len(prob["e_pose"]), ba_iterations=its, ba_outliers=int(np.asarray(outliers).sum()), ba_kfs=len(kfs), ba_points=len(pts))
        # write-back through float (Optimizer.cpp:419-449); observations flagged as outliers are erased (:424-434)
        for kj, kf in enumerate(kfs):
            if not fixed[kj]:
                kf["T"] = _T_from_pose7(poses[kj])
        self.mp_pos[pts] = points.astype(np.float32)
        eo = np.flatnonzero(outliers)
        ep, ept = prob["e_pose"], prob["e_point"]
        for e in eo:
            kf = kfs[ep[e]]
            kf["kp_mp"][kf["kp_mp"] == pts[ept[e]]] = -1
        # the current frame is the newest key frame: tracking continues from its refined pose
        self.last["T"] = self.kfs[-1]["T"].copy()
        self.last["kp_mp"] = self.kfs[-1]["kp_mp"].copy()
        rec.update(pose_after_ba=self.last["T"].copy())

    def feed(self, i, fisheye, gt):
        if self.state in ("no_images", "not_initialized"):
            self.state = "not_initialized"
            self._initialize(i, fisheye, gt)
        elif self.state == "ok":
            self._track(i, fisheye, gt)
        else:
            self.log.append(dict(frame=i, stage="lost"))


def run_sequence(camd, backend, frames, gts, **kw):
    """frames: fisheye images; gts: ground-truth (Rcw, tcw) per frame.  Returns (tracker, seconds per frame list)."""
    trk = Tracker(camd, backend, **kw)
    secs = []
    for i, (f, g) in enumerate(zip(frames, gts)):
        t0 = time.perf_counter()
        trk.feed(i, f, g)
        secs.append(time.perf_counter() - t0)
    return trk, secs


def render_sequence(camd, n, seed=0xC0FFEE, n_loop=300, start=0):
    scene = synth.room_scene(seed)
    frames, gts = [], []
    for i in range(start, start + n):
        R, t = synth.room_pose(i, n_loop)
        frames.append(synth.render_fisheye(camd, scene, R, t)); gts.append((R, t))
    return frames, gts


# ---- the same stream through the Python-free driver (cubemapslam_amd/host/closed_loop_driver.cpp): files in the reference's formats
def settings_yaml(camd, nfeatures=None):
    """a settings file with the keys System::System / Tracking::Tracking read (System.cpp:63-91, Tracking.cpp:61-93)"""
    lines = ["%YAML:1.0", "Camera.Iw: %d" % camd["Iw"], "Camera.Ih: %d" % camd["Ih"], "Camera.nrpol: 5", "Camera.nrinvpol: 12"]
    lines += ["Camera.a%d: %r" % (i, float(v)) for i, v in enumerate(camd["pol"])]
    lines += ["Camera.pol%d: %r" % (i, float(v)) for i, v in enumerate(list(camd["invpol"]) + [0.0] * (12 - len(camd["invpol"])))]
    lines += ["Camera.%s: %r" % (k, float(camd[k])) for k in ("c", "d", "e", "u0", "v0")]
    lines += ["Camera.fov: %r" % float(camd["fov_deg"]), "Camera.fps: 30.0", "Camera.RGB: 1", "Camera.withFisheyeMask: 0",
              "CubeFace.w: %d" % camd["face"], "CubeFace.h: %d" % camd["face"], "ORBextractor.nFeatures: %d" % (nfeatures or camd["nfeatures"]),
              "ORBextractor.scaleFactor: 1.2", "ORBextractor.nLevels: 8", "ORBextractor.iniThFAST: 20", "ORBextractor.minThFAST: 7"]
    return "\n".join(lines) + "\n"


def _write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def export_sequence(dirpath, camd, frames, gts, mask):
    """settings.yaml, images.txt ("<timestamp> <file>", cubemap_lafida.cpp:91-107), the frames and the cubemap mask as binary PGM, and the
    ground-truth poses the stand-ins for Initializer / CreateNewMapPoints read (first line: the box room's half extents)."""
    import os
    os.makedirs(os.path.join(dirpath, "imgs"), exist_ok=True)
    with open(os.path.join(dirpath, "settings.yaml"), "w") as f:
        f.write(settings_yaml(camd))
    with open(os.path.join(dirpath, "images.txt"), "w") as f:
        for i in range(len(frames)):
            f.write("%.6f imgs/%08d.pgm\n" % (1409666701.0 + i / 30.0, i))
    for i, fr in enumerate(frames):
        _write_pgm(os.path.join(dirpath, "imgs", "%08d.pgm" % i), fr)
    _write_pgm(os.path.join(dirpath, "mask.pgm"), mask)
    with open(os.path.join(dirpath, "ground_truth.txt"), "w") as f:
        f.write("room %r %r %r\n" % tuple(float(v) for v in synth.ROOM_HALF))
        for R, t in gts:
            f.write(" ".join(repr(float(v)) for v in list(np.asarray(R, np.float64).reshape(-1)) + list(np.asarray(t, np.float64))) + "\n")


def run_driver(dirpath, kf_every=5, ba_window=8, new_points_per_kf=400, warmup=6, device=0):
    """runs cubemap_closed_loop on an exported sequence; returns (exit code, per-frame records, stdout)"""
    import json, os, subprocess
    from . import build
    log = os.path.join(dirpath, "frames.jsonl")
    cmd = [build.DRIVER, os.path.join(dirpath, "settings.yaml"), os.path.join(dirpath, "images.txt"), os.path.join(dirpath, "imgs"), os.path.join(dirpath, "mask.pgm"),
           os.path.join(dirpath, "ground_truth.txt"), "--kf-every", str(kf_every), "--ba-window", str(ba_window), "--new-points", str(new_points_per_kf),
           "--warmup", str(warmup), "--log", log, "--trajectory", os.path.join(dirpath, "KeyFrameTrajectory.txt"), "--perf", os.path.join(dirpath, "perf.txt"),
           "--device", str(device)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    recs = [json.loads(l) for l in open(log)] if os.path.exists(log) else []
    return r.returncode, recs, r.stdout + r.stderr
