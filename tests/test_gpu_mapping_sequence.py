"""LocalMapping's per-key-frame sequence as ONE composite, product against oracle (round 5's verdict, item 7): every call has its own parity test in
test_gpu_parity.py; here the calls run in the order LocalMapping::Run makes them (LocalMapping.cpp:52-117) on one consistent scene, each stage fed by
the previous stage's OUTPUT, and every intermediate result is compared with the oracle running the same sequence:

  ProcessNewKeyFrame    four rendered fisheye frames of the box room -> remap + ORB extraction on the device -> cms_kfstore_put_from_frames (device to device)
  CreateNewMapPoints    key frame 0 against its three neighbours (cms_kfstore_create_new_map_points)          == orc.create_new_map_points
  map-point bookkeeping ComputeDistinctiveDescriptors / UpdateNormalAndDepth of every point                     == the oracle's
  SearchInNeighbors     both Fuse directions as one cms_kfstore_fuse_search_sets call (LocalMapping.cpp:388-466) == orc.fuse_search per job
  Fuse's surgery        Replace / AddObservation in list order (ORBMatcher.cpp:1213-1236) through the mirror's ORBMatcher::Fuse == a Python replay on the oracle's search
  LocalBundleAdjustment the window these key frames and points make (Optimizer.cpp:192-451)                     == orc.ba_run
  pose write-back       cms_kfstore_update_poses (Optimizer.cpp:419-431), then CreateNewMapPoints AGAIN on the refined poses == the oracle's

The second neighbour pair gets map points of its own first (an oracle-only CreateNewMapPoints among the neighbours: scene set-up), so that both Fuse
directions have work and the forward direction meets key points that already hold a point (the Replace branch)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from cubemapslam_amd import api, harness, synth

pytestmark = pytest.mark.gpu
KP = api.KP_DTYPE


def _host():
    from cubemapslam_amd import build
    api.lib()                                                      # (the host library links against the HIP one)
    L = C.CDLL(build.HOST_LIB)
    L.hm_last_error.restype = C.c_char_p
    return L


def _kf_from_frame(k, d, R, t):
    n = len(k)
    node = (d[:, 0].astype(np.int32) * 4 + (d[:, 1] >> 6)) % 1024           # FeatureVector stand-in: features binned by descriptor bits (DBoW2 is the host's)
    order = np.lexsort((np.arange(n), node)).astype(np.int32)
    ids, starts = np.unique(node[order], return_index=True)
    R = np.asarray(R, np.float32); t = np.asarray(t, np.float32)
    return dict(x=k["x"].copy(), y=k["y"].copy(), octave=k["octave"].copy(), angle=k["angle"].copy(), size=k["size"].copy(), response=k["response"].copy(), desc=d.copy(),
                mp=np.full(n, -1, np.int32), R=R, t=t, Ow=(-(R.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32),
                node_id=ids.astype(np.int32), node_off=np.concatenate([starts, [n]]).astype(np.int32), node_feat=order, median_depth=np.float32(2.0))


def _point_attributes(ctx, pts, kfs, sf):
    """ComputeDistinctiveDescriptors + UpdateNormalAndDepth of every point (MapPoint.cpp:243-308, 332-373), product and oracle; returns the oracle's
    (equal) values: desc, normal, min_dist, max_dist"""
    obs_off = np.concatenate([[0], np.cumsum([len(p["obs"]) for p in pts])]).astype(np.int32)
    odesc = np.concatenate([np.stack([kfs[k]["desc"][f] for k, f in p["obs"]]) for p in pts])
    oOw = np.concatenate([np.stack([kfs[k]["Ow"] for k, f in p["obs"]]) for p in pts]).astype(np.float32)
    pos = np.stack([p["pos"] for p in pts]).astype(np.float32)
    refOw = np.stack([kfs[p["obs"][0][0]]["Ow"] for p in pts]).astype(np.float32)
    reflev = np.array([kfs[p["obs"][0][0]]["octave"][p["obs"][0][1]] for p in pts], np.int32)
    bw = orc.distinctive_descriptors(obs_off, odesc)
    bg = api.distinctive_descriptors(ctx, obs_off, odesc)
    assert np.array_equal(bw, bg)
    nw, mnw, mxw = orc.update_normal_and_depth(obs_off, pos, oOw, refOw, reflev, sf)
    ng = np.zeros_like(nw); mng = np.zeros_like(mnw); mxg = np.zeros_like(mxw)
    api.update_normal_and_depth(ctx, obs_off, pos, oOw, refOw, reflev, ng, mng, mxg)
    assert np.array_equal(nw.view(np.uint32), ng.view(np.uint32)) and np.array_equal(mnw.view(np.uint32), mng.view(np.uint32)) and np.array_equal(mxw.view(np.uint32), mxg.view(np.uint32))
    desc = np.stack([odesc[obs_off[i] + bw[i]] for i in range(len(pts))])
    return desc, nw, mnw, mxw


def test_local_mapping_sequence_product_equals_oracle_stage_by_stage():
    F = 550
    camd = synth.camera("lafida", F)
    camd["nfeatures"] = 2000
    ocam = orc.make_camera(camd)
    mask = synth.cubemap_valid_mask(camd)
    scene = synth.room_scene(0xC0FFEE)
    idx = (0, 6, 12, 18)                                           # four cameras along the room's loop: ~8 cm baselines
    gts = [synth.room_pose(i, 300) for i in idx]
    frames = np.stack([synth.render_fisheye(camd, scene, R, t) for R, t in gts])
    ctx = api.Context(camd, nfeatures=2000, max_batch=4)
    ctx.set_mask(mask)
    ctx.upload(frames); ctx.process(4, True); ctx.area_grid(4); ctx.sync()
    g = ctx.geom
    sf = np.array([g.scale[l] for l in range(g.nlevels)], np.float32); sigma2 = (sf * sf).astype(np.float32); inv_s2 = (np.float32(1.0) / sigma2).astype(np.float32)
    kfs = []
    for b in range(4):
        k, d = ctx.fetch(b)
        assert len(k) > 800, (b, len(k))
        kfs.append(_kf_from_frame(k, d, gts[b][0], gts[b][1]))
    # ---- scene set-up (oracle only): the neighbours 1, 2, 3 already share map points Q
    def okfs():
        out = []
        for kf in kfs:
            q = dict(kf); q.pop("rays", None)
            out.append(orc.make_keyframe(ocam, q))
            kf["rays_oracle"] = q["rays"]
        return out
    oks = okfs()
    cur_mp = kfs[1]["mp"].copy()
    qn, q1, q2, qx = orc.create_new_map_points(ocam, oks[1][0], [oks[2][0], oks[3][0]], sf, sigma2, cur_mp)
    assert len(qn) > 100, len(qn)
    pts = []                                                       # the map: id = index; obs = [(key frame, feature)] in AddObservation order
    for j in range(len(qn)):
        pid = len(pts)
        pts.append(dict(pos=qx[j].copy(), obs=[(1, int(q1[j])), (2 + int(qn[j]), int(q2[j]))]))
        kfs[1]["mp"][q1[j]] = pid; kfs[2 + qn[j]]["mp"][q2[j]] = pid
    nQ = len(pts)
    # ---- ProcessNewKeyFrame: the four frames enter the store device to device (key points, descriptors, key rays, grid), FeatureVector + map-point slots from the host
    cg = api.Context(camd, nfeatures=2000, max_batch=1)
    st = api.KeyframeStore(cg, max_keyframes=5, max_features=4096, max_nodes=1024)
    st.put_from_frames(ctx, [(b, b, len(kfs[b]["x"]), kfs[b]) for b in range(4)])
    for b in range(4):
        got = st.debug_fetch(b)
        assert np.array_equal(got["rays"].view(np.uint32), kfs[b]["rays_oracle"].view(np.uint32)), b      # mvKeyRays: the device's against the oracle's camera model
        assert np.array_equal(got["mp"], kfs[b]["mp"]) and np.array_equal(got["desc"], kfs[b]["desc"]) and got["kp_cnt"] == len(kfs[b]["x"])
    # ---- CreateNewMapPoints: key frame 0 against neighbours 1, 2, 3
    oks = okfs()
    cur_mp = kfs[0]["mp"].copy()
    wn, w1, w2, wx = orc.create_new_map_points(ocam, oks[0][0], [oks[1][0], oks[2][0], oks[3][0]], sf, sigma2, cur_mp)
    gn, g1, g2, gx = st.create_new_map_points([(0, [1, 2, 3])], cap=4096)[0]
    assert len(wn) > 100, len(wn)
    assert np.array_equal(gn, wn) and np.array_equal(g1, w1) and np.array_equal(g2, w2) and np.array_equal(gx.view(np.uint32), wx.view(np.uint32))
    for j in range(len(wn)):                                      # LocalMapping.cpp:359-381: the new point's observations and the key frames' slots
        pid = len(pts)
        pts.append(dict(pos=wx[j].copy(), obs=[(0, int(w1[j])), (1 + int(wn[j]), int(w2[j]))]))
        kfs[0]["mp"][w1[j]] = pid; kfs[1 + wn[j]]["mp"][w2[j]] = pid
    for b in range(4):
        st.update(b, mp=kfs[b]["mp"])
    desc, normal, dmin, dmax = _point_attributes(cg, pts, kfs, sf)
    # ---- SearchInNeighbors: set 0 = the key frame's points into every neighbour, set 1 = the neighbours' points into the key frame; a point already
    # seen by the target is skipped (pMP->IsInKeyFrame, ORBMatcher.cpp:1146)
    def in_kf(ids, k):
        seen = set(int(v) for v in kfs[k]["mp"][kfs[k]["mp"] >= 0])
        return np.array([int(i) in seen for i in ids], np.uint8)
    ids0 = np.array(sorted(set(int(v) for v in kfs[0]["mp"][kfs[0]["mp"] >= 0])), np.int64)
    cand, mark = [], set()
    for k in (1, 2, 3):                                            # LocalMapping.cpp:432-447: first appearance over the target key frames' slots in slot order
        for v in kfs[k]["mp"]:
            if v >= 0 and int(v) not in mark:
                mark.add(int(v)); cand.append(int(v))
    ids1 = np.array(cand, np.int64)
    mk = lambda ids: dict(pos=np.stack([pts[i]["pos"] for i in ids]).astype(np.float32), normal=normal[ids], min_dist=dmin[ids], max_dist=dmax[ids], desc=desc[ids])
    sets = [mk(ids0), mk(ids1)]
    jobs = [(k, 0, in_kf(ids0, k)) for k in (1, 2, 3)] + [(0, 1, in_kf(ids1, 0))]
    got = st.fuse_search_sets(sets, jobs, th=3.0)
    want = []
    for (slot, si, skip) in jobs:
        q = sets[si]
        want.append(orc.fuse_search(ocam, oks[slot][0], skip, q["pos"], q["normal"], q["min_dist"], q["max_dist"], q["desc"], 3.0, sf, inv_s2))
    for j in range(len(jobs)):
        assert np.array_equal(got[j][0], want[j][0]) and np.array_equal(got[j][1], want[j][1]), j
    n_found = [int((w[0] >= 0).sum()) for w in want]
    assert sum(n_found[:3]) > 50 and n_found[3] > 20, n_found
    # ---- the surgery (ORBMatcher.cpp:1213-1236), job after job in the reference's order: through the mirror on the product's side, replayed in Python
    # on the oracle's search results; Replace keeps the point with more observations (the other one's observations move over, MapPoint::Replace)
    L = _host()
    cam = api.make_camera(camd)
    assert L.hm_set_camera(C.byref(cam)) == 0
    img = np.zeros((3 * F, 3 * F), np.uint8); kk = np.zeros(4096, KP); dd = np.zeros((4096, 32), np.uint8)
    L.hm_extract(2000, C.c_float(1.2), 8, 20, 7, api._p(img), 3 * F, api._p(mask), 3 * F, api._p(kk), api._p(dd), 4096)      # (sizes the mirror's shared context)
    L.hm_fuse.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_void_p]
    n_replace = 0
    for j, (slot, si, _) in enumerate(jobs):
        ids = (ids0, ids1)[si]
        kf = kfs[slot]
        skip = in_kf(ids, slot)                                    # (as of this moment: earlier jobs may have added observations)
        # product: search + additions through the mirror
        kps = np.zeros(len(kf["x"]), KP)
        for f_ in ("x", "y", "octave", "angle", "size", "response"):
            kps[f_] = kf[f_]
        slots_g = kf["mp"].astype(np.int64)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = kf["R"]; T[:3, 3] = kf["t"]
        q = sets[si]
        fused_g = np.zeros(len(ids), np.int32)
        nf = L.hm_fuse(len(kps), api._p(kps), api._p(np.ascontiguousarray(kf["desc"])), api._p(slots_g), api._p(T), len(ids), api._p(q["pos"]), api._p(q["normal"]),
                       api._p(q["min_dist"]), api._p(q["max_dist"]), api._p(q["desc"]), api._p(np.ascontiguousarray(ids)), api._p(skip), C.c_float(3.0), api._p(fused_g))
        assert nf >= 0, L.hm_last_error()
        # oracle: the same search, then the reference's decisions
        bi, _ = orc.fuse_search(ocam, oks[slot][0], skip, q["pos"], q["normal"], q["min_dist"], q["max_dist"], q["desc"], 3.0, sf, inv_s2)
        slots_w = kf["mp"].astype(np.int64)
        replaced = []
        for i, pid in enumerate(ids):
            if bi[i] < 0:
                continue
            held = int(slots_w[bi[i]])
            if held < 0:
                slots_w[bi[i]] = pid                                # AddObservation + AddMapPoint
                pts[int(pid)]["obs"].append((slot, int(bi[i])))
            elif held != int(pid):
                replaced.append((int(pid), held))                  # decided below, like the reference by Observations()
        assert np.array_equal(fused_g, bi) and nf == int((bi >= 0).sum()), j
        assert np.array_equal(slots_g, slots_w), j                  # the additions, in list order
        kf["mp"] = slots_w.astype(np.int32)
        for pid, held in replaced:                                  # ORBMatcher.cpp:1222-1229
            keep_, drop = (held, pid) if len(pts[held]["obs"]) > len(pts[pid]["obs"]) else (pid, held)
            for (k_, f_) in pts[drop]["obs"]:                       # MapPoint::Replace (MapPoint.cpp:198-241): observations move unless the survivor is seen there already
                if not any(k2 == k_ for k2, _ in pts[keep_]["obs"]):
                    pts[keep_]["obs"].append((k_, f_)); kfs[k_]["mp"][f_] = keep_
                else:
                    kfs[k_]["mp"][f_] = -1
            pts[drop]["obs"] = []
            n_replace += 1
        st.update(slot, mp=kfs[slot]["mp"])
    assert n_replace > 0                                            # the Replace branch was exercised
    # ---- LocalBundleAdjustment over the four key frames and every point they still hold (key frame 3 fixed)
    live = [i for i, p in enumerate(pts) if len(p["obs"]) >= 2]
    pindex = {pid: n_ for n_, pid in enumerate(live)}
    e_pose, e_point, e_obs, e_inv, e_face = [], [], [], [], []
    for k, kf in enumerate(kfs):
        for f_ in np.flatnonzero(kf["mp"] >= 0):
            pid = int(kf["mp"][f_])
            if pid not in pindex:
                continue
            px, py = float(kf["x"][f_]), float(kf["y"][f_])
            face = int(synth.face_of_pixel(F, np.array([px]), np.array([py]))[0])
            if face < 0:
                continue
            e_pose.append(k); e_point.append(pindex[pid]); e_obs.append((px - np.floor(px / F) * F, py - np.floor(py / F) * F)); e_inv.append(float(inv_s2[kf["octave"][f_]])); e_face.append(face)
    Ts = []
    for kf in kfs:
        T = np.eye(4, dtype=np.float32); T[:3, :3] = kf["R"]; T[:3, 3] = kf["t"]; Ts.append(T)
    prob = dict(poses=np.stack([harness._pose7_from_T(T) for T in Ts]), fixed=np.array([0, 0, 0, 1], np.uint8), points=np.stack([pts[i]["pos"] for i in live]).astype(np.float64),
                e_pose=np.array(e_pose, np.int32), e_point=np.array(e_point, np.int32), e_obs=np.ascontiguousarray(np.array(e_obs, np.float64)), e_invsig2=np.array(e_inv, np.float64),
                e_face=np.array(e_face, np.int8), fx=F / 2.0, fy=F / 2.0, cx=F / 2.0, cy=F / 2.0)
    assert len(e_pose) > 600
    gb = api.ba_run(prob); wb = orc.ba_run(prob)
    assert gb["rc"] == 0 and wb["rc"] == 0 and list(gb["stats"].iterations_done) == list(wb["stats"].iterations_done)
    assert np.array_equal(gb["outliers"], wb["outliers"])
    dp = np.abs(wb["points"] - prob["points"]).max()
    assert np.abs(gb["points"] - wb["points"]).max() <= 1e-4 * dp and np.abs(gb["poses"] - wb["poses"]).max() <= 1e-4 * np.abs(wb["poses"] - prob["poses"]).max()
    # ---- write-back through float (Optimizer.cpp:419-449) into the store, then CreateNewMapPoints again on the refined poses
    R2, t2, O2 = [], [], []
    for k in range(3):
        T = harness._T_from_pose7(wb["poses"][k])
        kfs[k]["R"] = T[:3, :3].copy(); kfs[k]["t"] = T[:3, 3].copy(); kfs[k]["Ow"] = (-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))).astype(np.float32)
        R2.append(kfs[k]["R"].reshape(9)); t2.append(kfs[k]["t"]); O2.append(kfs[k]["Ow"])
    st.update_poses([0, 1, 2], np.stack(R2), np.stack(t2), np.stack(O2))
    for k in range(3):
        hdr = st.debug_fetch(k)["header"]
        assert np.array_equal(hdr[6:21], np.concatenate([R2[k], t2[k], O2[k]]).astype(np.float32).view(np.uint32)), k
    # (every second point of key frame 0 is culled first -- LocalMapping::MapPointCulling, LocalMapping.cpp:177-207 --, so that there is something to triangulate again)
    for pid in range(nQ + 1, len(pts), 2):
        for k_, f_ in pts[pid]["obs"]:
            if kfs[k_]["mp"][f_] == pid:
                kfs[k_]["mp"][f_] = -1
        pts[pid]["obs"] = []
    for b in range(4):
        st.update(b, mp=kfs[b]["mp"])
    oks = okfs()
    cur_mp = kfs[0]["mp"].copy()
    wn, w1, w2, wx = orc.create_new_map_points(ocam, oks[0][0], [oks[1][0], oks[2][0], oks[3][0]], sf, sigma2, cur_mp)
    gn, g1, g2, gx = st.create_new_map_points([(0, [1, 2, 3])], cap=4096)[0]
    assert len(wn) > 30, len(wn)
    assert np.array_equal(gn, wn) and np.array_equal(g1, w1) and np.array_equal(g2, w2) and np.array_equal(gx.view(np.uint32), wx.view(np.uint32))
    print("composite: %d + %d map points, Fuse found %s, %d replaced, BA %d edges %s iterations, %d new points after the write-back" % (
        nQ, len(pts) - nQ, n_found, n_replace, len(e_pose), list(wb["stats"].iterations_done), len(wn)))
    st.close(); cg.close(); ctx.close()
