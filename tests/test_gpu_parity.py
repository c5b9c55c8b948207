"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle, stage by stage and end to end.
Integer / index / byte results must be bit-exact; BA within 1e-4 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import orc
from cubemapslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _cfg(name, F, nfeat, Ih=None):
    camd = synth.camera(name, F, Ih=Ih)
    return camd, orc.make_camera(camd), nfeat


def _sorted_rows(a):
    return a[np.lexsort((a[:, 2], a[:, 0], a[:, 1]))] if len(a) else a


def _compare_frame(ctx, b, o, ocam, cube, mask, tag):
    """stage-by-stage comparison of frame b of the last process() call against the oracle run on `cube`."""
    want_k, want_d = o.extract(ocam, cube, mask)
    L = ctx.geom.nlevels
    for l in range(L):
        lv_o = o.level(l)
        lv_g = ctx.debug_level(b, l)
        assert lv_g.shape == lv_o.shape, (tag, "level shape", l)
        nbad = int((lv_g != lv_o).sum())
        assert nbad == 0, (tag, "pyramid level %d: %d differing pixels" % (l, nbad))
    for l in range(L):
        c_o = _sorted_rows(o.candidates(l))
        c_g = _sorted_rows(ctx.debug_candidates(b, l))
        assert c_g.shape == c_o.shape and np.array_equal(c_g, c_o), (tag, "FAST candidates level %d: %d vs %d" % (l, len(c_g), len(c_o)))
    for l in range(L):
        d_o = o.distributed(l)
        d_g = ctx.debug_distributed(b, l)
        want = np.stack([d_o["x"], d_o["y"], d_o["response"]], 1).astype(np.int32) if len(d_o) else np.zeros((0, 3), np.int32)
        assert d_g.shape == want.shape and np.array_equal(d_g, want), (tag, "octree level %d: %d vs %d" % (l, len(d_g), len(want)))
    got_k, got_d = ctx.fetch(b)
    assert len(got_k) == len(want_k), (tag, "key-point count %d vs %d" % (len(got_k), len(want_k)))
    for f in ("x", "y", "size", "response", "octave", "angle"):
        bad = np.nonzero(got_k[f].view(np.uint32 if f != "octave" else np.int32) != want_k[f].view(np.uint32 if f != "octave" else np.int32))[0]
        assert len(bad) == 0, (tag, "key-point field %s differs at %d points, first %s vs %s" % (f, len(bad), got_k[f][bad[:3]], want_k[f][bad[:3]]))
    nbits = int(np.unpackbits(got_d ^ want_d).sum())
    assert nbits == 0, (tag, "descriptors: %d differing bits in %d rows" % (nbits, int((got_d != want_d).any(1).sum())))
    # Frame::mvKeyRays (Frame::ComputeKeyPointRays -> CamModelGeneral::TransformCubemapToRays, Frame.cpp:746-760): bit for bit
    want_r = orc.keyframe_rays(ocam, want_k["x"], want_k["y"])
    got_r = ctx.fetch_rays(b)
    assert got_r.shape == want_r.shape and np.array_equal(got_r.view(np.uint32), want_r.view(np.uint32)), (tag, "key rays", int((got_r.view(np.uint32) != want_r.view(np.uint32)).any(1).sum()))
    return len(got_k)


def test_lut_and_remap_bit_exact():
    for name, F, Ih in (("lafida", 450, None), ("front", 650, 1024)):
        camd, ocam, _ = _cfg(name, F, 2000, Ih)
        ctx = api.Context(camd, nfeatures=500)
        m1, m2 = orc.build_lut(ocam)
        lut = ctx.debug_lut()
        sx = np.rint((m1 * np.float32(32)).astype(np.float64)).astype(np.int64)
        sy = np.rint((m2 * np.float32(32)).astype(np.float64)).astype(np.int64)
        want = ((sx >> 5) | ((sy >> 5) << 11) | ((sx & 31) << 22) | ((sy & 31) << 27)).astype(np.uint32)
        assert np.array_equal(lut, want), (name, int((lut != want).sum()))
        fish = synth.texture(camd["Ih"], camd["Iw"], 21)
        canvas = np.full((3 * F, 3 * F), 77, np.uint8)
        got = ctx.remap(fish, canvas)
        ref = orc.fisheye_to_cubemap(ocam, m1, m2, fish)
        faces = np.zeros((3 * F, 3 * F), bool)
        for (ox, oy) in synth._FACE_ORIGIN.values():
            faces[oy * F:(oy + 1) * F, ox * F:(ox + 1) * F] = True
        assert np.array_equal(got[faces], ref[faces]), (name, int((got[faces] != ref[faces]).sum()))
        assert np.all(got[~faces] == 77)      # corner blocks of the caller's canvas are left untouched (System.cpp:327-355)
        ctx.close()


@pytest.mark.parametrize("gaussian_mode", [0, 1])
@pytest.mark.parametrize("name,F,nfeat,Ih", [("lafida", 450, 2000, None), ("front", 650, 3000, 1024), ("lafida", 550, 2000, None),
                                             ("front", 650, 3000, None)])
def test_extract_stage_by_stage_bit_exact(name, F, nfeat, Ih, gaussian_mode):
    """configs[0] (Lafida F=450), configs[1] (single 1280x1024 frame, F=650, 5-face ORB extract), configs[2]'s extractor geometry
    (Lafida F=550: what bench.py times) and configs[4]'s (front_cam 1280x720, F=650, nFeatures 3000) of BASELINE.json -- each under both
    definitions of the 8-bit Gaussian (cms_set_gaussian_mode: 0 integer, 1 the SSE2 float column of an x86 OpenCV <= 3.2, ORBExtractor.cpp:907-908)."""
    camd, ocam, _ = _cfg(name, F, nfeat, Ih)
    ctx = api.Context(camd, nfeatures=nfeat, max_batch=2)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    if gaussian_mode:
        ctx.set_gaussian_mode(gaussian_mode)
    m1, m2 = orc.build_lut(ocam)
    o = orc.Orb(nfeatures=nfeat, gaussian_column_mode=gaussian_mode)
    frames = np.stack([synth.texture(camd["Ih"], camd["Iw"], 1), synth.texture(camd["Ih"], camd["Iw"], 2)])
    ctx.upload(frames)
    ctx.process(2, True)
    ctx.sync()
    for b in range(2):
        cube = orc.fisheye_to_cubemap(ocam, m1, m2, frames[b])
        n = _compare_frame(ctx, b, o, ocam, cube, mask, "%s/F%d/frame%d" % (name, F, b))
        assert n > 500
    ctx.close()


@pytest.mark.parametrize("name,F,nfeat", [("lafida", 450, 2000), ("lafida", 550, 2000), ("lafida", 650, 2000), ("front", 650, 3000)])
def test_extract_with_the_reference_masks_bit_exact(name, F, nfeat):
    """The stage-by-stage comparison of test_extract_stage_by_stage_bit_exact with the masks the REFERENCE ships (Masks/gray_lafida_cubemap_mask_
    {450,550,650}.png, gray_cubemap_front_mask_650.png; committed as data by tests/golden/make_reference_fixtures.py) instead of the
    model-derived synthetic one: the irregular hand-drawn edge at the cull of ORBExtractor.cpp:887-904, at the face sizes of Config/*.yaml."""
    import refdata
    camd, ocam, _ = _cfg(name, F, nfeat)
    ctx = api.Context(camd, nfeatures=nfeat, max_batch=2)
    mask = refdata.reference_mask(name, F)
    ctx.set_mask(mask)
    m1, m2 = orc.build_lut(ocam)
    o = orc.Orb(nfeatures=nfeat)
    frames = np.stack([synth.texture(camd["Ih"], camd["Iw"], 51), synth.texture(camd["Ih"], camd["Iw"], 52)])
    ctx.upload(frames)
    ctx.process(2, True)
    ctx.sync()
    for b in range(2):
        cube = orc.fisheye_to_cubemap(ocam, m1, m2, frames[b])
        n = _compare_frame(ctx, b, o, ocam, cube, mask, "%s/F%d/reference mask/frame%d" % (name, F, b))
        assert n > 500
        gk, _ = ctx.fetch(b)
        assert np.all(mask[(gk["y"] + 0.5).astype(int), (gk["x"] + 0.5).astype(int)] != 0)
    ctx.close()


def test_extract_with_the_sse2_gaussian_definition():
    """cms_set_gaussian_mode(1): descriptors from the float-column Gaussian an x86 OpenCV <= 3.2 computes (ties to even)
    against the oracle in the same mode, bit-exact; key points (which do not depend on the blur) stay what mode 0 gives."""
    camd, ocam, _ = _cfg("lafida", 450, 2000)
    ctx = api.Context(camd, nfeatures=2000, max_batch=3)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    m1, m2 = orc.build_lut(ocam)
    frames = np.stack([synth.texture(camd["Ih"], camd["Iw"], s) for s in (31, 32, 33)])
    ctx.upload(frames)
    ctx.process(3, True); ctx.sync()
    base = [ctx.fetch(b) for b in range(3)]
    ctx.set_gaussian_mode(1)
    ctx.process(3, True); ctx.sync()
    o1 = orc.Orb(nfeatures=2000, gaussian_column_mode=1)
    for b in range(3):
        cube = orc.fisheye_to_cubemap(ocam, m1, m2, frames[b])
        wk, wd = o1.extract(ocam, cube, mask)
        gk, gd = ctx.fetch(b)
        assert np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gk.view(np.uint8), base[b][0].view(np.uint8))
        assert np.array_equal(gd, wd), (b, int(np.unpackbits(gd ^ wd).sum()))
    ctx.set_gaussian_mode(0)
    with pytest.raises(api.CmsError):
        ctx.set_gaussian_mode(2)
    ctx.close()


def test_extract_host_api_and_dense_texture():
    """ORBextractor::operator() drop-in on a caller-supplied cubemap image that is textured everywhere (corner blocks
    included, like a real call) -> many candidates per cell, octree final phase with ties, all-valid mask except a band."""
    camd, ocam, _ = _cfg("lafida", 250, 1500)
    ctx = api.Context(camd, nfeatures=1500)
    W = 750
    img = synth.texture(W, W, 5)
    img[200:420, 100:600] = (img[200:420, 100:600] // 20) + 90      # low-contrast slab -> minThFAST fallback cells
    mask = np.full((W, W), 255, np.uint8)
    mask[:70] = 0; mask[-70:] = 0; mask[:, :70] = 0; mask[:, -70:] = 0
    ctx.set_mask(mask)
    o = orc.Orb(nfeatures=1500)
    want_k, want_d = o.extract(ocam, img, mask)
    got_k, got_d = ctx.extract(img)
    assert len(got_k) == len(want_k) and len(got_k) > 400
    assert np.array_equal(got_k.view(np.uint8), want_k.view(np.uint8))
    assert np.array_equal(got_d, want_d)
    _compare_frame(ctx, 0, o, ocam, img, mask, "dense")
    # second call on the same context after a remap-based call: no stale state
    fish = synth.texture(camd["Ih"], camd["Iw"], 9)
    k1, d1 = ctx.remap_extract(fish)
    m1, m2 = orc.build_lut(ocam)
    cube = orc.fisheye_to_cubemap(ocam, m1, m2, fish)
    k2, d2 = o.extract(ocam, cube, mask)
    assert np.array_equal(k1.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d1, d2)
    _compare_frame(ctx, 0, o, ocam, cube, mask, "remap after a caller canvas (corner blocks rewritten)")
    # third call: the corner blocks are known to be 0 now, remap / resize / FAST skip them -- same results, every level
    fish = synth.texture(camd["Ih"], camd["Iw"], 10)
    k1, d1 = ctx.remap_extract(fish)
    cube = orc.fisheye_to_cubemap(ocam, m1, m2, fish)
    k2, d2 = o.extract(ocam, cube, mask)
    assert np.array_equal(k1.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d1, d2)
    _compare_frame(ctx, 0, o, ocam, cube, mask, "remap with corner skipping")
    ctx.close()


def test_extract_edge_cases():
    camd, ocam, _ = _cfg("lafida", 150, 1000)
    ctx = api.Context(camd, nfeatures=1000)
    W = 450
    o = orc.Orb(nfeatures=1000)
    # flat image: no corners anywhere -> zero key points, no hang
    flat = np.full((W, W), 128, np.uint8)
    mask = np.full((W, W), 255, np.uint8)
    ctx.set_mask(mask)
    k, d = ctx.extract(flat)
    assert len(k) == 0 and d.shape == (0, 32)
    # all-zero mask: everything culled
    img = synth.texture(W, W, 3)
    ctx.set_mask(np.zeros((W, W), np.uint8))
    k, d = ctx.extract(img)
    assert len(k) == 0
    # a handful of isolated corners (fewer than N/100 per level: the case the reference's loop cannot leave)
    sparse = np.full((W, W), 100, np.uint8)
    sparse[200:230, 210:250] = 200
    ctx.set_mask(mask)
    want_k, want_d = o.extract(ocam, sparse, mask)
    got_k, got_d = ctx.extract(sparse)
    assert np.array_equal(got_k.view(np.uint8), want_k.view(np.uint8)) and np.array_equal(got_d, want_d)
    # noise image: maximum corner density, candidate lists and octree at their largest
    noise = np.random.RandomState(4).randint(0, 256, (W, W)).astype(np.uint8)
    want_k, want_d = o.extract(ocam, noise, mask)
    got_k, got_d = ctx.extract(noise)
    assert len(got_k) == len(want_k)
    assert np.array_equal(got_k.view(np.uint8), want_k.view(np.uint8)) and np.array_equal(got_d, want_d)
    ctx.close()


def test_hamming_best2_and_matrix_bit_exact():
    camd, _, _ = _cfg("lafida", 150, 500)
    ctx = api.Context(camd, nfeatures=500)
    rs = np.random.RandomState(5)
    for (nq, nt, mean) in ((1, 1, 1), (257, 100, 3), (1500, 2000, 40), (300, 5000, 700)):
        q = synth.descriptors(nq, 10 + nq)
        t = synth.descriptors(nt, 20 + nt)
        # plant near-duplicates so that exact ties on the minimum occur
        for i in range(0, min(nq, nt), 3):
            t[i] = q[i]
            if i + 1 < nt:
                t[i + 1] = q[i]
        off, idx = synth.candidate_lists(nq, nt, mean, 30 + nq)
        lvl = rs.randint(0, 8, nt).astype(np.int32)
        excl = (rs.uniform(size=nt) < 0.1).astype(np.uint8)
        got = ctx.hamming_best2(q, t, off, idx, lvl, excl)
        # oracle has no exclusion input: filter the lists first (the reference skips them inside the scan, ORBMatcher.cpp:91-95)
        keep = excl[idx] == 0
        cnt = np.add.reduceat(np.r_[keep.astype(np.int64), 0], off[:-1]) * (np.diff(off) > 0) if len(idx) else np.zeros(nq, np.int64)
        off2 = np.zeros(nq + 1, np.int32); off2[1:] = np.cumsum(cnt)
        want = orc.hamming_best2(q, t, off2, idx[keep], lvl)
        for k in want:
            assert np.array_equal(got[k], want[k]), (nq, nt, k, int((got[k] != want[k]).sum()))
    a = synth.descriptors(70, 1); b = synth.descriptors(333, 2)
    assert np.array_equal(ctx.hamming_matrix(a, b), orc.hamming_matrix(a, b))
    # linearity-style property at full size: distance of a descriptor to itself is 0 and best match finds it
    big = synth.descriptors(9000, 7)
    off = np.arange(0, 9001 * 64, 64, dtype=np.int32)[:9001]
    idx = ((np.arange(9000)[:, None] + np.arange(64)[None, :] * 131) % 9000).astype(np.int32).ravel()
    got = ctx.hamming_best2(big, big, off, idx)
    assert np.all(got["best_dist"] == 0) and np.array_equal(got["best_idx"], np.arange(9000))
    ctx.close()


def test_ba_linearize_matches_oracle():
    prob = synth.ba_problem(K=6, P=500, obs_per_point=4, F=650, seed=11)
    for robust in (True, False):
        g = api.ba_linearize(prob, robust=robust)
        w = orc.ba_linearize(prob, robust=robust)
        assert np.array_equal(g["err"], w["err"])       # residual incl. the float round trip: same operations, same bits expected
        for k in ("Hpp", "bp", "Hll", "bl", "Hpl"):
            scale = np.abs(w[k]).max()
            assert np.allclose(g[k], w[k], rtol=1e-9, atol=1e-9 * scale), (robust, k, np.abs(g[k] - w[k]).max() / scale)
        assert abs(g["chi"][0] - w["chi"][0]) <= 1e-10 * abs(w["chi"][0])


def _ba_updates_close(prob, poses, pts, w, tol=1e-4, tag=""):
    """BA parity bar (north_star: 1e-4 relative on pose / point UPDATES), per block: every key frame's translation update, every key
    frame's rotation update (quaternion difference) and every point's update is compared on its own against the oracle's update of
    that block, relative to that block's update norm, with an absolute floor of 1 % of the median block update (a block the
    optimisation barely moves is not held to 1e-4 of nothing).  Metres and quaternion units are never mixed in one norm."""
    def chk(got, want, ref0, what):
        du_w = want - ref0
        err = np.linalg.norm(got - want, axis=1)
        nrm = np.linalg.norm(du_w, axis=1)
        moved = nrm[nrm > 0]
        floor = 0.01 * (np.median(moved) if len(moved) else 0.0)
        lim = tol * np.maximum(nrm, floor)
        bad = np.nonzero(err > lim)[0]
        assert len(bad) == 0, (tag, what, "%d blocks beyond %g relative; worst err %.3g for an update of %.3g (floor %.3g)" %
                               (len(bad), tol, err[bad].max(), nrm[bad[np.argmax(err[bad])]], floor))
        return float((err / np.maximum(nrm, floor)).max()) if len(err) else 0.0
    q0 = prob["poses"][:, 3:] / np.linalg.norm(prob["poses"][:, 3:], axis=1, keepdims=True)
    q0 = np.where(q0[:, 3:4] < 0, -q0, q0)           # the SE3Quat constructor's normalisation (se3quat.h:58-64)
    worst = (chk(poses[:, :3], w["poses"][:, :3], prob["poses"][:, :3], "pose translation"),
             chk(poses[:, 3:], w["poses"][:, 3:], q0, "pose rotation"),
             chk(pts, w["points"], prob["points"], "point"))
    return worst


def _ba_updates_close_or_cascade(prob, poses, pts, w, tol=1e-4, tag=""):
    """The parity bar with the one exception the reference's own arithmetic forces.  multipinhole_project rounds the camera-frame point to
    float inside the double optimisation (g2o_cubemap_vertices_edges.cpp:225-233): two runs whose estimates differ in the 15th digit can
    round a coordinate differently, every flip moves a residual by a float ulp and makes further flips likelier -- the oracle does this to
    ITSELF when its input moves by 1e-12 m (tests/test_oracle_ba.py::test_reference_algorithm_is_chaotic_at_float_rounding: key frames
    ~1e-6, a few hundred points up to ~1e-3 of their update).  About one 80 k-edge window in twenty cascades between product and oracle, in
    every kernel path and in round 2's library alike.  Such a window is held to: key-frame updates within `tol` per block as always; point
    updates no worse than the oracle's own self-difference under a 1e-12 m perturbation (three times as many points beyond `tol`, five times
    the worst value).  Returns (worst relative error, cascaded?)."""
    try:
        return max(_ba_updates_close(prob, poses, pts, w, tol, tag)), False
    except AssertionError as strict:
        if "'point'" not in str(strict):
            raise                                         # a key-frame block beyond the bar is never excused
    from test_oracle_ba import ba_point_update_errors
    rs = np.random.RandomState(1)
    p2 = dict(prob); p2["points"] = prob["points"] + rs.normal(0, 1e-12, prob["points"].shape)
    w2 = orc.ba_run(p2)
    r_self = ba_point_update_errors(prob, w2, w)
    r_prod = ba_point_update_errors(prob, dict(points=pts), w)
    assert (r_prod > tol).sum() <= 3 * (r_self > tol).sum() + 3 and r_prod.max() <= 5 * max(r_self.max(), tol), (
        tag, "points beyond the bar %d (oracle against itself: %d), worst %.3g (%.3g)" % ((r_prod > tol).sum(), (r_self > tol).sum(), r_prod.max(), r_self.max()))
    return float(r_prod.max()), True


def _ba_compare(prob, tol=1e-4, count_cascade=None):
    """count_cascade: a list; the window is then held to the cascade-aware bar (_ba_updates_close_or_cascade) and whether it cascaded is appended --
    a window known to cascade at float-rounding level is COUNTED by its test, not replaced by a tamer seed"""
    g = api.ba_run(prob)
    w = orc.ba_run(prob)
    assert g["rc"] == 0 and w["rc"] == 0
    gs, ws = g["stats"], w["stats"]
    assert list(gs.iterations_done) == list(ws.iterations_done), (list(gs.iterations_done), list(ws.iterations_done))
    for i in range(2):
        assert abs(gs.chi2_final[i] - ws.chi2_final[i]) <= 1e-6 * abs(ws.chi2_final[i])
    if count_cascade is None:
        _ba_updates_close(prob, g["poses"], g["points"], w, tol)
    else:
        worst, cascaded = _ba_updates_close_or_cascade(prob, g["poses"], g["points"], w, tol, tag="K %d P %d" % (len(prob["poses"]), len(prob["points"])))
        count_cascade.append((cascaded, worst))
    assert np.array_equal(g["outliers"], w["outliers"])
    return g, w


def test_ba_run_small_window():
    _ba_compare(synth.ba_problem(K=6, P=400, obs_per_point=4, F=650, seed=9))
    # all poses fixed but one; stop flag set before the call
    prob = synth.ba_problem(K=4, P=100, obs_per_point=3, F=650, seed=2)
    prob["fixed"][:3] = 1
    _ba_compare(prob)
    out = api.ba_run(prob, stop=True)
    assert out["rc"] == 1 and np.array_equal(out["points"], prob["points"])


def test_ba_run_larger_windows_all_solver_paths():
    """reduced system of 24 pose blocks (blocked LDL^T), 6*29 = 174 (register-tiled scalar LDL^T) and 6*39 = 234
    (global-memory fallback)."""
    _ba_compare(synth.ba_problem(K=25, P=1500, obs_per_point=5, F=550, seed=3))
    # the three-lane solve's limits: 25 free key frames (16 wavefronts of 21 blocks, 150 unknowns = the back substitution's third register) and one
    _ba_compare(synth.ba_problem(K=26, P=1500, obs_per_point=5, F=550, seed=16))
    # ... and seed 6 of the same shape: one of the windows whose POINTS cascade at float-rounding level between product and oracle (DESIGN.md section 2:
    # 16 of 1500 points at 2e-4 with every solve kernel since round 4).  It is held to the cascade-aware bar -- key frames within 1e-4 per block as
    # always, points no worse than the oracle against itself under a 1e-12 m perturbation -- and COUNTED here, not avoided
    counted = []
    _ba_compare(synth.ba_problem(K=26, P=1500, obs_per_point=5, F=550, seed=6), count_cascade=counted)
    assert len(counted) == 1, "the K = 26, seed 6 window was not checked"
    print("K = 26, seed 6: %s, worst relative point-update error %.3g (counted, not excused)" % ("CASCADED at float-rounding level" if counted[0][0] else "within 1e-4", counted[0][1]))
    _ba_compare(synth.ba_problem(K=2, P=120, obs_per_point=2, F=550, seed=7))
    _ba_compare(synth.ba_problem(K=30, P=2000, obs_per_point=5, F=550, seed=5))
    _ba_compare(synth.ba_problem(K=40, P=2500, obs_per_point=6, F=550, seed=4))


def test_ba_optimize_many_lockstep_equals_individual_runs():
    """cms_ba_optimize_many: two different windows advanced in lock-step give what two separate runs give."""
    probs = [synth.ba_problem(K=6, P=400, obs_per_point=4, F=650, seed=9), synth.ba_problem(K=9, P=700, obs_per_point=5, F=550, seed=13)]
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for ba, p, st in zip(bas, probs, stats):
        poses, pts, flags = ba.read()
        w = orc.ba_run(p)
        assert list(st.iterations_done) == list(w["stats"].iterations_done)
        _ba_updates_close(p, poses, pts, w)
        assert np.array_equal(flags, w["outliers"])
        ba.close()


def test_ba_optimize_many_mixed_sizes_point_major_schur():
    """The batched driver with the point-major Schur kernel on windows of very different shape: 19 / 24 / 11 free key frames
    (171 / 276 / 55 off-diagonal pose pairs -> different owner-thread layouts in one launch), 3 to 6 observations per point."""
    probs = [synth.ba_problem(K=20, P=3000, obs_per_point=4, F=550, seed=21), synth.ba_problem(K=25, P=1500, obs_per_point=6, F=550, seed=22),
             synth.ba_problem(K=12, P=900, obs_per_point=3, F=650, seed=23), synth.ba_problem(K=20, P=3000, obs_per_point=4, F=550, seed=24)]
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for ba, p, st in zip(bas, probs, stats):
        poses, pts, flags = ba.read()
        w = orc.ba_run(p)
        assert list(st.iterations_done) == list(w["stats"].iterations_done)
        _ba_updates_close(p, poses, pts, w)
        assert np.array_equal(flags, w["outliers"])
        ba.close()


def test_ba_optimize_many_config4_size_eight_different_windows():
    """The path bench.py times: cms_ba_optimize_many (kb_ba_* kernels, device-side Levenberg, point-major Schur with rebuilt 6x3
    blocks) on a group of EIGHT windows of configs[3] size (K = 20, E ~ 80 000) built from eight different seeds, so the windows of a
    launch take different accept / reject decisions and finish after different numbers of rounds.  Every window against its own
    oracle run: iteration counts of both stages, chi2, outlier flags, pose and point updates per block."""
    probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i) for i in range(8)]
    # make the group heterogeneous in behaviour, not only in data: one window starts far from the optimum (more iterations, rejected
    # trials), one has almost no outliers
    rng = np.random.default_rng(7)
    probs[3]["points"] = probs[3]["points"] + rng.normal(0, 0.05, probs[3]["points"].shape)
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    its = set()
    for i, (ba, p, st) in enumerate(zip(bas, probs, stats)):
        assert 78000 < ba.E < 82000
        poses, pts, flags = ba.read()
        w = orc.ba_run(p)
        ws = w["stats"]
        assert list(st.iterations_done) == list(ws.iterations_done), (i, list(st.iterations_done), list(ws.iterations_done))
        for j in range(2):
            assert abs(st.chi2_final[j] - ws.chi2_final[j]) <= 1e-6 * abs(ws.chi2_final[j]), (i, j)
        assert st.n_outliers_mid == ws.n_outliers_mid and st.n_outliers_final == ws.n_outliers_final, i
        assert np.array_equal(flags, w["outliers"]), (i, int((flags != w["outliers"]).sum()))
        _ba_updates_close(p, poses, pts, w, tag="window %d" % i)
        its.add(tuple(st.iterations_done))
        ba.close()


def test_ba_optimize_many_stop_flag_raised_mid_run():
    """forceStopFlag semantics of the grouped driver (Optimizer.cpp:256-257, 359-366; g2o polls the flag at every trial boundary): a
    second thread raises the flag while the group is running.  With a stop pointer the driver queues a trial only after the previous
    one finished, so what every window is left with must be EXACTLY the state after a whole number of its trials: it equals the oracle
    stopped after r trials for some r -- iteration counts of both stages, outlier flags and the estimate."""
    import threading
    import time
    probs = [synth.ba_problem(K=20, P=6000, obs_per_point=4, F=550, seed=70 + i) for i in range(4)]
    full = [orc.ba_run(p) for p in probs]
    total = max(sum(f["stats"].iterations_done) for f in full)
    warm = [api.BundleAdjuster(p) for p in probs]           # first call allocates the group's resources: keep that out of the timing
    api.ba_optimize_many(warm, (1, 0))
    hit = None
    for delay in (0.001, 0.002, 0.0005, 0.004, 0.00025, 0.008, 0.0001):
        bas = [api.BundleAdjuster(p) for p in probs]
        stop = np.zeros(1, np.uint8)
        out = {}

        def run():
            out["res"] = api.ba_optimize_many(bas, (5, 10), stop_array=stop)
        th = threading.Thread(target=run)
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < delay:
            pass
        stop[0] = 1
        th.join()
        rc, stats = out["res"]
        assert rc in (0, 1)
        done = [tuple(st.iterations_done) for st in stats]
        res = [ba.read() for ba in bas]
        for ba in bas:
            ba.close()
        if rc == 1 or all(sum(d) == 0 for d in done):
            continue                      # too early: nothing ran (covered by test_ba_run_small_window); try a longer delay
        if all(d == tuple(f["stats"].iterations_done) for d, f in zip(done, full)):
            continue                      # the flag came too late: the run completed; try a shorter delay
        hit = (done, res)
        break
    for ba in warm:
        ba.close()
    assert hit is not None, "could not raise the flag mid-run with any delay"
    done, res = hit
    for i, (p, (poses, pts, flags), d) in enumerate(zip(probs, res, done)):
        cands = []
        for r in range(1, total + 3):
            c = orc.ba_run_stop_after(p, r)
            if tuple(c["stats"].iterations_done) == d:
                cands.append((r, c))
        assert cands, ("no whole number of trials explains window %d's iteration counts" % i, d)
        errs = []
        for r, c in cands:
            try:
                assert np.array_equal(flags, c["outliers"]), "flags"
                _ba_updates_close(p, poses, pts, c, tag="window %d stopped after trial %d" % (i, r))
                errs = None
                break
            except AssertionError as e:
                errs.append((r, str(e)[:200]))
        assert errs is None, errs


def test_ba_run_config4_full_size():
    """BASELINE.json configs[3]: 20 keyframes x 4000 edges (E = 80 000, P = 20 000), 1e-4 vs the CPU path."""
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=650, seed=42)
    assert 78000 < len(prob["e_pose"]) < 82000
    g, w = _ba_compare(prob)
    assert 0.05 < g["outliers"].mean() < 0.20      # 5 % gross outliers + the 5 % chi2 tail of the inliers + unobservable ones


def _pose_close(got, want, ref0, tol=1e-4):
    """pose UPDATE within tol relative (north_star): compare the change from the initial pose"""
    du = np.linalg.norm(np.concatenate([want[:3] - ref0[:3], want[3:] - ref0[3:]]))
    err = np.linalg.norm(np.concatenate([got[:3] - want[:3], got[3:] - want[3:]]))
    return err <= tol * max(du, 1e-12), (err, du)


def test_pose_optimization_matches_oracle():
    """Optimizer::PoseOptimization: four-round pose-only LM in one kernel vs the CPU oracle -- same inlier
    count, identical outlier flags and per-round iteration counts, pose update within 1e-4 relative."""
    from cubemapslam_amd import synth as sy
    # frames of up to 1024 edges run with their edges in registers; a batch holding a larger frame takes the in-memory variant: both here
    small = [sy.pose_problem(N=n, seed=s, outlier_frac=o) for n, s, o in
             ((600, 11, 0.1), (1024, 12, 0.2), (257, 13, 0.05), (40, 14, 0.0), (9, 15, 0.0), (2, 16, 0.0), (300, 17, 0.5), (1000, 18, 0.3))]
    pos = api.PoseOptimizer(len(small), sum(len(p["Xw"]) for p in small))
    ninl_s, poses_s, outs_s, stats_s = pos.optimize(small)
    for f, pr in enumerate(small):
        w_n, w_pose, w_out, w_st = orc.pose_optimize(pr)
        assert ninl_s[f] == w_n and np.array_equal(outs_s[f], w_out), (f, ninl_s[f], w_n)
        assert stats_s[f].rounds == w_st.rounds, f
        same_iters = list(stats_s[f].iterations_done) == list(w_st.iterations_done)
        if len(pr["Xw"]) >= 3:
            # a round that is already converged ends when ten damped trials in a row fail to lower chi2 by even one ulp; whether the
            # last one does is decided by the last bit of chi2 (the device Jacobian contracts to FMAs), so a round may run one
            # iteration more or fewer there -- accepted only when the poses then agree to 1e-10 of the update instead of 1e-4
            ok, info = _pose_close(poses_s[f], w_pose, pr["pose0"] / np.concatenate([[1, 1, 1], [np.linalg.norm(pr["pose0"][3:])] * 4]),
                                   tol=1e-4 if same_iters else 1e-10)
            assert ok, (f, info, list(stats_s[f].iterations_done), list(w_st.iterations_done))
        else:
            assert same_iters, f
    pos.close()
    probs = [sy.pose_problem(N=n, seed=s, outlier_frac=o) for n, s, o in
             ((600, 1, 0.1), (150, 2, 0.3), (1500, 3, 0.05), (40, 4, 0.0), (9, 5, 0.0), (2, 6, 0.0), (300, 7, 0.5))]
    po = api.PoseOptimizer(len(probs), sum(len(p["Xw"]) for p in probs))
    ninl, poses, outs, stats = po.optimize(probs)
    for f, pr in enumerate(probs):
        w_n, w_pose, w_out, w_st = orc.pose_optimize(pr)
        assert ninl[f] == w_n, (f, ninl[f], w_n)
        assert np.array_equal(outs[f], w_out), (f, int((outs[f] != w_out).sum()))
        assert stats[f].rounds == w_st.rounds and list(stats[f].iterations_done) == list(w_st.iterations_done), \
            (f, list(stats[f].iterations_done), list(w_st.iterations_done))
        if len(pr["Xw"]) >= 3:
            ok, info = _pose_close(poses[f], w_pose, pr["pose0"] / np.concatenate([[1, 1, 1], [np.linalg.norm(pr["pose0"][3:])] * 4]))
            assert ok, (f, info)
        else:
            assert np.array_equal(poses[f], pr["pose0"])
    # launching again restarts from the uploaded poses: same answer (resident-problem path used by bench.py)
    po.launch()
    ninl2, poses2, outs2, _ = po.fetch()
    assert np.array_equal(ninl, ninl2) and np.array_equal(poses, poses2)
    po.close()
    # one-shot entry point
    n1, pose1, out1, st1 = api.pose_optimize(probs[0])
    assert n1 == ninl[0] and np.array_equal(out1, outs[0]) and np.array_equal(pose1, poses[0])


def test_pose_direct_call_leaves_the_resident_batch_alone():
    """a small cms_pose_optimize_batch (the direct path through a pinned block of its own) between cms_pose_launch and cms_pose_fetch of
    a resident batch on the SAME handle must neither drop the resident batch ("nothing launched") nor disturb its result."""
    probs = [synth.pose_problem(N=n, seed=s, outlier_frac=0.1) for n, s in ((600, 21), (300, 22), (900, 23), (450, 24), (120, 25), (700, 26), (64, 27), (500, 28), (333, 29), (256, 30))]
    po = api.PoseOptimizer(len(probs), sum(len(p["Xw"]) for p in probs))
    ninl0, poses0, outs0, _ = po.optimize(probs)
    po.upload(probs); po.launch()                      # asynchronous: the batch is in flight / resident
    small = [synth.pose_problem(N=200, seed=41, outlier_frac=0.2), synth.pose_problem(N=350, seed=42, outlier_frac=0.0)]
    n_s, poses_s, outs_s, _ = po.optimize_batch(small)   # nf <= 8: the direct path
    for f, pr in enumerate(small):
        w_n, w_pose, w_out, _ = orc.pose_optimize(pr)
        assert n_s[f] == w_n and np.array_equal(outs_s[f], w_out)
    ninl1, poses1, outs1, _ = po.fetch()               # the resident batch's results are still there
    assert np.array_equal(ninl0, ninl1) and np.array_equal(poses0, poses1) and all(np.array_equal(a, b) for a, b in zip(outs0, outs1))
    po.close()


@pytest.mark.parametrize("views", ["track", "random"])
def test_ba_deterministic_mode_is_bit_repeatable_and_matches_the_oracle(views):
    """cms_ba_set_deterministic(1): windows created afterwards run the fixed-order kernels -- bit-identical poses, points and flags run after run
    (the reference's g2o is single-threaded and so deterministic), at the parity bar of the default path; windows of both kinds in ONE
    cms_ba_optimize_many call run as separate groups and each gives its own kind's result; switching the mode off restores the default path."""
    prob = synth.ba_problem(K=12, P=6000, obs_per_point=4, F=550, seed=91, views=views)
    w = orc.ba_run(prob)
    assert not api.ba_get_deterministic()
    api.ba_set_deterministic(True)
    try:
        assert api.ba_get_deterministic()
        runs = [api.ba_run(prob) for _ in range(4)]
        det = api.BundleAdjuster(prob)
    finally:
        api.ba_set_deterministic(False)
    for g in runs[1:]:
        assert np.array_equal(g["poses"], runs[0]["poses"]) and np.array_equal(g["points"], runs[0]["points"]) and np.array_equal(g["outliers"], runs[0]["outliers"])
    g = runs[0]
    assert list(g["stats"].iterations_done) == list(w["stats"].iterations_done) and np.array_equal(g["outliers"], w["outliers"])
    _ba_updates_close_or_cascade(prob, g["poses"], g["points"], w, tag="deterministic " + views)
    # a mixed call: one window created under the mode, one after it was switched off
    dflt = api.BundleAdjuster(prob)
    _, stats = api.ba_optimize_many([det, dflt], (5, 10))
    o_det, o_dflt = det.read(), dflt.read()
    assert np.array_equal(o_det[0], runs[0]["poses"]) and np.array_equal(o_det[1], runs[0]["points"])      # the deterministic window: the same bits again
    for st in stats:
        assert list(st.iterations_done) == list(w["stats"].iterations_done)
    _ba_updates_close_or_cascade(prob, o_dflt[0], o_dflt[1], w, tag="default next to deterministic " + views)
    det.close(); dflt.close()


def test_ba_deterministic_windows_repeat_twenty_times_whatever_their_company():
    """The deterministic mode's contract on the windows bench.py optimises (K = 20, 80 k observations, tracked views -- signature runs through
    kb_ba_lin_schur_runs_det -- and random views -- every point through kb_ba_lin_schur_edges_det): TWENTY runs of a window give the same bits
    (poses, points, outlier flags); a window gives those bits again when it is optimised in one cms_ba_optimize_many call with other deterministic
    windows (how a window's chunks are cut into workgroups does not depend on the group: BA_DET_RANGES), when its group call creates it
    (cms_ba_create_many, the plan kernel included) and whatever order the group lists its windows in; iteration counts and flags are the oracle's."""
    probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=77, views="track"),
             synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=78, views="random"),
             synth.ba_problem(K=12, P=6000, obs_per_point=4, F=550, seed=91, views="track"),
             synth.ba_problem(K=20, P=9000, obs_per_point=5, F=550, seed=79, views="track")]
    api.ba_set_deterministic(True)
    try:
        alone = [api.ba_run(p) for p in probs]
        for i in (0, 1):
            for r in range(19):
                g = api.ba_run(probs[i])
                assert np.array_equal(g["poses"], alone[i]["poses"]) and np.array_equal(g["points"], alone[i]["points"]) and \
                    np.array_equal(g["outliers"], alone[i]["outliers"]), (i, r)
        for order in ((0, 1, 2, 3), (3, 0, 2, 1), (2, 2, 0)):
            bas = [api.BundleAdjuster(probs[i]) for i in order]
            api.ba_optimize_many(bas, (5, 10))
            for i, ba in zip(order, bas):
                o = ba.read()
                assert np.array_equal(o[0], alone[i]["poses"]) and np.array_equal(o[1], alone[i]["points"]), (order, i)
                ba.close()
        grp = api.ba_create_many([dict(probs[i], _plan_on_device=True) for i in (0, 3, 0)])
        api.ba_optimize_many(list(grp), (5, 10))
        for i, o in zip((0, 3, 0), api.ba_read_many(grp)):
            assert np.array_equal(o[0], alone[i]["poses"]) and np.array_equal(o[1], alone[i]["points"]), i
        for ba in grp:
            ba.close()
    finally:
        api.ba_set_deterministic(False)
    for p, g in zip(probs[:2], alone[:2]):
        w = orc.ba_run(p)
        assert list(g["stats"].iterations_done) == list(w["stats"].iterations_done) and np.array_equal(g["outliers"], w["outliers"])
        _ba_updates_close_or_cascade(p, g["poses"], g["points"], w, tag="deterministic K=20")


def test_ba_deterministic_mode_with_a_workgroup_count_of_the_callers_choice():
    """cms_ba_set_deterministic(n >= 2): windows created afterwards are cut into n workgroups (a host that optimises one window per call passes 64, the bridge does).
    The count belongs to the window: three runs give the same bits; windows of two counts in ONE cms_ba_optimize_many call run as groups of their own and each
    gives the bits it gives alone; iteration counts and flags are the oracle's."""
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=77, views="track")
    try:
        api.ba_set_deterministic(64)
        assert api.ba_get_deterministic()
        runs64 = [api.ba_run(prob) for _ in range(3)]
        w64 = api.BundleAdjuster(prob)
        api.ba_set_deterministic(True)
        run16 = api.ba_run(prob)
        w16 = api.BundleAdjuster(prob)
    finally:
        api.ba_set_deterministic(False)
    for g in runs64[1:]:
        assert np.array_equal(g["poses"], runs64[0]["poses"]) and np.array_equal(g["points"], runs64[0]["points"]) and np.array_equal(g["outliers"], runs64[0]["outliers"])
    api.ba_optimize_many([w16, w64], (5, 10))
    o16, o64 = w16.read(), w64.read()
    assert np.array_equal(o64[0], runs64[0]["poses"]) and np.array_equal(o64[1], runs64[0]["points"])
    assert np.array_equal(o16[0], run16["poses"]) and np.array_equal(o16[1], run16["points"])
    w16.close(); w64.close()
    w = orc.ba_run(prob)
    for g in (runs64[0], run16):
        assert list(g["stats"].iterations_done) == list(w["stats"].iterations_done) and np.array_equal(g["outliers"], w["outliers"])
        _ba_updates_close_or_cascade(prob, g["poses"], g["points"], w, tag="deterministic, chosen workgroup count")


def test_ba_deterministic_mode_on_windows_of_every_solver_path():
    """Deterministic windows the fixed-order fused chain does not take fall back to deterministic kernels of their own -- 27 and 29 free key frames (no
    three-lane solve: the pair-owner kernel on the host's full plan), 39 (register-tiled scalar solve) -- and small or odd windows (two key frames, a
    single free one, 25 free = the three-lane solve's limit, fewer wavefronts per workgroup) run the fixed-order kernels: three runs give the same bits,
    iteration counts and flags are the oracle's, the estimates pass the default path's bar."""
    shapes = [dict(K=2, P=120, obs_per_point=2, seed=7), dict(K=3, P=300, obs_per_point=3, seed=8), dict(K=26, P=1500, obs_per_point=5, seed=16),
              dict(K=28, P=1500, obs_per_point=5, seed=3), dict(K=30, P=2000, obs_per_point=5, seed=5), dict(K=40, P=2500, obs_per_point=6, seed=4),
              dict(K=23, P=4000, obs_per_point=4, seed=31, views="track"), dict(K=25, P=5000, obs_per_point=4, seed=32, views="track")]
    for sh in shapes:
        prob = synth.ba_problem(F=550, **sh)
        api.ba_set_deterministic(True)
        try:
            runs = [api.ba_run(prob) for _ in range(3)]
        finally:
            api.ba_set_deterministic(False)
        for g in runs[1:]:
            assert np.array_equal(g["poses"], runs[0]["poses"]) and np.array_equal(g["points"], runs[0]["points"]) and np.array_equal(g["outliers"], runs[0]["outliers"]), sh
        w = orc.ba_run(prob)
        g = runs[0]
        assert list(g["stats"].iterations_done) == list(w["stats"].iterations_done) and np.array_equal(g["outliers"], w["outliers"]), sh
        _ba_updates_close_or_cascade(prob, g["poses"], g["points"], w, tag="deterministic %r" % (sh,))


def test_c_abi_error_paths():
    """the C-ABI reports misuse with a status code and a message instead of crashing or silently truncating"""
    import ctypes as C
    L = api.lib()
    camd = synth.camera("lafida", 150)
    ctx = api.Context(camd, nfeatures=800, max_batch=2)
    ctx.set_mask(np.full((450, 450), 255, np.uint8))
    fish = synth.texture(camd["Ih"], camd["Iw"], 3)
    k, d = ctx.remap_extract(fish)
    assert len(k) > 50
    # caller capacity too small -> CMS_ERR_OVERFLOW (-4), count still reported
    kp = np.zeros(8, api.KP_DTYPE); ds = np.zeros((8, 32), np.uint8); n = C.c_int(0)
    rc = L.cms_frames_fetch(ctx.h, 0, kp.ctypes.data_as(C.c_void_p), ds.ctypes.data_as(C.c_void_p), 8, C.byref(n))
    assert rc == -4 and n.value == len(k) and b"capacity" in L.cms_last_error()
    # batch larger than the context was created for / bad frame index -> CMS_ERR_ARG (-1)
    assert L.cms_frames_process(ctx.h, 3, 1) == -1
    assert L.cms_frames_fetch(ctx.h, 5, None, None, 0, C.byref(n)) == -1
    ctx.close()
    # BA: edge referring to a key frame that does not exist, face id out of range
    prob = synth.ba_problem(K=4, P=50, obs_per_point=3, F=550, seed=2)
    bad = dict(prob); bad["e_pose"] = prob["e_pose"].copy(); bad["e_pose"][0] = 99
    with pytest.raises(api.CmsError):
        api.BundleAdjuster(bad)
    bad = dict(prob); bad["e_face"] = prob["e_face"].copy(); bad["e_face"][3] = 5
    with pytest.raises(api.CmsError):
        api.BundleAdjuster(bad)
    # pose optimisation: more edges than the handle holds, unknown face (the reference exits the process there)
    pp = synth.pose_problem(N=100, seed=3)
    po = api.PoseOptimizer(1, 50)
    with pytest.raises(api.CmsError):
        po.upload([pp])
    po.close()
    bad = dict(pp); bad["face"] = pp["face"].copy(); bad["face"][0] = 7
    with pytest.raises(api.CmsError):
        api.pose_optimize(bad)


def test_features_in_area_matches_oracle():
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea on the device: identical candidate lists, order included,
    for queries biased to face edges and corners (all 41 unfolding cases), with level filters; host and device-pointer entry."""
    import test_area_emu as te
    for F, seed in ((550, 21), (150, 22)):
        camd = synth.camera("lafida", F)
        ocam = orc.make_camera(camd)
        ctx = api.Context(camd, nfeatures=2000, max_batch=2)
        kx, ky, ko = te._keypoints(F, 2000, seed)
        kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko
        ctx.area_set_keypoints(1, kps)
        ctx.area_set_keypoints(0, kps[:7])
        ctx.area_grid(2)
        qx, qy, qr, lo, hi, _ = te._queries(F, 5000, 30 + seed)
        want_off, want_idx = orc.features_in_area(ocam, kx, ky, ko, qx, qy, qr, lo, hi)
        got_off, got_idx = ctx.features_in_area(1, qx, qy, qr, lo, hi)
        assert np.array_equal(got_off, want_off) and np.array_equal(got_idx, want_idx), (F, len(got_idx), len(want_idx))
        assert len(want_idx) > 2000
        # the other frame slot has its own (tiny) grid
        w0, i0 = orc.features_in_area(ocam, kx[:7], ky[:7], ko[:7], qx, qy, qr, lo, hi)
        g0, j0 = ctx.features_in_area(0, qx, qy, qr, lo, hi)
        assert np.array_equal(g0, w0) and np.array_equal(j0, i0)
        # capacity too small -> CMS_ERR_OVERFLOW
        with pytest.raises(api.CmsError):
            ctx.features_in_area(1, qx, qy, qr, lo, hi, cap=10)
        ctx.close()


def test_features_in_area_on_extracted_frames_feeds_the_matcher():
    """extract two frames, build their grids on the device, generate the SearchByProjection windows of frame 0's key points in
    frame 1 on the device and match -- no candidate list ever built on the host; equals oracle lists + oracle Hamming scan."""
    import torch
    camd = synth.camera("lafida", 250)
    ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=1000, max_batch=2)
    mask = synth.cubemap_valid_mask(camd, erode=5, band=30)
    ctx.set_mask(mask)
    big = synth.texture(camd["Ih"] + 16, camd["Iw"] + 16, 4)
    frames = np.stack([big[:camd["Ih"], :camd["Iw"]], big[2:2 + camd["Ih"], 3:3 + camd["Iw"]]]).copy()
    ctx.upload(frames); ctx.process(2, True); ctx.sync()
    (k0, d0), (k1, d1) = ctx.fetch(0), ctx.fetch(1)
    assert len(k0) > 200 and len(k1) > 200
    ctx.area_grid(2)
    scales = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    qx = k0["x"].copy(); qy = k0["y"].copy(); qr = (np.float32(15.0) * scales[k0["octave"]]).astype(np.float32)
    lo = (k0["octave"] - 1).astype(np.int32); hi = (k0["octave"] + 1).astype(np.int32)
    want_off, want_idx = orc.features_in_area(ocam, k1["x"], k1["y"], k1["octave"], qx, qy, qr, lo, hi)
    nq, cap, kp_cap = len(qx), len(want_idx) + 64, ctx.geom.kp_cap
    dev = torch.device("cuda", 0)
    dq = [torch.from_numpy(a).to(dev) for a in (qx, qy, qr, lo, hi)]
    d_cnt = torch.zeros(nq, dtype=torch.int32, device=dev); d_off = torch.zeros(nq + 1, dtype=torch.int32, device=dev)
    d_idx = torch.zeros(cap, dtype=torch.int32, device=dev); d_tot = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.features_in_area_device(1, nq, [t.data_ptr() for t in dq], d_cnt.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), cap, kp_cap, d_tot.data_ptr())
    ctx.sync()
    assert int(d_tot.item()) == len(want_idx)
    assert np.array_equal(d_off.cpu().numpy(), want_off) and np.array_equal(d_idx.cpu().numpy()[:len(want_idx)], want_idx + kp_cap)
    # ... and straight into the Hamming scan on the device (rows of frame 0 are the queries, candidates are rows of frame 1)
    _, d_desc, _ = ctx.results_ptrs()
    d_qrow = torch.arange(nq, dtype=torch.int32, device=dev)
    d_lvl = torch.zeros(2 * kp_cap, dtype=torch.int32, device=dev)
    d_lvl[kp_cap:kp_cap + len(k1)] = torch.from_numpy(k1["octave"].astype(np.int32)).to(dev)
    outs = [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(5)]
    ctx.hamming_best2_device(d_desc, d_qrow.data_ptr(), nq, d_desc, d_off.data_ptr(), d_idx.data_ptr(), d_lvl.data_ptr(), None, [o.data_ptr() for o in outs])
    ctx.sync()
    alld = np.zeros((2 * kp_cap, 32), np.uint8); alld[:len(d0)] = d0; alld[kp_cap:kp_cap + len(d1)] = d1
    lvl = np.zeros(2 * kp_cap, np.int32); lvl[kp_cap:kp_cap + len(k1)] = k1["octave"]
    want = orc.hamming_best2(alld[:nq], alld, want_off, (want_idx + kp_cap).astype(np.int32), lvl)
    assert np.array_equal(outs[0].cpu().numpy(), want["best_idx"]) and np.array_equal(outs[1].cpu().numpy(), want["best_dist"])
    assert np.array_equal(outs[3].cpu().numpy(), want["second_dist"])
    assert (want["best_idx"] >= 0).sum() > 100
    ctx.close()


def _local_map_case(F, n, seed, order):
    import test_area_emu as te
    kx, ky, ko = te._keypoints(F, n, seed)
    kd = synth.descriptors(len(kx), seed + 1)
    pr = synth.local_map_problem(F, kx, ky, ko, kd, seed=seed + 2, order=order)
    return kx, ky, ko, kd, pr


def test_search_local_points_matches_oracle():
    """Tracking::SearchLocalPoints on the device: Frame::isInFrustum for every map point, the windows, and the
    sequential greedy of ORBMatcher::SearchByProjection(F, vpMapPoints, th) reproduced by parallel rounds -- identical in-view flags,
    projections, predicted levels, matches and key-point ownership, for random and for spatially sorted (long claim chains) lists."""
    # the last case has windows of hundreds of candidates: the 64-per-window first guess overflows and the entry repeats the query
    for F, n, seed, order, th in ((550, 2000, 41, "random", 1.0), (550, 2000, 42, "spatial", 5.0), (150, 600, 43, "random", 5.0), (150, 1900, 44, "random", 14.0)):
        camd = synth.camera("lafida", F)
        ocam = orc.make_camera(camd)
        kx, ky, ko, kd, pr = _local_map_case(F, n, seed, order)
        ctx = api.Context(camd, nfeatures=2000, max_batch=2)
        kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko
        ctx.area_set_keypoints(1, kps); ctx.area_set_descriptors(1, kd)
        ctx.area_set_keypoints(0, kps[:5]); ctx.area_set_descriptors(0, kd[:5])
        ctx.area_grid(2)
        fr = orc.is_in_frustum(ocam, pr["pose15"], pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"])
        taken = np.full(len(kx), -1, np.int32); taken[::9] = 10**6
        want_kp = taken.copy()
        want, nm = orc.search_local_points(ocam, kx, ky, ko, kd, pr["scale_factors"], fr, pr["desc"], want_kp, th=th)
        got_kp = taken.copy()
        got = ctx.search_local_points(1, pr["pose15"], pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], got_kp, th=th)
        assert np.array_equal(got["in_view"], fr["in_view"]), (F, order)
        for k in ("proj_x", "proj_y", "view_cos"):
            assert np.array_equal(got[k], fr[k]), (F, order, k)
        assert np.array_equal(got["level"], fr["level"])
        assert np.array_equal(got["match"], want) and got["n_matches"] == nm, (F, order, got["n_matches"], nm)
        assert np.array_equal(got_kp, want_kp)
        assert nm > 100 and 1 <= got["rounds"] <= len(want)
        # conflicts really occurred: some map point lost its best key point to an earlier one
        ctx.close()


def test_search_local_points_batch_on_extracted_frames():
    """the device-pointer entries over a whole batch: three extracted frames, each with its own pose and local map; projection, windows
    and greedy search run back to back on the ctx stream without a host round trip; every frame equals the oracle run on its own."""
    import torch
    camd = synth.camera("lafida", 250)
    ocam = orc.make_camera(camd)
    B = 3
    ctx = api.Context(camd, nfeatures=1000, max_batch=B)
    ctx.set_mask(synth.cubemap_valid_mask(camd, erode=5, band=30))
    big = synth.texture(camd["Ih"] + 16, camd["Iw"] + 16, 9)
    frames = np.stack([big[dy:dy + camd["Ih"], dx:dx + camd["Iw"]] for dy, dx in ((0, 0), (3, 5), (7, 2))]).copy()
    ctx.upload(frames); ctx.process(B, True); ctx.sync()
    fetched = [ctx.fetch(b) for b in range(B)]
    ctx.area_grid(B)
    kp_cap = ctx.geom.kp_cap
    probs = [synth.local_map_problem(250, k["x"], k["y"], k["octave"], d, seed=60 + b, order=("spatial" if b == 1 else "random"))
             for b, (k, d) in enumerate(fetched)]
    mp_off = np.concatenate([[0], np.cumsum([len(p["pos"]) for p in probs])]).astype(np.int32)
    nmp = int(mp_off[-1])
    dev = torch.device("cuda", 0)
    cat = lambda key, dt: torch.from_numpy(np.concatenate([p[key] for p in probs]).astype(dt)).to(dev)
    d_pose = torch.from_numpy(np.stack([p["pose15"] for p in probs])).to(dev)
    d_frame = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), np.diff(mp_off))).to(dev)
    d_pos, d_nrm, d_min, d_max, d_desc = cat("pos", np.float32), cat("normal", np.float32), cat("min_dist", np.float32), cat("max_dist", np.float32), cat("desc", np.uint8)
    d_vis = torch.zeros(nmp, dtype=torch.uint8, device=dev)
    d_px, d_py, d_vc, d_qr = (torch.zeros(nmp, dtype=torch.float32, device=dev) for _ in range(4))
    d_lvl, d_qmin, d_qmax, d_cnt, d_match = (torch.zeros(nmp, dtype=torch.int32, device=dev) for _ in range(5))
    d_off = torch.zeros(nmp + 1, dtype=torch.int32, device=dev); d_tot = torch.zeros(1, dtype=torch.int32, device=dev)
    cap = 64 * nmp
    d_idx = torch.zeros(cap, dtype=torch.int32, device=dev); d_pd = torch.zeros(cap, dtype=torch.int16, device=dev)
    taken = [np.full(kp_cap, -1, np.int32) for _ in range(B)]
    for b in range(B):
        taken[b][:len(fetched[b][0]):11] = 10**6
    d_kpmp = torch.from_numpy(np.concatenate(taken)).to(dev)
    d_mpoff = torch.from_numpy(mp_off).to(dev); d_rounds = torch.zeros(B, dtype=torch.int32, device=dev)
    th = 5.0
    ctx.is_in_frustum_device(nmp, d_frame.data_ptr(), d_pose.data_ptr(), d_pos.data_ptr(), d_nrm.data_ptr(), d_min.data_ptr(), d_max.data_ptr(), 0.5, th,
                             [t.data_ptr() for t in (d_vis, d_px, d_py, d_lvl, d_vc)], [t.data_ptr() for t in (d_qr, d_qmin, d_qmax)])
    ctx.features_in_area_batch_device(nmp, d_frame.data_ptr(), [t.data_ptr() for t in (d_px, d_py, d_qr, d_qmin, d_qmax)], d_cnt.data_ptr(),
                                      d_off.data_ptr(), d_idx.data_ptr(), cap, d_tot.data_ptr())
    ctx.search_local_points_device(B, d_mpoff.data_ptr(), d_desc.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), d_pd.data_ptr(), 0.8, 100,
                                   d_kpmp.data_ptr(), d_match.data_ptr(), d_rounds.data_ptr())
    ctx.sync()
    assert int(d_tot.item()) <= cap
    match = d_match.cpu().numpy(); kpmp = d_kpmp.cpu().numpy(); vis = d_vis.cpu().numpy()
    total = 0
    for b in range(B):
        k, d = fetched[b]; p = probs[b]
        fr = orc.is_in_frustum(ocam, p["pose15"], p["pos"], p["normal"], p["min_dist"], p["max_dist"])
        want_kp = taken[b][:len(k)].copy()
        want, nm = orc.search_local_points(ocam, k["x"], k["y"], k["octave"], d, p["scale_factors"], fr, p["desc"], want_kp, th=th)
        sl = slice(mp_off[b], mp_off[b + 1])
        assert np.array_equal(vis[sl], fr["in_view"])
        assert np.array_equal(d_px.cpu().numpy()[sl], fr["proj_x"]) and np.array_equal(d_lvl.cpu().numpy()[sl], fr["level"])
        got = match[sl]
        assert np.array_equal(np.where(got >= 0, got - b * kp_cap, -1), want), b
        # ownership is written with the global map-point index
        got_kp = kpmp[b * kp_cap: b * kp_cap + len(k)]
        assert np.array_equal(np.where((got_kp >= 0) & (got_kp < 10**6), got_kp - mp_off[b], got_kp), want_kp), b
        total += nm
    assert total > 300
    assert (d_rounds.cpu().numpy() >= 1).all()
    ctx.close()


def _keyframes(F, n_kf, n_pts, seed):
    ocam = orc.make_camera(synth.camera("lafida", F))
    S = synth.keyframe_set(F, n_kf=n_kf, n_pts=n_pts, seed=seed)
    oks = [orc.make_keyframe(ocam, k) for k in S["kfs"]]          # fills k["rays"] too
    gks = [api.make_keyframe(k) for k in S["kfs"]]
    return ocam, S, oks, gks


def test_create_new_map_points_matches_oracle():
    """LocalMapping::CreateNewMapPoints on the device: SearchForTriangulation over the FeatureVectors with the
    epipolar gate, ray triangulation (4x4 Jacobi SVD in float) and all acceptance tests, neighbours in sequence; two independent
    jobs in one launch.  Same new points in the same order, coordinates bit-identical."""
    F = 550
    camd = synth.camera("lafida", F)
    ctx = api.Context(camd, nfeatures=2000, max_batch=1)
    jobs_g, want = [], []
    keep = []
    for seed, n_kf in ((5, 5), (6, 3)):
        ocam, S, oks, gks = _keyframes(F, n_kf, 2400, seed)
        keep.append((S, oks, gks))
        cur_mp = S["kfs"][0]["mp"].copy()
        want.append(orc.create_new_map_points(ocam, oks[0][0], [k for k, _ in oks[1:]], S["scale_factors"], S["level_sigma2"], cur_mp))
        jobs_g.append((gks[0][0], [k for k, _ in gks[1:]]))
    got = api.create_new_map_points(ctx, jobs_g)
    for j in range(2):
        wn, w1, w2, wx = want[j]
        gn, g1, g2, gx = got[j]
        assert len(wn) > 200, len(wn)
        assert np.array_equal(gn, wn) and np.array_equal(g1, w1) and np.array_equal(g2, w2), (j, len(gn), len(wn))
        assert np.array_equal(gx.view(np.uint32), wx.view(np.uint32)), (j, np.abs(gx - wx).max())
        assert len(np.unique(wn)) >= 2                                # several neighbours contributed
    # the neighbour-after-neighbour kernel (used when the rotation histogram is on) gives the same records
    os.environ["CMS_TRI_SEQUENTIAL"] = "1"
    try:
        got_seq = api.create_new_map_points(ctx, jobs_g)
    finally:
        del os.environ["CMS_TRI_SEQUENTIAL"]
    for j in range(2):
        for x, y in zip(got_seq[j], got[j]):
            assert np.array_equal(x, y)
    # orientation check on (ORBMatcher(.., true)) only changes the search: compare the pair search through a one-neighbour job
    S, oks, gks = keep[0]
    E = orc.compute_e12(S["kfs"][0], S["kfs"][1])
    m, _ = orc.search_for_triangulation(ocam, oks[0][0], oks[1][0], E, S["scale_factors"], S["level_sigma2"], check_ori=True)
    got1 = api.create_new_map_points(ctx, [(gks[0][0], [gks[1][0]])], check_orientation=True)[0]
    assert set(got1[1]) <= set(np.flatnonzero(m >= 0)) and len(got1[1]) > 50
    assert np.array_equal(got1[2], m[got1[1]])
    # capacity too small
    with pytest.raises(api.CmsError):
        api.create_new_map_points(ctx, jobs_g, cap=10)
    # resident key frames: same records from slots; then a pose / map-point update changes the outcome like the oracle says
    store = api.KeyframeStore(ctx, max_keyframes=16, max_features=2048, max_nodes=1024)
    slot = 0
    slot_jobs = []
    for S, oks, gks in keep:
        ids = []
        for K, _ in gks:
            store.put(slot, K); ids.append(slot); slot += 1
        slot_jobs.append((ids[0], ids[1:]))
    got_res = store.create_new_map_points(slot_jobs)
    for j in range(2):
        for x, y in zip(got_res[j], got[j]):
            assert np.array_equal(x, y)
    S, oks, gks = keep[1]
    new_mp = S["kfs"][0]["mp"].copy(); new_mp[::3] = 7
    k1 = dict(S["kfs"][0]); k1["mp"] = new_mp
    t2 = (S["kfs"][1]["t"] * np.float32(1.5)).astype(np.float32)
    k2 = dict(S["kfs"][1]); k2["t"] = t2; k2["Ow"] = (-(k2["R"].astype(np.float64).T @ t2.astype(np.float64))).astype(np.float32)
    store.update(slot_jobs[1][0], mp=new_mp)
    store.update(slot_jobs[1][1][0], t=t2, Ow=k2["Ow"])
    O1, _k1 = orc.make_keyframe(ocam, k1); O2, _k2 = orc.make_keyframe(ocam, k2)
    cur_mp = new_mp.copy()
    w = orc.create_new_map_points(ocam, O1, [O2] + [k for k, _ in oks[2:]], S["scale_factors"], S["level_sigma2"], cur_mp)
    g = store.create_new_map_points([slot_jobs[1]])[0]
    assert len(w[0]) > 50 and all(np.array_equal(x, y) for x, y in zip(g[:3], w[:3])) and np.array_equal(g[3].view(np.uint32), w[3].view(np.uint32))
    with pytest.raises(api.CmsError):
        store.create_new_map_points([(15, [0])])            # empty slot
    store.close()
    ctx.close()


def test_search_for_triangulation_matches_oracle():
    """cms_search_for_triangulation (ORBMatcher::SearchForTriangulation alone, ORBMatcher.cpp:971-1125, the call of LocalMapping.cpp:254) against the
    oracle: match lists with the caller's E12 and with the library's ComputeE12, with and without the rotation histogram."""
    F = 450
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=2000, max_batch=1)
    ks = synth.keyframe_set(F, n_kf=4, n_pts=2400, seed=41)
    oks = [orc.make_keyframe(ocam, q) for q in ks["kfs"]]
    gks = [api.make_keyframe(q) for q in ks["kfs"]]
    total = 0
    for j in (1, 2, 3):
        E12 = orc.compute_e12(ks["kfs"][0], ks["kfs"][j])
        for check in (False, True):
            want, wn = orc.search_for_triangulation(ocam, oks[0][0], oks[j][0], E12, ks["scale_factors"], ks["level_sigma2"], check)
            got, gn = api.search_for_triangulation(ctx, gks[0][0], gks[j][0], E12, check)
            assert gn == wn and np.array_equal(got, want), (j, check, gn, wn, int((got != want).sum()))
            got2, gn2 = api.search_for_triangulation(ctx, gks[0][0], gks[j][0], None, check)      # ComputeE12 inside the library
            assert gn2 == wn and np.array_equal(got2, want), (j, check, "library E12", gn2, wn)
            total += wn
    assert total > 600
    ctx.close()


def test_fuse_search_matches_oracle():
    """search half of ORBMatcher::Fuse: the key point every map point would be fused with"""
    import test_area_emu as te
    F = 550
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    kx, ky, ko = te._keypoints(F, 2000, 81)
    kd = synth.descriptors(len(kx), 82)
    pr = synth.local_map_problem(F, kx, ky, ko, kd, seed=83)
    ctx = api.Context(camd, nfeatures=2000, max_batch=1)
    kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko
    ctx.area_set_keypoints(0, kps); ctx.area_set_descriptors(0, kd); ctx.area_grid(1)
    kf = dict(x=kx, y=ky, octave=ko, angle=np.zeros(len(kx), np.float32), desc=kd, mp=np.full(len(kx), -1, np.int32),
              R=pr["pose15"][:9], t=pr["pose15"][9:12], Ow=pr["pose15"][12:], node_id=np.zeros(0, np.int32), node_off=np.zeros(1, np.int32),
              node_feat=np.zeros(0, np.int32), median_depth=1.0, rays=np.zeros((len(kx), 3), np.float32))
    K, _keep = orc.make_keyframe(ocam, kf)
    skip = (np.arange(len(pr["pos"])) % 13 == 0).astype(np.uint8)
    sf = pr["scale_factors"]; inv_s2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)
    for th in (3.0, 8.0):
        wi, wd = orc.fuse_search(ocam, K, skip, pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], th, sf, inv_s2)
        gi, gd = api.fuse_search(ctx, 0, pr["pose15"], skip, pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], th)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd), (th, (gi != wi).sum())
        assert (wi >= 0).sum() > 300 and (wi[skip > 0] == -1).all()
    ctx.close()


def test_tracking_and_mapping_edge_cases():
    """empty and degenerate inputs of the local-map search, CreateNewMapPoints and Fuse entries behave like the reference's loops:
    nothing to do -> nothing returned, no error."""
    import test_area_emu as te
    F = 250
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=1000, max_batch=1)
    kx, ky, ko = te._keypoints(F, 500, 5)
    kd = synth.descriptors(len(kx), 6)
    kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko
    pr = synth.local_map_problem(F, kx, ky, ko, kd, seed=7)
    # no map points
    ctx.area_set_keypoints(0, kps); ctx.area_set_descriptors(0, kd); ctx.area_grid(1)
    kp_mp = np.full(len(kx), -1, np.int32)
    r = ctx.search_local_points(0, pr["pose15"], np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                                np.zeros((0, 32), np.uint8), kp_mp)
    assert r["n_matches"] == 0 and len(r["match"]) == 0
    # every key point already taken -> no match, ownership untouched
    taken = np.full(len(kx), 99, np.int32)
    r = ctx.search_local_points(0, pr["pose15"], pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], taken)
    assert r["n_matches"] == 0 and (r["match"] == -1).all() and (taken == 99).all() and r["in_view"].sum() > 50
    # all map points behind / out of range -> nothing in view
    far = pr["max_dist"] * np.float32(1e-3)
    r = ctx.search_local_points(0, pr["pose15"], pr["pos"], pr["normal"], far * np.float32(0.5), far, pr["desc"], kp_mp)
    fr = orc.is_in_frustum(ocam, pr["pose15"], pr["pos"], pr["normal"], far * np.float32(0.5), far)
    assert np.array_equal(r["in_view"], fr["in_view"]) and r["n_matches"] == 0
    # a frame without key points
    ctx.area_set_keypoints(0, kps[:0]); ctx.area_grid(1)
    r = ctx.search_local_points(0, pr["pose15"], pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], np.zeros(0, np.int32))
    fr = orc.is_in_frustum(ocam, pr["pose15"], pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"])
    assert r["n_matches"] == 0 and np.array_equal(r["in_view"], fr["in_view"])
    bi, bd = api.fuse_search(ctx, 0, pr["pose15"], np.zeros(len(pr["pos"]), np.uint8), pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], 3.0)
    assert (bi == -1).all() and (bd == 256).all()
    # Fuse with everything skipped
    ctx.area_set_keypoints(0, kps); ctx.area_set_descriptors(0, kd); ctx.area_grid(1)
    bi, bd = api.fuse_search(ctx, 0, pr["pose15"], np.ones(len(pr["pos"]), np.uint8), pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], 3.0)
    assert (bi == -1).all()
    # CreateNewMapPoints: no neighbours; a neighbour without features / without a FeatureVector; every feature already mapped; baseline too short
    S = synth.keyframe_set(F, n_kf=3, n_pts=900, seed=8)
    for k in S["kfs"]:
        k["rays"] = orc.keyframe_rays(ocam, k["x"], k["y"])
    g = [api.make_keyframe(k) for k in S["kfs"]]
    assert len(api.create_new_map_points(ctx, [(g[0][0], [])])[0][0]) == 0
    empty = dict(S["kfs"][1]); n0 = 0
    for key in ("x", "y", "octave", "angle", "desc", "rays", "mp"):
        empty[key] = empty[key][:n0]
    empty["node_id"] = np.zeros(0, np.int32); empty["node_off"] = np.zeros(1, np.int32); empty["node_feat"] = np.zeros(0, np.int32)
    ge = api.make_keyframe(empty)
    nofv = dict(S["kfs"][2]); nofv["node_id"] = np.zeros(0, np.int32); nofv["node_off"] = np.zeros(1, np.int32); nofv["node_feat"] = np.zeros(0, np.int32)
    gn = api.make_keyframe(nofv)
    res = api.create_new_map_points(ctx, [(g[0][0], [ge[0], gn[0], g[1][0]])])[0]
    oks = [orc.make_keyframe(ocam, k) for k in (S["kfs"][0], empty, nofv, S["kfs"][1])]
    w = orc.create_new_map_points(ocam, oks[0][0], [k for k, _ in oks[1:]], S["scale_factors"], S["level_sigma2"], S["kfs"][0]["mp"].copy())
    assert len(w[0]) > 20 and (w[0] == 2).all() and all(np.array_equal(a, b) for a, b in zip(res[:3], w[:3]))
    full = dict(S["kfs"][0]); full["mp"] = np.zeros(len(full["x"]), np.int32)
    gf = api.make_keyframe(full)
    assert len(api.create_new_map_points(ctx, [(gf[0], [g[1][0], g[2][0]])])[0][0]) == 0
    near = dict(S["kfs"][1]); near["median_depth"] = np.float32(1e4)          # baseline / depth < 0.01 -> neighbour skipped
    gnr = api.make_keyframe(near)
    assert len(api.create_new_map_points(ctx, [(g[0][0], [gnr[0]])])[0][0]) == 0
    # malformed FeatureVector is refused, not read out of bounds
    bad = dict(S["kfs"][1]); bad["node_feat"] = bad["node_feat"].copy(); bad["node_feat"][0] = 10**6
    gb = api.make_keyframe(bad)
    with pytest.raises(api.CmsError):
        api.create_new_map_points(ctx, [(g[0][0], [gb[0]])])
    ctx.close()


def test_map_point_bookkeeping_matches_oracle():
    """MapPoint::ComputeDistinctiveDescriptors and MapPoint::UpdateNormalAndDepth, batched over map points with 0 .. 70 observations:
    same chosen observation (first on ties) and bit-identical normal / distance range."""
    from test_oracle_tri import _map_point_batch
    camd = synth.camera("lafida", 250)
    ctx = api.Context(camd, nfeatures=500, max_batch=1)
    off, desc, pos, obs_Ow, ref_Ow, ref_level = _map_point_batch(51, npts=3000)
    want = orc.distinctive_descriptors(off, desc)
    got = api.distinctive_descriptors(ctx, off, desc)
    assert np.array_equal(got, want) and (want >= 0).sum() > 2000 and (want > 0).sum() > 500
    sf = np.array([ctx.geom.scale[l] for l in range(8)], np.float32)
    wn, wmn, wmx = orc.update_normal_and_depth(off, pos, obs_Ow, ref_Ow, ref_level, sf)
    gn = np.zeros((len(pos), 3), np.float32); gmn = np.zeros(len(pos), np.float32); gmx = np.zeros(len(pos), np.float32)
    api.update_normal_and_depth(ctx, off, pos, obs_Ow, ref_Ow, ref_level, gn, gmn, gmx)
    assert np.array_equal(gn.view(np.uint32), wn.view(np.uint32)) and np.array_equal(gmn.view(np.uint32), wmn.view(np.uint32))
    assert np.array_equal(gmx.view(np.uint32), wmx.view(np.uint32))
    # nothing to do / bad lists
    assert len(api.distinctive_descriptors(ctx, np.zeros(1, np.int32), np.zeros((0, 32), np.uint8))) == 0
    with pytest.raises(api.CmsError):
        api.distinctive_descriptors(ctx, np.array([0, 5, 3], np.int32), desc[:5])
    ctx.close()


def test_search_by_projection_frames_matches_oracle():
    """ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) whole on the device: projection with the current pose, windows,
    greedy best match (parallel rounds), rotation histogram -- identical matches and key-point ownership, histogram on and off."""
    import test_area_emu as te
    for F, n, seed in ((550, 2000, 91), (250, 800, 92), (150, 1900, 93)):
        camd = synth.camera("lafida", F)
        ocam = orc.make_camera(camd)
        kx, ky, ko = te._keypoints(F, n, seed)
        kd = synth.descriptors(len(kx), seed + 1)
        ka = np.random.default_rng(seed + 2).uniform(0, 360, len(kx)).astype(np.float32)
        pr = synth.motion_model_problem(F, kx, ky, ko, ka, kd, seed=seed + 3)
        ctx = api.Context(camd, nfeatures=2000, max_batch=2)
        kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko; kps["angle"] = ka
        ctx.area_set_keypoints(1, kps); ctx.area_set_descriptors(1, kd)
        ctx.area_set_keypoints(0, kps[:3]); ctx.area_set_descriptors(0, kd[:3])
        ctx.area_grid(2)
        for check, th in ((True, 15.0), (False, 7.0), (True, 60.0)):
            taken = np.full(len(kx), -1, np.int32); taken[::8] = 10**6
            want_kp = taken.copy()
            want, nm = orc.search_by_projection_frames(ocam, pr["pose12"][:9], pr["pose12"][9:], kx, ky, ko, ka, kd, pr["scale_factors"], pr["valid"], pr["Xw"],
                                                       pr["octave"], pr["angle"], pr["desc"], want_kp, th=th, check_ori=check)
            got_kp = taken.copy()
            got, gn = ctx.search_by_projection(1, pr["pose12"], pr["valid"], pr["Xw"], pr["octave"], pr["angle"], pr["desc"], got_kp, th=th, check_ori=check)
            assert np.array_equal(got, want) and gn == nm, (F, check, (got != want).sum(), gn, nm)
            assert np.array_equal(got_kp, want_kp)
            assert nm > 100
        ctx.close()


def test_search_for_initialization_matches_oracle():
    """ORBMatcher::SearchForInitialization (ORBMatcher.cpp:676-794) on the device against the oracle: match lists, counts and the updated
    vbPrevMatched, with and without the rotation histogram, on synthetic pairs with competing near-duplicates (take-over path) and on
    two consecutive EXTRACTED frames of the 3 x nFeatures initialisation extractor (Tracking.cpp:95-96, 145-148)."""
    import test_oracle_track as tot
    F = 350
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=2000, max_batch=2)
    for seed, check, rot, window in ((21, True, 20.0, 100), (22, False, 0.0, 100), (23, True, 0.0, 40), (24, True, 5.0, 100)):
        k1, d1, k2, d2 = tot._init_pair(F, 1200, seed, rot_deg=rot)
        ctx.area_set_keypoints(1, k2); ctx.area_set_descriptors(1, d2)
        ctx.area_set_keypoints(0, k2[:1]); ctx.area_set_descriptors(0, d2[:1])
        ctx.area_grid(2)
        prev_w = np.stack([k1["x"], k1["y"]], 1).astype(np.float32); prev_g = prev_w.copy()
        want_m, want_n = orc.search_for_initialization(ocam, k1, d1, k2, d2, prev_w, window, 0.9, check)
        got_m, got_n = ctx.search_for_initialization(1, k1, d1, prev_g, window, 0.9, check)
        assert got_n == want_n and np.array_equal(got_m, want_m), (seed, got_n, want_n, int((got_m != want_m).sum()))
        assert np.array_equal(prev_g.view(np.uint32), prev_w.view(np.uint32))
        assert want_n > 100
    ctx.close()
    # extracted frames, initialisation extractor (3 x nFeatures), frame 1 is frame 0 drifted by a few pixels
    F = 450
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    nf = 3 * camd["nfeatures"]
    ctx = api.Context(camd, nfeatures=nf, max_batch=2)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    big = synth.texture(camd["Ih"] + 16, camd["Iw"] + 16, 77)
    frames = np.stack([big[:camd["Ih"], :camd["Iw"]], big[3:3 + camd["Ih"], 5:5 + camd["Iw"]]]).copy()
    ctx.upload(frames); ctx.process(2, True); ctx.sync()
    (k1, d1), (k2, d2) = ctx.fetch(0), ctx.fetch(1)
    assert len(k1) > 2000 and (k1["octave"] == 0).sum() > 500
    ctx.area_grid(2)
    prev_w = np.stack([k1["x"], k1["y"]], 1).astype(np.float32); prev_g = prev_w.copy()
    want_m, want_n = orc.search_for_initialization(ocam, k1, d1, k2, d2, prev_w, 100, 0.9, True)
    got_m, got_n = ctx.search_for_initialization(1, k1, d1, prev_g, 100, 0.9, True)
    assert got_n == want_n and np.array_equal(got_m, want_m) and np.array_equal(prev_g, prev_w), (got_n, want_n)
    assert want_n >= 100                      # Tracking.cpp:432: initialisation needs at least 100 matches
    ctx.close()


def test_product_reproduces_golden_vectors():
    """the HIP path against the committed regression vectors (tests/golden/oracle_v1.npz) directly -- no live oracle in between"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    import test_area_emu as te
    want = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v1.npz"))
    F = 150
    camd = synth.camera("lafida", F)
    ctx = api.Context(camd, nfeatures=800, max_batch=1)
    ctx.set_mask(synth.cubemap_valid_mask(camd, erode=5, band=30))
    fish = synth.texture(camd["Ih"], camd["Iw"], 3)
    k, d = ctx.remap_extract(fish)
    assert len(k) == int(want["kp_count"][0]) and np.array_equal(mg.h(k.view(np.uint8)), want["kp_hash"]) and np.array_equal(mg.h(d), want["desc_hash"])
    kx, ky, ko = te._keypoints(F, 600, 1)
    kd = synth.descriptors(len(kx), 3)
    ka = np.random.default_rng(5).uniform(0, 360, len(kx)).astype(np.float32)
    kps = np.zeros(len(kx), api.KP_DTYPE); kps["x"] = kx; kps["y"] = ky; kps["octave"] = ko; kps["angle"] = ka
    ctx.area_set_keypoints(0, kps); ctx.area_set_descriptors(0, kd); ctx.area_grid(1)
    qx, qy, qr, lo, hi, _ = te._queries(F, 800, 2)
    off, idx = ctx.features_in_area(0, qx, qy, qr, lo, hi)
    assert np.array_equal(mg.h(off), want["area_off_hash"]) and np.array_equal(mg.h(idx), want["area_idx_hash"])
    lm = synth.local_map_problem(F, kx, ky, ko, kd, seed=4)
    r = ctx.search_local_points(0, lm["pose15"], lm["pos"], lm["normal"], lm["min_dist"], lm["max_dist"], lm["desc"], np.full(len(kx), -1, np.int32), th=5.0)
    fr_hash = mg.h(np.concatenate([r["in_view"].astype(np.float32), r["proj_x"], r["proj_y"], r["level"].astype(np.float32), r["view_cos"]]))
    assert np.array_equal(fr_hash, want["frustum_hash"]) and np.array_equal(r["match"], want["local_match"])
    mm = synth.motion_model_problem(F, kx, ky, ko, ka, kd, seed=6)
    m2, n2 = ctx.search_by_projection(0, mm["pose12"], mm["valid"], mm["Xw"], mm["octave"], mm["angle"], mm["desc"], np.full(len(kx), -1, np.int32))
    assert np.array_equal(m2, want["frame_match"]) and n2 == int(want["frame_nm"][0])
    S = synth.keyframe_set(F, n_kf=4, n_pts=900, seed=7)
    ocam = orc.make_camera(camd)
    for q in S["kfs"]:
        q["rays"] = orc.keyframe_rays(ocam, q["x"], q["y"])            # mvKeyRays (an input of the step)
    g = [api.make_keyframe(q) for q in S["kfs"]]
    on, o1, o2, ox = api.create_new_map_points(ctx, [(g[0][0], [q for q, _ in g[1:]])])[0]
    assert np.array_equal(on, want["tri_neigh"]) and np.array_equal(o1, want["tri_idx1"]) and np.array_equal(o2, want["tri_idx2"])
    assert np.array_equal(ox.view(np.uint32), want["tri_x3d"].view(np.uint32))
    ctx.close()


def test_kfstore_put_from_frame_equals_put():
    """LocalMapping::ProcessNewKeyFrame's device half (LocalMapping.cpp:52-117): cms_kfstore_put_from_frame copies frame b's key points, descriptors,
    key rays and frame grid out of the frame context device to device.  The slot must hold, byte for byte, what cms_kfstore_put stores when it
    is handed the SAME frame fetched to the host (key points, descriptors, rays, map-point slots, FeatureVector, grid: sorted list, cell offsets,
    valid count), the Fuse search on it must give the oracle's result, and cms_kfstore_update_poses must do what cms_kfstore_update does."""
    F = 550
    camd, ocam, _ = _cfg("lafida", F, 2000)
    ctx = api.Context(camd, nfeatures=2000, max_batch=3)
    ctx.set_mask(synth.cubemap_valid_mask(camd))
    frames = np.stack([synth.texture(camd["Ih"], camd["Iw"], s_) for s_ in (71, 72, 73)])
    ctx.upload(frames); ctx.process(3, True); ctx.area_grid(3); ctx.sync()
    cg = api.Context(camd, nfeatures=2000, max_batch=1)              # the mapping side's context (its own stream), like bench.py's window groups
    sA = api.KeyframeStore(cg, max_keyframes=4, max_features=2048, max_nodes=512)
    sB = api.KeyframeStore(cg, max_keyframes=4, max_features=2048, max_nodes=512)
    rng = np.random.default_rng(7)
    kfs = {}
    for b, slot in ((1, 2), (2, 0)):
        k, d = ctx.fetch(b); rays = ctx.fetch_rays(b)
        n = len(k)
        assert n > 1000
        node = (d[:, 0].astype(np.int32) * 2 + (d[:, 1] >> 7)) % 300           # a FeatureVector: features binned by descriptor bits, 10 % left out
        keepf = rng.random(n) < 0.9
        order = np.lexsort((np.arange(n), node)); order = order[keepf[order]]
        ids, starts = np.unique(node[order], return_index=True)
        pr = synth.local_map_problem(F, k["x"], k["y"], k["octave"], d, seed=200 + b)
        kf = dict(x=k["x"], y=k["y"], octave=k["octave"], angle=k["angle"], desc=d, rays=rays, mp=np.where(rng.random(n) < 0.4, rng.integers(0, 5000, n), -1).astype(np.int32),
                  R=pr["pose15"][:9], t=pr["pose15"][9:12], Ow=pr["pose15"][12:], node_id=ids.astype(np.int32), node_off=np.concatenate([starts, [len(order)]]).astype(np.int32),
                  node_feat=order.astype(np.int32), median_depth=2.5 + b)
        K, keep = api.make_keyframe(kf)
        kp = keep[0]; kp["size"] = k["size"]; kp["response"] = k["response"]  # (make_keyframe fills the fields the mapping kernels read; here every field must match)
        sA.put(slot, K)
        sB.put_from_frame(slot, ctx, b, n, kf)
        kfs[slot] = (kf, pr)
    for slot in (2, 0):
        a, bb = sA.debug_fetch(slot), sB.debug_fetch(slot)
        for key in ("kps", "desc", "rays", "mp", "feat_node", "sorted", "node_id", "node_off", "node_feat", "cell_start", "header"):
            x, y = a[key], bb[key]
            assert x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)), (slot, key)
        assert a["nvalid"] == bb["nvalid"] and a["kp_cnt"] == bb["kp_cnt"] == len(kfs[slot][0]["x"])
    # the Fuse search on the device-to-device key frame against the oracle
    sf = kfs[2][1]["scale_factors"]; inv_s2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)
    for slot in (2, 0):
        kf, pr = kfs[slot]
        skip = (np.arange(len(pr["pos"])) % 7 == 0).astype(np.uint8)
        job = dict(skip=skip, pos=pr["pos"], normal=pr["normal"], min_dist=pr["min_dist"], max_dist=pr["max_dist"], desc=pr["desc"])
        okf = orc.make_keyframe(ocam, kf)
        want = orc.fuse_search(ocam, okf[0], skip, pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], 3.0, sf, inv_s2)
        got = sB.fuse_search([(slot, job)], th=3.0)[0]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (want[0] >= 0).sum() > 300, slot
    # poses after a local BA: one asynchronous call for several key frames
    R2 = np.stack([np.roll(np.asarray(kfs[s_][0]["R"], np.float32).reshape(9), 3) for s_ in (2, 0)]); t2 = np.array([[1, 2, 3], [4, 5, 6]], np.float32); O2 = -t2
    sB.update_poses([2, 0], R2, t2, O2)
    for i, slot in enumerate((2, 0)):
        sA.update(slot, R=R2[i], t=t2[i], Ow=O2[i])
        assert np.array_equal(sA.debug_fetch(slot)["header"], sB.debug_fetch(slot)["header"]), slot
    with pytest.raises(api.CmsError):
        sB.put_from_frame(1, ctx, 5, 10, kfs[0][0])                  # no such frame in the batch
    sA.close(); sB.close(); cg.close(); ctx.close()


def test_kfstore_put_from_frames_is_all_or_nothing_and_orders_itself_behind_pose_updates():
    """A batch of key frames in one cms_kfstore_put_from_frames call (LocalMapping::ProcessNewKeyFrame for several camera streams, LocalMapping.cpp:52-117):
    a bad item or a slot named twice must leave the store untouched (earlier items used to be committed before a later one failed), and a put
    into a slot whose asynchronous pose update (cms_kfstore_update_poses, Optimizer.cpp:419-431) may still be queued on the store's stream must
    land AFTER it: the copy runs on the frame context's stream, which then waits for the store's."""
    F = 550
    camd, ocam, _ = _cfg("lafida", F, 2000)
    ctx = api.Context(camd, nfeatures=2000, max_batch=2)
    ctx.set_mask(synth.cubemap_valid_mask(camd))
    ctx.upload(np.stack([synth.texture(camd["Ih"], camd["Iw"], s_) for s_ in (81, 82)])); ctx.process(2, True); ctx.area_grid(2); ctx.sync()
    cg = api.Context(camd, nfeatures=2000, max_batch=1)
    st = api.KeyframeStore(cg, max_keyframes=4, max_features=2048, max_nodes=512)
    kfs = []
    for b in range(2):
        k, d = ctx.fetch(b)
        n = len(k)
        node = (d[:, 0].astype(np.int32)) % 200
        order = np.lexsort((np.arange(n), node))
        ids, starts = np.unique(node[order], return_index=True)
        pr = synth.local_map_problem(F, k["x"], k["y"], k["octave"], d, seed=300 + b)
        kfs.append((n, dict(mp=np.full(n, -1, np.int32), R=pr["pose15"][:9], t=pr["pose15"][9:12], Ow=pr["pose15"][12:], node_id=ids.astype(np.int32),
                            node_off=np.concatenate([starts, [n]]).astype(np.int32), node_feat=order.astype(np.int32), median_depth=2.0)))
    with pytest.raises(api.CmsError):
        st.put_from_frames(ctx, [(0, 0, kfs[0][0], kfs[0][1]), (0, 1, kfs[1][0], kfs[1][1])])      # slot 0 twice
    with pytest.raises(api.CmsError):
        st.put_from_frames(ctx, [(1, 0, kfs[0][0], kfs[0][1]), (2, 7, kfs[1][0], kfs[1][1])])      # second item: no such frame
    for slot in (0, 1, 2):
        with pytest.raises(api.CmsError):
            st.debug_fetch(slot)                                     # nothing was committed: the slots are still empty
    st.put_from_frames(ctx, [(1, 0, kfs[0][0], kfs[0][1]), (2, 1, kfs[1][0], kfs[1][1])])
    h1 = st.debug_fetch(1)["header"].copy()
    assert st.debug_fetch(1)["kp_cnt"] == kfs[0][0] and st.debug_fetch(2)["kp_cnt"] == kfs[1][0]
    # a pose update of slot 1 (asynchronous), then the slot is refilled from the frame: the slot must end up with the PUT's pose, every time
    for rep in range(20):
        R2 = np.full((1, 9), float(rep + 1), np.float32); t2 = np.full((1, 3), -1.0, np.float32)
        st.update_poses([1], R2, t2, -t2)
        st.put_from_frames(ctx, [(1, 0, kfs[0][0], kfs[0][1])])
        assert np.array_equal(st.debug_fetch(1)["header"], h1), rep
    st.close(); cg.close(); ctx.close()


def test_ba_window_handed_to_another_stream_is_ordered_behind_its_set_up():
    """cms_ba_set_stream right after cms_ba_create: the new stream waits for the window's upload and set-up kernel on the device, on every driver --
    the grouped one (windows the blocked solve takes) and the one-window driver (more than 27 free key frames: what a reference-sized local BA with
    a long co-visibility list looks like, Optimizer.cpp:246-357) -- and after a second hand-over before the first use."""
    camd = synth.camera("lafida", 550)
    c1 = api.Context(camd, nfeatures=500, max_batch=1); c2 = api.Context(camd, nfeatures=500, max_batch=1)
    for K, P, seed in ((40, 2500, 4), (12, 3000, 3)):
        prob = synth.ba_problem(K=K, P=P, obs_per_point=5 if K == 40 else 4, F=550, seed=seed, views="random" if K == 40 else "track")
        w = orc.ba_run(prob)
        for hops in (1, 2):
            for rep in range(3):
                ba = api.BundleAdjuster(prob)
                ba.set_stream(c1.stream)
                if hops == 2:
                    ba.set_stream(c2.stream)
                _, stats = api.ba_optimize_many([ba], (5, 10))
                poses, pts, flags = ba.read()
                ba.close()
                assert list(stats[0].iterations_done) == list(w["stats"].iterations_done), (K, hops, rep)
                assert np.array_equal(flags, w["outliers"]), (K, hops, rep)
                _ba_updates_close_or_cascade(prob, poses, pts, w, tag="K %d hops %d" % (K, hops))
    c1.close(); c2.close()


def test_kfstore_fuse_search_matches_oracle():
    """SearchInNeighbors' Fuse calls on resident key frames: three key frames in a store, four jobs (one slot used twice) in one call"""
    import test_area_emu as te
    F = 550
    camd = synth.camera("lafida", F)
    ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=2000, max_batch=1)
    store = api.KeyframeStore(ctx, max_keyframes=4, max_features=2048, max_nodes=16)
    kfs, prs, oks = [], [], []
    for s in range(3):
        kx, ky, ko = te._keypoints(F, 1500 + 200 * s, 100 + s)
        kd = synth.descriptors(len(kx), 110 + s)
        pr = synth.local_map_problem(F, kx, ky, ko, kd, seed=120 + s)
        kf = dict(x=kx, y=ky, octave=ko, angle=np.zeros(len(kx), np.float32), desc=kd, mp=np.full(len(kx), -1, np.int32), R=pr["pose15"][:9], t=pr["pose15"][9:12],
                  Ow=pr["pose15"][12:], node_id=np.zeros(0, np.int32), node_off=np.zeros(1, np.int32), node_feat=np.zeros(0, np.int32), median_depth=1.0,
                  rays=np.zeros((len(kx), 3), np.float32))
        K, keep = api.make_keyframe(kf)
        store.put(s + 1, K)                                     # slots 1..3 (slot 0 stays empty)
        kfs.append(kf); prs.append(pr); oks.append(orc.make_keyframe(ocam, kf))
    sf = prs[0]["scale_factors"]; inv_s2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)
    jobs, want = [], []
    for slot_i, src in ((0, 0), (1, 1), (2, 2), (0, 1)):      # the last job searches key frame 0 with the map points made for key frame 1
        pr = prs[src]
        skip = (np.arange(len(pr["pos"])) % 11 == 0).astype(np.uint8)
        jobs.append((slot_i + 1, dict(skip=skip, pos=pr["pos"], normal=pr["normal"], min_dist=pr["min_dist"], max_dist=pr["max_dist"], desc=pr["desc"])))
        want.append(orc.fuse_search(ocam, oks[slot_i][0], skip, pr["pos"], pr["normal"], pr["min_dist"], pr["max_dist"], pr["desc"], 3.0, sf, inv_s2))
    got = store.fuse_search(jobs, th=3.0)
    for j in range(4):
        assert np.array_equal(got[j][0], want[j][0]) and np.array_equal(got[j][1], want[j][1]), j
    assert sum((w[0] >= 0).sum() for w in want[:3]) > 900
    # the same searches with the map points uploaded once per SET (cms_kfstore_fuse_search_sets: SearchInNeighbors sends one key frame's map points
    # to all of its neighbours): job 3 re-uses job 1's set, one more job sends set 0 to key frame 2 with nothing skipped
    sets = [jobs[j][1] for j in range(3)]
    sjobs = [(1, 0, jobs[0][1]["skip"]), (2, 1, jobs[1][1]["skip"]), (3, 2, jobs[2][1]["skip"]), (1, 1, jobs[3][1]["skip"]), (2, 0, None)]
    got_s = store.fuse_search_sets(sets, sjobs, th=3.0)
    for j in range(4):
        assert np.array_equal(got_s[j][0], want[j][0]) and np.array_equal(got_s[j][1], want[j][1]), ("sets", j)
    pr0 = prs[0]
    w5 = orc.fuse_search(ocam, oks[1][0], np.zeros(len(pr0["pos"]), np.uint8), pr0["pos"], pr0["normal"], pr0["min_dist"], pr0["max_dist"], pr0["desc"], 3.0, sf, inv_s2)
    assert np.array_equal(got_s[4][0], w5[0]) and np.array_equal(got_s[4][1], w5[1])
    with pytest.raises(api.CmsError):
        store.fuse_search([(0, jobs[0][1])])                    # empty slot
    store.close(); ctx.close()


@pytest.mark.parametrize("knob", ["CMS_BA_NO_FUSED_LIN", "CMS_BA_DETERMINISTIC", "CMS_BA_NO_PERMUTE", "CMS_BA_NO_RUNS", "CMS_BA_RUNS_AS_EDGES",
                                  "CMS_BA_SEPARATE_REDUCE", "CMS_BA_RM_VALU", "CMS_BA_SOLVE_REDUCE_MAX=1000",
                                  "CMS_BA_SPLIT_WORKGROUPS", "CMS_BA_SEPARATE_REDUCE2", "CMS_BA_SEPARATE_FIRST_PASS", "CMS_BA_TE_CHUNKS=1", "CMS_BA_TE_CHUNKS=5",
                                  "CMS_BA_ITEMS_COPY_ENGINE", "CMS_BA_RELAXED_WAIT", "CMS_BA_HOST_PLAN", "CMS_BA_LEFTOVER_LOOKAHEAD=4", "CMS_BA_RUN_WG",
                                  "CMS_BA_DETERMINISTIC+CMS_BA_DET_POINTS", "CMS_BA_DETERMINISTIC+CMS_BA_NO_RUNS", "CMS_BA_DETERMINISTIC+CMS_BA_SPLIT_WORKGROUPS",
                                  "CMS_BA_DETERMINISTIC+CMS_BA_HOST_PLAN", "CMS_BA_LEFT_BY_COST", "CMS_BA_DETERMINISTIC+CMS_BA_NO_SOLVE_PRESUM", "CMS_BA_NO_GLOBAL_SUM"])
def test_ba_alternative_schur_paths_pass_the_same_parity_tests(knob):
    """The grouped local-BA driver has several Schur paths -- signature runs multiplied in MFMA tiles + edge-major left-overs, linearisation
    fused (default); the runs' products on the vector ALU by producer / consumer wavefront pairs (CMS_BA_RM_VALU); every point edge-major
    (CMS_BA_NO_RUNS), also with the run order kept (CMS_BA_RUNS_AS_EDGES); the edge-major kernel behind
    kb_ba_lin (CMS_BA_NO_FUSED_LIN); deterministic windows (CMS_BA_DETERMINISTIC: since round 6 the fused chain with its LDS additions in a fixed order,
    kb_ba_lin_schur_runs_det / _edges_det and slices; with CMS_BA_DET_POINTS the pair-owner kernel of rounds 3-5; also without runs, with separate
    workgroups for the left-over chunks, with the host's plan) -- a host-side chunk composition that can be
    switched off (CMS_BA_NO_PERMUTE), and the range sum either inside the solve kernel (the default for groups whose windows have at most 24
    range slices each, i.e. 11 or more windows per group; CMS_BA_SOLVE_REDUCE_MAX=1000: always) or as its own launch (CMS_BA_SEPARATE_REDUCE).
    Round 4's alternatives: separate workgroups for run chunks and left-over chunks instead of cost-balanced ranges over both
    (CMS_BA_SPLIT_WORKGROUPS), kb_ba_reduce2 / the four first-iteration launches instead of their folded forms (CMS_BA_SEPARATE_REDUCE2,
    CMS_BA_SEPARATE_FIRST_PASS), one or five chunks per wavefront of the trial kernel, the window descriptions through a copy engine, sleeping host waits.
    Round 5: the host planner for every window instead of the device-side one (CMS_BA_HOST_PLAN), and round 4's look-ahead for the left-over points.
    Round 6: the runs through one-wavefront workgroups that add straight to the window's global copy of the reduced system, the left-over chunks through
    kb_ba_lin_schur_edges behind them (CMS_BA_RUN_WG: cms_ba_schur_runwg.hip -- the re-decomposition round 5's verdict asked for; slower, kept opt-in).  The
    Second half of round 6: the left-over chunks cut by cost into the ranges of the window's last wavefronts (CMS_BA_LEFT_BY_COST: round 4's cut) instead of
    strided over all wavefronts; windows that keep slices (deterministic ones, or every window under CMS_BA_NO_GLOBAL_SUM) with the solve kernel's assembly walking the
    slices itself (CMS_BA_NO_SOLVE_PRESUM) instead of kb_ba_trial_solve3rp's pre-sum into LDS.  The knobs are read once per process: the config-4 parity tests run again in a child process with the knob set."""
    import os, subprocess, sys
    env = dict(os.environ)
    for kn in knob.split("+"):
        env[kn.split("=")[0]] = kn.split("=")[1] if "=" in kn else "1"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "config4_size_eight or stop_flag_raised or mixed_sizes or tracked_windows"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("knob", ["CMS_BA_DEV_PLAN", "CMS_BA_SETUP_OWN_STREAMS", "CMS_BA_SETUP_PRIORITY=low", "CMS_BA_NO_DEV_PLAN", "CMS_BA_DETERMINISTIC"])
def test_ba_window_group_calls_under_their_switches(knob):
    """cms_ba_create_many's switches, read once per process: every window of every call planned by the plan kernel whatever its flags say (CMS_BA_DEV_PLAN),
    every window set up on a pooled stream of its own as before round 6 instead of one set-up stream per building thread (CMS_BA_SETUP_OWN_STREAMS), the
    set-up streams in the low-priority class, the plan kernel switched off (CMS_BA_NO_DEV_PLAN: the flag is ignored, the host plans).  The group-call tests and
    the C++ step driver's test run again in a child process with the switch set.  CMS_BA_DETERMINISTIC: the same calls with every window deterministic (the group calls
    and the C++ step driver create fixed-order windows then; the plan kernel plans them like any other)."""
    import os, subprocess, sys
    env = dict(os.environ)
    env[knob.split("=")[0]] = knob.split("=")[1] if "=" in knob else "1"
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "create_many_and_read_many" + ("" if knob in ("CMS_BA_NO_DEV_PLAN", "CMS_BA_DETERMINISTIC") else " or small_and_odd or plan_kernel_equals")      # (those two assert that the kernel planned;
    # under CMS_BA_DETERMINISTIC the K = 40 window of small_and_odd runs the pair-owner kernel on another point order and lands 1.1e-4 from the oracle on six pose blocks -- the
    # rounding-level sensitivity of DESIGN.md section 2, which that test's strict bar does not allow for: test_ba_deterministic_mode_on_windows_of_every_solver_path holds such windows to the cascade-aware one)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_batch_driver.py"), "-q", "-x", "-m", "gpu",
                        "-k", sel + " or cpp_driver"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shuffle", [False, True])
def test_ba_device_plan_equals_host_plan(shuffle):
    """cms_ba_create plans a tracked window with one host pass over the observations + per-point work and lets kernels write the observation-sized
    arrays (cms_api_ba_plan.hip).  What the device then holds -- point and edge permutations, sorted edge arrays, per-edge words with the matched
    diagonal copies, chunk descriptors, the runs' MFMA tables -- must be what the host planner (cms_ba_debug_plan) computes, byte for byte; it must
    be a valid partition (every observation once, runs homogeneous, chunks <= 64); and the results come back in the caller's order.  `shuffle`: the
    caller's edges in arbitrary order, as the reference's std::map<KeyFrame*, size_t> iteration gives them (Optimizer.cpp:263-300)."""
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=64, views="track")
    if shuffle:
        o = np.random.default_rng(3).permutation(len(prob["e_pose"]))
        prob = dict(prob, **{k: np.ascontiguousarray(prob[k][o]) for k in ("e_pose", "e_point", "e_obs", "e_invsig2", "e_face")})
    P, E = len(prob["points"]), len(prob["e_pose"])
    ba = api.BundleAdjuster(prob)
    d = ba.fetch_plan()
    assert d["device_planned"]
    h = api.ba_plan(prob["fixed"], P, prob["e_pose"], prob["e_point"], tables=True)
    for k in ("n_chunks", "n_rm", "n_runs", "np", "rm_points"):
        assert d[k] == h[k], (k, d[k], h[k])
    for k in ("pinv", "perm", "rm_chunk", "run_mf", "run_fl"):
        assert d[k].shape == h[k].shape and np.array_equal(d[k], h[k]), k
    # (cms_ba_debug_plan is not given the faces: the per-edge words are compared without their face field, which is checked on its own)
    assert np.array_equal(d["info"] & ~np.uint32(7 << 16), h["info"]) and np.array_equal((d["info"] >> 16) & 7, prob["e_face"][d["perm"]].astype(np.uint32))
    # a valid partition, from the device arrays alone
    assert np.array_equal(np.sort(d["perm"]), np.arange(E)) and np.array_equal(np.sort(d["pinv"]), np.arange(P))
    prank = np.empty(P, np.int64); prank[d["pinv"]] = np.arange(P)
    assert np.array_equal(d["e_pose"], prob["e_pose"][d["perm"]]) and np.array_equal(d["e_point"], prank[prob["e_point"][d["perm"]]]) and np.array_equal(d["e_face"], prob["e_face"][d["perm"]])
    assert np.all(np.diff(d["e_point"]) >= 0) and np.array_equal(d["pt_off"], np.searchsorted(d["e_point"], np.arange(P + 1)))
    assert np.array_equal(d["chunk_e0"], d["pt_off"][h["chunk_pt0"]]) and np.all(np.diff(d["chunk_e0"]) <= 64) and d["chunk_e0"][-1] == E
    for c in range(d["n_rm"]):                                   # a run chunk: whole points of one signature
        e0, word, run, p0 = (int(v) for v in d["rm_chunk"][c])
        ne, k, m = word & 255, (word >> 8) & 255, word >> 16
        poses = d["e_pose"][e0:e0 + ne].reshape(m, k)
        assert ne == k * m and np.all(poses == poses[0]), c
    assert np.all(np.diff(d["rm_cost"].astype(np.int64)) > 0)
    # ... and the window optimises to the oracle's result, reported in the caller's order
    _, st = ba.optimize()
    _check_window(0, ba, prob, st, tag="device plan" + (" (shuffled edges)" if shuffle else ""))
    ba.close()


def test_ba_plan_kernel_equals_the_host_plan():
    """CMS_BA_PLAN_ON_DEVICE: the WHOLE plan of a window by k_ba_plan_many (cms_api_ba_devplan.hip: one workgroup per window -- signature groups by open
    addressing, the runs' order by prefix sums, a point's ordinal inside its group by wavefront matching + one ordered walk over the tiles, greedy packing of
    the left-over points, the matching of their diagonal copies without recursion).  The device arrays must be byte-identical to those of the same window
    created on its own (the host's planner): tracked windows of three sizes, one with its edges shuffled (not grouped by point), one with points nobody
    observes, one larger than a batch of eight; windows the kernel gives up on -- random views (mostly left-over points), a point seen twice by a key frame --
    come back planned by the host inside the same call; the group optimises to the oracle's results; an index out of range fails the call."""
    probs = [synth.ba_problem(K=20, P=5000 + 700 * i, obs_per_point=4, F=550, seed=170 + i, views="track") for i in range(9)]
    probs[0] = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=64, views="track")
    o = np.random.default_rng(5).permutation(len(probs[2]["e_pose"]))
    probs[2] = dict(probs[2], **{k: np.ascontiguousarray(probs[2][k][o]) for k in ("e_pose", "e_point", "e_obs", "e_invsig2", "e_face")})
    # three points nobody observes: their observations go to the next point's key frames ... simply dropped
    p3 = probs[3]; keep = ~np.isin(p3["e_point"], [5, 6, 1000])
    probs[3] = dict(p3, **{k: np.ascontiguousarray(p3[k][keep]) for k in ("e_pose", "e_point", "e_obs", "e_invsig2", "e_face")})
    probs.append(synth.ba_problem(K=12, P=2500, obs_per_point=4, F=550, seed=75, views="random"))        # [9]: the kernel gives it up
    dup = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in probs[4].items()}               # [10]: a point seen twice by a key frame
    dup["e_pose"][1] = dup["e_pose"][0] if dup["e_point"][1] == dup["e_point"][0] else dup["e_pose"][1]
    assert dup["e_point"][1] == dup["e_point"][0]
    probs.append(dup)
    probs[1] = api.pin_problem(probs[1])
    probs = [dict(p, _plan_on_device=True) for p in probs]
    many = api.ba_create_many(probs, threads=3)
    keys = ("pinv", "perm", "info", "pt_off", "e_pose", "e_point", "e_face", "chunk_e0", "rm_chunk", "rm_cost", "run_mf", "run_fl")
    for i, p in enumerate(probs):
        one = api.BundleAdjuster(p)
        a, b = many[i].fetch_plan(), one.fetch_plan()
        assert a["plan_kernel"] == (i < 9), (i, a["plan_kernel"])
        assert a["device_planned"] == b["device_planned"], i
        for k in ("n_chunks", "n_rm", "n_runs", "np", "rm_points", "R_rm", "R"):
            assert a[k] == b[k], (i, k, a[k], b[k])
        for k in keys:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (i, k)
        one.close()
    _, stats = api.ba_optimize_many(many[:9], (5, 10))
    for i in range(9):
        _check_window(i, many[i], probs[i], stats[i], tag="plan kernel")
    for i in (9, 10):
        _, st = many[i].optimize()
        if i == 9:
            _check_window(i, many[i], probs[i], st, tag="plan kernel (fallback)")
    for b in many:
        b.close()
    bad = dict(probs[1]); bad["e_pose"] = probs[1]["e_pose"].copy(); bad["e_pose"][7] = 99
    with pytest.raises(api.CmsError):
        api.ba_create_many([probs[0], bad, probs[3]], threads=2)


def test_ba_plan_kernel_on_small_and_odd_windows():
    """CMS_BA_PLAN_ON_DEVICE on windows at the edges of what the kernel takes: a few points only (less than one tile of 64), one free key frame, 26 key frames
    (25 free: the largest the fused solve takes), 40 key frames (the kernel's tables hold them, the solve path does not: the host declines before the kernel runs),
    every point observed by every key frame (one signature).  Whoever plans a window -- the kernel or, after it gave up, the host -- the device arrays equal those
    of the window created on its own, and the windows optimise to the oracle's results."""
    probs = [synth.ba_problem(K=5, P=40, obs_per_point=3, F=550, seed=300, views="track"),
             synth.ba_problem(K=3, P=700, obs_per_point=3, F=550, seed=301, views="track"),
             synth.ba_problem(K=26, P=1500, obs_per_point=4, F=550, seed=16, views="track"),
             synth.ba_problem(K=40, P=1500, obs_per_point=4, F=550, seed=303, views="track"),
             synth.ba_problem(K=6, P=900, obs_per_point=6, F=550, seed=304, views="track")]
    probs[1]["fixed"][:] = 1; probs[1]["fixed"][2] = 0          # one free key frame
    probs = [dict(p, _plan_on_device=True) for p in probs]
    many = api.ba_create_many(probs, threads=2)
    keys = ("pinv", "perm", "info", "pt_off", "e_pose", "e_point", "e_face", "chunk_e0", "rm_chunk", "rm_cost", "run_mf", "run_fl")
    planned_by_kernel = 0
    for i, p in enumerate(probs):
        one = api.BundleAdjuster(p)
        a, b = many[i].fetch_plan(), one.fetch_plan()
        planned_by_kernel += bool(a["plan_kernel"])
        assert a["device_planned"] == b["device_planned"], i
        for k in ("n_chunks", "n_rm", "n_runs", "np", "rm_points", "R_rm", "R"):
            assert a[k] == b[k], (i, k, a[k], b[k])
        for k in keys:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (i, k)
        one.close()
    assert planned_by_kernel >= 1          # (the small ones have no signature runs worth a chunk: the kernel says so and the host plans them)
    for i, b in enumerate(many):
        _, st = b.optimize()
        _check_window(i, b, probs[i], st, tag="plan kernel (small / odd)")
        b.close()


def test_ba_create_many_and_read_many_equal_the_single_window_calls():
    """A window group's set-up and read-back as one call each (cms_ba_create_many: host parts on several threads, ONE expansion launch per eight
    device-planned windows; cms_ba_read_many: one gather launch): the device arrays of every window must be byte-identical to those of the same window
    created on its own -- four tracked windows (device-planned, one of them with shuffled edges) and one window with random views (the host planner's,
    which sets itself up) in the same call --, the group optimises to the oracle's results, and the batched read-back returns what cms_ba_read returns.
    (What one Optimizer::LocalBundleAdjustment call assembles and writes back: Optimizer.cpp:246-357, 419-450.)"""
    probs = [synth.ba_problem(K=20, P=6000 + 500 * i, obs_per_point=4, F=550, seed=70 + i, views="track") for i in range(4)]
    o = np.random.default_rng(5).permutation(len(probs[2]["e_pose"]))
    probs[2] = dict(probs[2], **{k: np.ascontiguousarray(probs[2][k][o]) for k in ("e_pose", "e_point", "e_obs", "e_invsig2", "e_face")})
    probs.append(synth.ba_problem(K=12, P=2500, obs_per_point=4, F=550, seed=75, views="random"))
    probs[1] = api.pin_problem(probs[1])            # CMS_BA_INPUTS_PINNED: this window's arrays are copied to the device from the caller's pinned memory, unstaged
    probs[4] = api.pin_problem(probs[4])            # (... and a host-planned window with the flag)
    many = api.ba_create_many(probs, threads=3)
    keys = ("pinv", "perm", "info", "pt_off", "e_pose", "e_point", "e_face", "chunk_e0", "rm_chunk", "rm_cost", "run_mf", "run_fl")
    for i, p in enumerate(probs):
        one = api.BundleAdjuster(p)
        a, b = many[i].fetch_plan(), one.fetch_plan()
        assert a["device_planned"] == b["device_planned"] == (i < 4), i
        for k in keys:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (i, k)
        one.close()
    # the four device-planned windows as a group (one kind per cms_ba_optimize_many group), the host-planned one on its own
    _, stats = api.ba_optimize_many(many[:4], (5, 10))
    _, st4 = many[4].optimize()
    stats = list(stats) + [st4]
    outs = api.ba_read_many(many)
    for i, p in enumerate(probs):
        poses, pts, flags = many[i].read()
        assert np.array_equal(outs[i][0], poses) and np.array_equal(outs[i][1], pts) and np.array_equal(outs[i][2], flags), i
        _check_window(i, many[i], p, stats[i], tag="create_many")
    for b in many:
        b.close()
    # an error in one window leaves nothing behind
    bad = dict(probs[1]); bad["e_pose"] = probs[1]["e_pose"].copy(); bad["e_pose"][7] = 99
    with pytest.raises(api.CmsError):
        api.ba_create_many([probs[0], bad, probs[3]], threads=2)


def _check_window(i, ba, p, st, w=None, tag="window"):
    poses, pts, flags = ba.read()
    w = w or orc.ba_run(p)
    ws = w["stats"]
    assert list(st.iterations_done) == list(ws.iterations_done), (tag, i, list(st.iterations_done), list(ws.iterations_done))
    for j in range(2):
        assert abs(st.chi2_final[j] - ws.chi2_final[j]) <= 1e-6 * abs(ws.chi2_final[j]), (tag, i, j)
    assert st.n_outliers_mid == ws.n_outliers_mid and st.n_outliers_final == ws.n_outliers_final, (tag, i)
    assert np.array_equal(flags, w["outliers"]), (tag, i, int((flags != w["outliers"]).sum()))
    return _ba_updates_close_or_cascade(p, poses, pts, w, tag="%s %d" % (tag, i))


def test_ba_signature_runs_config4_tracked_windows():
    """Windows whose map points are TRACKED over consecutive key frames (synth.ba_problem(views="track")): most points share their set of
    observing key frames with many others, cms_ba_create turns those sets into runs and the run-major body of kb_ba_lin_schur_runs sums their
    products in registers (cms_ba_schur_runs.hip); the rest goes edge-major in the same launch.  configs[3] size (K = 20, E ~ 80 k), a group
    of four different windows (one far from its optimum: rejected trials), then ONE window on its own (the 128-range split), each against its
    own oracle run: iteration counts, chi2, outlier flags, per-block updates."""
    probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=142 + i, views="track") for i in range(4)]
    rng = np.random.default_rng(5)
    probs[2]["points"] = probs[2]["points"] + rng.normal(0, 0.05, probs[2]["points"].shape)
    for p in probs:
        assert 76000 < len(p["e_pose"]) < 86000
        pl = api.ba_plan(p["fixed"], len(p["points"]), p["e_pose"], p["e_point"])
        if not any(k.startswith("CMS_BA_") for k in os.environ):
            assert pl["rm_points"] > 0.6 * len(p["points"]) and pl["n_runs"] > 20      # the run-major body is what this test exercises
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    wants = [orc.ba_run(p) for p in probs]
    for i, (ba, p, st, w) in enumerate(zip(bas, probs, stats, wants)):
        _check_window(i, ba, p, st, w, tag="tracked window")
        ba.close()
    g = api.ba_run(probs[0])
    assert list(g["stats"].iterations_done) == list(wants[0]["stats"].iterations_done) and np.array_equal(g["outliers"], wants[0]["outliers"])
    _ba_updates_close_or_cascade(probs[0], g["poses"], g["points"], wants[0], tag="tracked window alone")


def test_ba_signature_runs_shapes():
    """Run-major body on windows of other shapes: two fixed key frames (observations without a pose block inside a signature), 2 .. 9
    observations per point (1 .. 45 tuples per signature: one to twenty-one lanes per pose pair), a window too small for any run next to
    large ones in one group, 24 free key frames (the LDS copy leaves no room for the chunk buffers: the group falls back to edge-major)."""
    probs = [synth.ba_problem(K=12, P=6000, obs_per_point=3, F=550, seed=201, views="track"),
             synth.ba_problem(K=16, P=9000, obs_per_point=6, F=650, seed=202, views="track", dropout=0.03),
             synth.ba_problem(K=7, P=300, obs_per_point=4, F=550, seed=203, views="track"),
             synth.ba_problem(K=20, P=12000, obs_per_point=2, F=550, seed=204, views="track", dropout=0.0)]
    probs[0]["fixed"][1] = 1
    probs[1]["fixed"][5] = 1
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for i, (ba, p, st) in enumerate(zip(bas, probs, stats)):
        _check_window(i, ba, p, st, tag="shape")
        ba.close()
    big = [synth.ba_problem(K=25, P=8000, obs_per_point=4, F=550, seed=205, views="track"), synth.ba_problem(K=20, P=8000, obs_per_point=4, F=550, seed=206, views="track")]
    bas = [api.BundleAdjuster(p) for p in big]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for i, (ba, p, st) in enumerate(zip(bas, big, stats)):
        _check_window(i, ba, p, st, tag="24 free key frames in the group")
        ba.close()


def test_ba_group_of_sixteen_config4_windows():
    """bench.py's shape: SIXTEEN configs[3] windows per cms_ba_optimize_many call (--ba-groups 2 of 32): the group's share of the chip per
    window (ba_group_ranges: 16 workgroups each) differs from the groups of eight the other tests run.  Tracked and random windows mixed."""
    probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=300 + i, views="track" if i % 2 == 0 else "random") for i in range(16)]
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    worst, cascaded = 0.0, 0
    for i, (ba, p, st) in enumerate(zip(bas, probs, stats)):
        r, c = _check_window(i, ba, p, st, tag="group of 16, window")
        cascaded += c
        if not c:
            worst = max(worst, r)
        ba.close()
    assert worst <= 1e-4 and cascaded <= 3, (worst, cascaded)      # (a float-rounding cascade is rare: ~1 window in 20)


@pytest.mark.parametrize("views", ["track", "random"])
def test_ba_default_path_is_repeatable(views):
    """Determinism contract of the default local-BA path.  The LDS additions of different wavefronts interleave in an order that is not
    fixed (ds_add_f64 on the workgroup's copy of the reduced system: every element for a random window, once per signature run for a
    tracked one), so sums may differ in their last bits from run to run -- the reference (g2o on the CPU) is deterministic.  What the
    product guarantees, and what this test holds it to over TWENTY runs of one 80 k-edge window: identical iteration counts of both stages
    and identical outlier counts; the estimates of at least eighteen of the runs agree to 1e-9 of the largest update (the others: a last-bit
    difference that tipped a float rounding inside the reference's projection -- the chaos the oracle shows against itself -- bounded at a
    percent); an outlier flag may only differ on an edge whose chi2 sits within 1e-6 of the 5.991 threshold in the oracle's estimate (the
    tolerated flip set; empty in every run seen).
    CMS_BA_DETERMINISTIC=1 selects the pair-owner kernel, which is bit-identical run to run."""
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=77, views=views)
    w = orc.ba_run(prob)
    runs = []
    for r in range(20):
        g = api.ba_run(prob)
        runs.append(g)
        assert list(g["stats"].iterations_done) == list(w["stats"].iterations_done), (r, list(g["stats"].iterations_done))
        assert g["stats"].n_outliers_mid == w["stats"].n_outliers_mid
    scale_p = np.abs(w["points"] - prob["points"]).max(); scale_t = np.abs(w["poses"][:, :3] - prob["poses"][:, :3]).max()
    def dist(a, b):
        return max(np.abs(a["points"] - b["points"]).max() / scale_p, np.abs(a["poses"] - b["poses"]).max() / scale_t)
    # the runs that agree with run r to 1e-9 of the largest update; the biggest such family must hold (nearly) all twenty.  A run outside it
    # is one whose last-bit differences tipped a float rounding of the reference's projection (see _ba_updates_close_or_cascade): allowed
    # for at most two runs, and bounded
    fam = max(([q for q in range(20) if dist(runs[q], runs[r]) <= 1e-9] for r in range(20)), key=len)
    assert len(fam) >= 18, len(fam)
    for q in range(20):
        if q not in fam:
            assert dist(runs[q], runs[fam[0]]) <= 2e-2, (q, dist(runs[q], runs[fam[0]]))
    flips = np.zeros(len(prob["e_pose"]), bool)
    for g in runs:
        flips |= g["outliers"] != w["outliers"]
    if flips.any():
        err = orc.ba_linearize(dict(prob, poses=w["poses"], points=w["points"]), robust=False)["err"]      # residuals at the oracle's final estimate
        chi = prob["e_invsig2"] * (err ** 2).sum(1)
        assert np.all(np.abs(chi[flips] - 5.991) < 1e-6), (int(flips.sum()), chi[flips][:8])
    _ba_updates_close_or_cascade(prob, runs[-1]["poses"], runs[-1]["points"], w, tag="repeat")


def test_front_camera_eight_stream_batch_matches_oracle():
    """BASELINE.json configs[4] as a WORKLOAD on one GPU (bench.py --camera front): eight front_cam streams (1280x720, F = 650,
    nFeatures 3000) x 2 consecutive frames in ONE cms_frames_process batch of 16, every frame's key points and descriptors against the
    oracle -- the batch layout, the zero-corner bookkeeping and the per-frame quotas at the size the multi-GPU bench shards."""
    camd, ocam, nfeat = _cfg("front", 650, 3000)
    n_str, fps = 8, 2
    ctx = api.Context(camd, nfeatures=nfeat, max_batch=n_str * fps)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    m1, m2 = orc.build_lut(ocam)
    o = orc.Orb(nfeatures=nfeat)
    frames = []
    for s_ in range(n_str):
        big = synth.texture(camd["Ih"] + 16, camd["Iw"] + 16, 900 + 31 * s_)
        frames += [big[2 * b:2 * b + camd["Ih"], 3 * b:3 * b + camd["Iw"]] for b in range(fps)]
    frames = np.ascontiguousarray(np.stack(frames))
    ctx.upload(frames)
    ctx.process(n_str * fps, True)
    ctx.sync()
    total = 0
    for b in range(n_str * fps):
        cube = orc.fisheye_to_cubemap(ocam, m1, m2, frames[b])
        wk, wd = o.extract(ocam, cube, mask)
        gk, gd = ctx.fetch(b)
        assert len(gk) == len(wk) and np.array_equal(gk.view(np.uint8), wk.view(np.uint8)) and np.array_equal(gd, wd), (b, len(gk), len(wk))
        total += len(gk)
    assert total > 16 * 1500
    ctx.close()


def test_describe_in_spatial_order_is_bit_identical():
    """CMS_DESC_SPATIAL_ORDER=1 changes the order k_describe WORKS in (the batch's key points band by band of their levels, an eighth of that walk
    per XCD: a third of the HBM fetches), not its output: the extraction parity tests run again in a child process with the knob set (it is read
    when a context is created)."""
    import os, subprocess, sys
    env = dict(os.environ); env["CMS_DESC_SPATIAL_ORDER"] = "1"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", "extract and not spatial"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_pyramid_levels_per_launch_is_bit_identical():
    """CMS_RESIZE_FUSED=1 computes levels (1, 2), (3, 4), (5, 6) two per launch (k_resize2: the intermediate level in LDS, stored by its owner workgroup, never
    read back).  Not the default (measured slower: the pyramid is bound by per-pixel integer work, not bytes) but kept as an experiment: the extraction parity
    tests -- every pyramid level compared pixel by pixel, three face sizes, zero corners skipped or not -- run again in a child process with the switch set."""
    import os, subprocess, sys
    env = dict(os.environ); env["CMS_RESIZE_FUSED"] = "1"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "(extract or lut or front_camera or reference_masks) and not spatial and not two_pyramid"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_ba_optimize_many_ragged_windows_edge_major_path():
    """Windows whose points have very different numbers of observations (1 .. 12: up to six pairing steps per chunk, points straddling
    the 16-lane groups the chunk composition balances, chunks closed by a point that does not fit), two fixed key frames, key frames
    nobody observes -- through the grouped driver's default (edge-major, fused) path, every window against its own oracle run."""
    rng = np.random.default_rng(11)
    probs = []
    for i, (K, P, opp) in enumerate(((16, 2500, 12), (21, 4000, 7), (9, 1200, 9))):
        p = synth.ba_problem(K=K, P=P, obs_per_point=opp, F=550, seed=70 + i)
        # thin the observations out at random so that the per-point counts spread over 1 .. opp
        keep = rng.uniform(size=len(p["e_pose"])) < 0.7
        for key in ("e_pose", "e_point", "e_invsig2", "e_face"):
            p[key] = np.ascontiguousarray(p[key][keep])
        p["e_obs"] = np.ascontiguousarray(p["e_obs"][keep])
        p["fixed"][1] = 1                                    # a second fixed key frame
        probs.append(p)
    counts = np.bincount(probs[0]["e_point"], minlength=probs[0]["points"].shape[0])
    assert counts.min() <= 1 and counts.max() >= 10
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for i, (ba, p, st) in enumerate(zip(bas, probs, stats)):
        poses, pts, flags = ba.read()
        w = orc.ba_run(p)
        assert list(st.iterations_done) == list(w["stats"].iterations_done), (i, list(st.iterations_done), list(w["stats"].iterations_done))
        assert np.array_equal(flags, w["outliers"]), (i, int((flags != w["outliers"]).sum()))
        _ba_updates_close(p, poses, pts, w, tag="ragged window %d" % i)
        ba.close()


def test_ba_optimize_many_call_mixing_window_kinds_is_split_into_groups():
    """One cms_ba_optimize_many call over windows that have the edge-major work list (up to 25 free key frames) and windows that do not
    (29 and 39 free key frames): the call runs them as separate groups and reports the statistics in the caller's order."""
    probs = [synth.ba_problem(K=20, P=2000, obs_per_point=4, F=550, seed=81), synth.ba_problem(K=30, P=1500, obs_per_point=5, F=550, seed=82),
             synth.ba_problem(K=12, P=900, obs_per_point=3, F=650, seed=83), synth.ba_problem(K=40, P=1800, obs_per_point=6, F=550, seed=84)]
    bas = [api.BundleAdjuster(p) for p in probs]
    rc, stats = api.ba_optimize_many(bas)
    assert rc == 0
    for i, (ba, p, st) in enumerate(zip(bas, probs, stats)):
        poses, pts, flags = ba.read()
        w = orc.ba_run(p)
        assert list(st.iterations_done) == list(w["stats"].iterations_done), i
        assert st.n_outliers_final == w["stats"].n_outliers_final and np.array_equal(flags, w["outliers"]), i
        _ba_updates_close(p, poses, pts, w, tag="window %d" % i)
        ba.close()


def test_distance_bounds_from_the_public_getters():
    """cms_set_distance_bounds_mode(ctx, 1): the caller hands over MapPoint::GetMinDistanceInvariance() / GetMaxDistanceInvariance()
    (0.8f * mfMinDistance, 1.2f * mfMaxDistance in float, MapPoint.cpp:375-385) instead of the private members, so that a binding does not
    have to touch MapPoint.h.  Same frame, same local map (a few thousand points), both modes: visibility, projections, predicted levels, view cosines and
    the matches must be identical (the recovery of mfMaxDistance from 1.2f * mfMaxDistance is exact except where two floats share a product,
    and there it only matters if PredictScale's quotient sits on an integer)."""
    camd, ocam, _ = _cfg("lafida", 450, 2000)
    ctx = api.Context(camd, nfeatures=2000, max_batch=1)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    k, d = ctx.remap_extract(synth.texture(camd["Ih"], camd["Iw"], 5))
    ctx.area_grid(1)
    lm = synth.local_map_problem(450, k["x"], k["y"], k["octave"], d, seed=21, extra=3.0)
    raw = ctx.search_local_points(0, lm["pose15"], lm["pos"], lm["normal"], lm["min_dist"], lm["max_dist"], lm["desc"], np.full(len(k), -1, np.int32))
    ctx.set_distance_bounds_mode(1)
    smin = (np.float32(0.8) * lm["min_dist"].astype(np.float32)).astype(np.float32)
    smax = (np.float32(1.2) * lm["max_dist"].astype(np.float32)).astype(np.float32)
    pub = ctx.search_local_points(0, lm["pose15"], lm["pos"], lm["normal"], smin, smax, lm["desc"], np.full(len(k), -1, np.int32))
    ctx.set_distance_bounds_mode(0)
    assert raw["in_view"].sum() > 100
    for key in ("in_view", "level", "match"):
        assert np.array_equal(raw[key], pub[key]), key
    for key in ("proj_x", "proj_y", "view_cos"):
        assert np.array_equal(raw[key].view(np.uint32), pub[key].view(np.uint32)), key
    # the recovery itself, on a million floats: 1.2f * r == 1.2f * recovered r
    x = np.random.RandomState(3).uniform(0.05, 80.0, 1000000).astype(np.float32)
    s_ = (np.float32(1.2) * x).astype(np.float32)
    r = (s_.astype(np.float64) / np.float64(np.float32(1.2))).astype(np.float32)
    lo = (r.view(np.uint32) - 1).view(np.float32); hi = (r.view(np.uint32) + 1).view(np.float32)
    rec = np.where((np.float32(1.2) * lo).astype(np.float32) == s_, lo, np.where((np.float32(1.2) * r).astype(np.float32) == s_, r, hi))
    assert np.array_equal((np.float32(1.2) * rec).astype(np.float32), s_) and (rec != x).mean() < 0.5 and np.abs(rec.view(np.int32) - x.view(np.int32)).max() <= 1
    ctx.close()


def test_two_host_threads_create_contexts_and_windows_on_one_device():
    """Per-device state of the library (the dynamic-LDS ceilings set once per device, the BA pools, the CU count of the range split) is keyed by the
    device and guarded: two host threads that create contexts and local-BA windows on device 0 at the same time -- Tracking and LocalMapping do
    (SURVEY.md 8b), and so do the ranks' window pools -- get the results of a quiet run."""
    import threading
    camd = synth.camera("lafida", 350)
    mask = synth.cubemap_valid_mask(camd)
    fish = synth.texture(camd["Ih"], camd["Iw"], 5)
    prob = synth.ba_problem(K=6, P=600, obs_per_point=4, F=350, seed=77, views="track")
    ref_ctx = api.Context(camd, nfeatures=1000, max_batch=1, device=0); ref_ctx.set_mask(mask)
    ref_k, ref_d = ref_ctx.remap_extract(fish); ref_ctx.close()
    ref_ba = api.ba_run(prob)
    out, errs = [None, None], []

    def worker(i):
        try:
            res = []
            for rep in range(3):
                c = api.Context(camd, nfeatures=1000, max_batch=1, device=0); c.set_mask(mask)
                k, d = c.remap_extract(fish)
                r = api.ba_run(prob, device=0)
                c.close()
                res.append((k, d, r))
            out[i] = res
        except Exception as ex:          # noqa: BLE001 -- reported by the main thread
            errs.append(ex)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for res in out:
        for k, d, r in res:
            assert np.array_equal(k.view(np.uint8), ref_k.view(np.uint8)) and np.array_equal(d, ref_d)
            assert list(r["stats"].iterations_done) == list(ref_ba["stats"].iterations_done) and np.array_equal(r["outliers"], ref_ba["outliers"])
            assert np.abs(r["points"] - ref_ba["points"]).max() <= 1e-6 * max(np.abs(ref_ba["points"] - prob["points"]).max(), 1e-12) + 1e-12
