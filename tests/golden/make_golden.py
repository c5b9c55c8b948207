"""Regenerates tests/golden/oracle_v1.npz: outputs of the CPU oracle (oracle/liborc.so) on small seeded inputs, kept as regression
vectors.  The reference itself cannot be run in this image, so these pin the ORACLE across rounds -- a change in
any stage of the restatement shows up in tests/test_golden.py -- not the oracle against the reference.
Usage:  python tests/golden/make_golden.py"""
import hashlib, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc
from cubemapslam_amd import synth
import test_area_emu as te


def h(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:16], np.uint8).copy()


def build():
    out = {}
    F = 150
    camd = synth.camera("lafida", F)
    cam = orc.make_camera(camd)
    m1, m2 = orc.build_lut(cam)
    fish = synth.texture(camd["Ih"], camd["Iw"], 3)
    cube = orc.fisheye_to_cubemap(cam, m1, m2, fish)
    mask = synth.cubemap_valid_mask(camd, erode=5, band=30)
    k, d = orc.Orb(nfeatures=800).extract(cam, cube, mask)
    out["cube_hash"] = h(cube); out["kp_count"] = np.array([len(k)]); out["kp_hash"] = h(k.view(np.uint8)); out["desc_hash"] = h(d)
    out["kp_head"] = k[:16].view(np.uint8).copy(); out["desc_head"] = d[:16].copy()
    kx, ky, ko = te._keypoints(F, 600, 1)
    qx, qy, qr, lo, hi, _ = te._queries(F, 800, 2)
    off, idx = orc.features_in_area(cam, kx, ky, ko, qx, qy, qr, lo, hi)
    out["area_off_hash"] = h(off); out["area_idx_hash"] = h(idx); out["area_total"] = np.array([len(idx)])
    kd = synth.descriptors(len(kx), 3)
    lm = synth.local_map_problem(F, kx, ky, ko, kd, seed=4)
    fr = orc.is_in_frustum(cam, lm["pose15"], lm["pos"], lm["normal"], lm["min_dist"], lm["max_dist"])
    kp_mp = np.full(len(kx), -1, np.int32)
    match, nm = orc.search_local_points(cam, kx, ky, ko, kd, lm["scale_factors"], fr, lm["desc"], kp_mp, th=5.0)
    out["frustum_hash"] = h(np.concatenate([fr["in_view"].astype(np.float32), fr["proj_x"], fr["proj_y"], fr["level"].astype(np.float32), fr["view_cos"]]))
    out["local_match"] = match; out["local_nm"] = np.array([nm])
    ka = np.random.default_rng(5).uniform(0, 360, len(kx)).astype(np.float32)
    mm = synth.motion_model_problem(F, kx, ky, ko, ka, kd, seed=6)
    kp2 = np.full(len(kx), -1, np.int32)
    m2_, n2 = orc.search_by_projection_frames(cam, mm["pose12"][:9], mm["pose12"][9:], kx, ky, ko, ka, kd, mm["scale_factors"], mm["valid"], mm["Xw"], mm["octave"],
                                              mm["angle"], mm["desc"], kp2, th=15.0, check_ori=True)
    out["frame_match"] = m2_; out["frame_nm"] = np.array([n2])
    S = synth.keyframe_set(F, n_kf=4, n_pts=900, seed=7)
    oks = [orc.make_keyframe(cam, q) for q in S["kfs"]]
    on, o1, o2, ox = orc.create_new_map_points(cam, oks[0][0], [q for q, _ in oks[1:]], S["scale_factors"], S["level_sigma2"], S["kfs"][0]["mp"].copy())
    out["tri_neigh"] = on; out["tri_idx1"] = o1; out["tri_idx2"] = o2; out["tri_x3d"] = ox
    prob = synth.ba_problem(K=6, P=300, obs_per_point=4, F=650, seed=9)
    w = orc.ba_run(prob)
    st = w["stats"]
    out["ba_iterations"] = np.array(list(st.iterations_done)); out["ba_chi2_final"] = np.array(list(st.chi2_final))
    out["ba_outliers"] = np.array([st.n_outliers_mid, st.n_outliers_final]); out["ba_points_head"] = w["points"][:8].copy()
    pp = synth.pose_problem(N=200, F=550, seed=2)
    n_in, pose, outl, _ = orc.pose_optimize(pp)
    out["pose_inliers"] = np.array([n_in]); out["pose_pose"] = pose; out["pose_outliers"] = outl
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **build())
    print("wrote", os.path.join(HERE, "oracle_v1.npz"))
