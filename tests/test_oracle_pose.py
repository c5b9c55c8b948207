"""Oracle checks for Optimizer::PoseOptimization: the restatement in oracle/orc_ba.cpp against properties
that do not depend on its own code -- ground-truth recovery, outlier recovery, the reference's early exits."""
import numpy as np
import orc
from cubemapslam_amd import synth


def _rot_angle_deg(qa, qb):
    d = abs(float(np.dot(qa / np.linalg.norm(qa), qb / np.linalg.norm(qb))))
    return np.rad2deg(2 * np.arccos(min(1.0, d)))


def test_pose_recovers_ground_truth_without_noise():
    pr = synth.pose_problem(N=400, seed=3, outlier_frac=0.0)
    # replace the noisy observations by exact projections: the optimum is the ground truth and every edge is an inlier
    R = np.zeros((3, 3)); q = pr["pose_gt"][3:]
    x, y, z, w = q
    R[:] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    Xc = pr["Xw"] @ R.T + pr["pose_gt"][:3]
    face, up, vp = synth.rays_to_cubemap(550, Xc)
    keep = face == pr["face"]
    F = 550
    pr["obs"][keep, 0] = up[keep] - np.floor(up[keep] / F) * F
    pr["obs"][keep, 1] = vp[keep] - np.floor(vp[keep] / F) * F
    for k in ("Xw", "obs", "invsig2", "face"):
        pr[k] = np.ascontiguousarray(pr[k][keep])
    n_in, pose, out, st = orc.pose_optimize(pr)
    assert n_in == keep.sum() and out.sum() == 0
    assert np.linalg.norm(pose[:3] - pr["pose_gt"][:3]) < 1e-4
    assert _rot_angle_deg(pose[3:], pr["pose_gt"][3:]) < 1e-3
    assert st.rounds == 4 and all(1 <= st.iterations_done[i] <= 10 for i in range(4))


def test_pose_flags_gross_outliers_and_improves_the_estimate():
    pr = synth.pose_problem(N=800, seed=11, outlier_frac=0.15)
    n_in, pose, out, st = orc.pose_optimize(pr)
    gross = pr["gross"]
    assert n_in == len(out) - out.sum() == len(out) - st.n_bad
    # nearly every gross mismatch is rejected, nearly every clean observation kept
    assert out[gross].mean() > 0.9 and out[~gross].mean() < 0.12
    e0 = np.linalg.norm(pr["pose0"][:3] - pr["pose_gt"][:3]); e1 = np.linalg.norm(pose[:3] - pr["pose_gt"][:3])
    assert e1 < 0.2 * e0 and _rot_angle_deg(pose[3:], pr["pose_gt"][3:]) < 0.1


def test_pose_early_exits():
    pr = synth.pose_problem(N=50, seed=5)
    # fewer than 3 correspondences: returns 0 and leaves the pose alone (Optimizer.cpp:131-132)
    n_in, pose, out, st = orc.pose_optimize(pr, n=2)
    assert n_in == 0 and np.array_equal(pose, pr["pose0"]) and st.rounds == 0
    # fewer than 10 edges: a single round (Optimizer.cpp:175-176)
    n_in, pose, out, st = orc.pose_optimize(pr, n=8)
    assert st.rounds == 1 and st.iterations_done[0] >= 1 and st.iterations_done[1] == 0
    assert n_in == 8 - out.sum()


def test_pose_optimisation_against_a_numpy_restatement():
    """oracle/orc_ba.cpp's PoseOptimization against tests/npref_pose.py, the same procedure written from the reference's text with numpy's dense
    solver: inlier counts, outlier flags and the iteration counts of the four rounds identical, poses equal to 1e-9 (one 6x6 system: the two
    programs differ in the order of a few hundred additions)."""
    import npref_pose
    for seed, N in ((2, 300), (5, 120), (9, 40), (14, 600), (23, 8)):
        pr = synth.pose_problem(N=N, seed=seed, outlier_frac=0.12)
        n_w, pose_w, out_w, its_w = npref_pose.pose_optimize(pr)
        n_g, pose_g, out_g, st = orc.pose_optimize(pr)
        assert n_g == n_w and np.array_equal(out_g, out_w), (seed, n_g, n_w, int((out_g != out_w).sum()))
        assert [st.iterations_done[i] for i in range(st.rounds)] == its_w, (seed, [st.iterations_done[i] for i in range(4)], its_w)
        assert np.abs(pose_g - pose_w).max() <= 1e-9, (seed, np.abs(pose_g - pose_w).max())
