"""The device-side planner of cms_ba_create (cubemapslam_amd/csrc/cms_api_ba_plan.hip), on the host: cms_ba_debug_plan_fast runs its host half
(ONE pass over the observations, then per-point work) and then the expansion kernels' bodies -- the same __host__ __device__ functions the
kernels call -- over host arrays.  For every window both planners take, the result must be byte-identical to cms_ba_debug_plan's (the host
planner, whose index arithmetic tests/test_ba_runs_cpu.py replays against the plain Schur sum); windows it does not take must say so.
No GPU, no oracle: list construction.  (The graph is what Optimizer::LocalBundleAdjustment hands to g2o, Optimizer.cpp:192-363.)"""
import numpy as np
import pytest

from cubemapslam_amd import api, synth

KEYS = ("pinv", "perm", "info", "chunk_pt0", "rm_chunk", "run_mf", "run_fl")


def _both(prob):
    P = len(prob["points"])
    a = api.ba_plan(prob["fixed"], P, prob["e_pose"], prob["e_point"], tables=True)
    f = api.ba_plan_fast(prob["fixed"], P, prob["e_pose"], prob["e_point"])
    return a, f


def _assert_same(a, f, tag=""):
    assert f["usable"], tag
    for k in ("n_chunks", "n_rm", "n_runs", "np", "rm_points", "R_rm", "R"):
        assert a[k] == f[k], (tag, k, a[k], f[k])
    for k in KEYS:
        assert a[k].shape == f[k].shape and np.array_equal(a[k], f[k]), (tag, k)


@pytest.mark.parametrize("K,P,opp,seed", [(20, 22150, 4, 42), (20, 22150, 4, 43), (12, 3000, 4, 3), (8, 2400, 4, 19), (20, 9000, 6, 7), (20, 6000, 2, 8), (16, 8000, 5, 9)])
def test_device_planner_equals_host_planner_on_tracked_windows(K, P, opp, seed):
    prob = synth.ba_problem(K=K, P=P, obs_per_point=opp, F=550, seed=seed, views="track")
    a, f = _both(prob)
    assert a["n_runs"] > 0
    if not f["usable"]:
        assert 3 * (P - a["rm_points"]) > P      # only windows whose left-over points are more than a third are handed back
        return
    _assert_same(a, f, (K, P, opp, seed))


def test_callers_edge_order_does_not_matter():
    """The reference adds a map point's observations in the order of a std::map keyed by KeyFrame POINTER (Optimizer.cpp:263-300): arbitrary.
    Edges shuffled inside their points (still grouped by point), and edges shuffled completely (the planner then groups them itself): same internal
    structure, the permutation maps to the same (point, key frame) pairs."""
    prob = synth.ba_problem(K=20, P=6000, obs_per_point=4, F=550, seed=12, views="track")
    a0, f0 = _both(prob)
    _assert_same(a0, f0, "sorted")
    rng = np.random.default_rng(5)
    E = len(prob["e_pose"])
    # (1) shuffled inside the points
    o1 = np.lexsort((rng.uniform(size=E), prob["e_point"]))
    # (2) shuffled completely
    o2 = rng.permutation(E)
    for tag, o in (("inside points", o1), ("completely", o2)):
        q = dict(prob, e_pose=np.ascontiguousarray(prob["e_pose"][o]), e_point=np.ascontiguousarray(prob["e_point"][o]))
        a, f = _both(q)
        _assert_same(a, f, tag)
        for k in ("pinv", "chunk_pt0", "rm_chunk", "run_mf", "run_fl"):
            assert np.array_equal(f[k], f0[k]), (tag, k)
        # the same observation (point, key frame) sits at every internal position
        assert np.array_equal(q["e_pose"][f["perm"]], prob["e_pose"][f0["perm"]]) and np.array_equal(q["e_point"][f["perm"]], prob["e_point"][f0["perm"]]), tag
        assert np.array_equal(f["info"], f0["info"]), tag


def test_windows_the_device_planner_hands_back():
    # random views: (almost) every point its own signature -> no runs worth the name -> the look-ahead composition of the host planner
    prob = synth.ba_problem(K=20, P=4000, obs_per_point=4, F=550, seed=1, views="random")
    assert not api.ba_plan_fast(prob["fixed"], 4000, prob["e_pose"], prob["e_point"])["usable"]
    # a point seen twice by one key frame: the pair-owner kernel's case
    prob = synth.ba_problem(K=12, P=3000, obs_per_point=4, F=550, seed=3, views="track")
    ep = prob["e_pose"].copy()
    first = np.nonzero(prob["e_point"] == prob["e_point"][0])[0]
    ep[first[1]] = ep[first[0]]
    assert not api.ba_plan_fast(prob["fixed"], 3000, ep, prob["e_point"])["usable"]
    # every key frame fixed: nothing to plan
    assert not api.ba_plan_fast(np.ones(12, np.uint8), 3000, prob["e_pose"], prob["e_point"])["usable"]
    # an index out of range is an error, not a silent hand-back
    bad = prob["e_point"].copy(); bad[5] = 3000
    with pytest.raises(api.CmsError):
        api.ba_plan_fast(prob["fixed"], 3000, prob["e_pose"], bad)


def test_points_nobody_observes_and_two_fixed_key_frames():
    prob = synth.ba_problem(K=14, P=5000, obs_per_point=4, F=550, seed=21, views="track")
    keep = prob["e_point"] % 97 != 5                              # points 5, 102, ... lose all their observations
    q = dict(prob, e_pose=np.ascontiguousarray(prob["e_pose"][keep]), e_point=np.ascontiguousarray(prob["e_point"][keep]))
    fixed = prob["fixed"].copy(); fixed[3] = 1
    a = api.ba_plan(fixed, 5000, q["e_pose"], q["e_point"], tables=True)
    f = api.ba_plan_fast(fixed, 5000, q["e_pose"], q["e_point"])
    _assert_same(a, f, "lone points")
