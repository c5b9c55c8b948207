"""Host side of the edge-major Schur kernel (no GPU): the chunk composition cms_ba_create builds (cms_ba_debug_compose).
Checked here: it is a valid work list (a permutation of the points, chunks of whole points with at most 64 observations, copy ranks in
range), the caller's order comes back when the look-ahead is off, and it does what it is for -- fewer LDS bank repeats per group of 16
lanes than the caller's order, under the cost model measured with tools/probe/lds_atomics.hip (a group of 16 consecutive lanes takes one
slot plus one per further lane on its fullest bank; the bank of a tuple is its pose pair's index mod 16)."""
import numpy as np
from cubemapslam_amd import api, synth

DSTRIDE = 33      # BA_SE_DSTRIDE (cms_ba_schur_edges.hip)


def slot_cost(prob, pinv, pt0, rank):
    """(off-diagonal slots, diagonal slots) summed over chunks, steps and 16-lane groups, in units of one conflict-free group"""
    fixed = prob["fixed"]; K = len(fixed)
    slot = np.full(K, -1); slot[fixed == 0] = np.arange(int((fixed == 0).sum())); npf = int((fixed == 0).sum())
    order = np.lexsort((prob["e_pose"], prob["e_point"]))
    ep, ek = prob["e_point"][order], prob["e_pose"][order]
    off = np.zeros(prob["points"].shape[0] + 1, np.int64); np.add.at(off, ep + 1, 1); off = np.cumsum(off)
    opair = lambda a, b: a * npf - a * (a + 1) // 2 + (b - a - 1)
    tot_off = tot_diag = 0
    for c in range(len(pt0) - 1):
        lane = 0
        per = {}          # (step, group) -> list of banks
        dg = {}
        for ip in range(pt0[c], pt0[c + 1]):
            p = pinv[ip]
            poses = ek[off[p]:off[p + 1]]; k = len(poses)
            for a in range(k):
                s = slot[poses[a]]
                if s >= 0:
                    dg.setdefault((lane + a) >> 4, []).append((DSTRIDE * (int(rank[off[p] + a]) * npf + s)) & 15)
            for dd in range(1, k // 2 + 1):
                for a in range(k):
                    if 2 * dd == k and a >= dd:
                        break
                    s1, s2 = slot[poses[a]], slot[poses[(a + dd) % k]]
                    if s1 < 0 or s2 < 0 or s1 == s2:
                        continue
                    per.setdefault((dd, (lane + a) >> 4), []).append(opair(min(s1, s2), max(s1, s2)) & 15)
            lane += k
        assert lane <= 64
        tot_off += sum(np.bincount(v, minlength=16).max() for v in per.values())
        tot_diag += sum(np.bincount(v, minlength=16).max() for v in dg.values())
    return tot_off, tot_diag


def check_valid(prob, pinv, pt0, rank):
    P = prob["points"].shape[0]
    assert sorted(pinv.tolist()) == list(range(P))
    assert pt0[0] == 0 and pt0[-1] == P and np.all(np.diff(pt0) > 0)
    npe = np.bincount(prob["e_point"], minlength=P)
    for c in range(len(pt0) - 1):
        assert npe[pinv[pt0[c]:pt0[c + 1]]].sum() <= 64
    assert rank.max() <= 3


def test_composition_is_a_valid_work_list_and_lowers_the_bank_repeats():
    prob = synth.ba_problem(K=20, P=3000, obs_per_point=4, F=550, seed=5)
    P = prob["points"].shape[0]
    pinv0, pt00, rank0 = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=1)
    check_valid(prob, pinv0, pt00, rank0)
    assert np.array_equal(pinv0, np.arange(P))                                   # look-ahead off: the caller's order
    pinv, pt0, rank = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=48)
    check_valid(prob, pinv, pt0, rank)
    o0, d0 = slot_cost(prob, pinv0, pt00, rank0)
    o1, d1 = slot_cost(prob, pinv, pt0, rank)
    _, dz = slot_cost(prob, pinv, pt0, np.zeros_like(rank))                      # every diagonal tuple on copy 0
    print("LDS slots per window (off-diagonal, diagonal): caller's order %d + %d, composed %d + %d (diagonal on one copy: %d)" % (o0, d0, o1, d1, dz))
    assert o1 <= 0.8 * o0 and d1 <= 0.8 * dz and len(pt0) <= len(pt00) + 2
    # deterministic
    pinv2, pt02, rank2 = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=48)
    assert np.array_equal(pinv, pinv2) and np.array_equal(pt0, pt02) and np.array_equal(rank, rank2)


def test_composition_edge_cases():
    # points without observations, a point with many observations, every key frame fixed but one, ragged observation counts
    rs = np.random.RandomState(3)
    K, P = 9, 400
    e_point, e_pose = [], []
    for p in range(P):
        k = 0 if p % 17 == 0 else (K if p == 5 else rs.randint(1, 7))
        for kf in rs.permutation(K)[:k]:
            e_point.append(p); e_pose.append(kf)
    prob = dict(fixed=np.array([1] + [0] * (K - 1), np.uint8), points=np.zeros((P, 3)), e_pose=np.array(e_pose, np.int32), e_point=np.array(e_point, np.int32))
    for la in (1, 8, 48):
        pinv, pt0, rank = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=la)
        check_valid(prob, pinv, pt0, rank)
    prob["fixed"] = np.array([1] * (K - 1) + [0], np.uint8)          # one free key frame: no pairs at all
    pinv, pt0, rank = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=48)
    check_valid(prob, pinv, pt0, rank)


def test_composition_in_segments_is_valid_and_repeatable():
    """Windows of more than 4096 points are composed in segments by several host threads; the segments are fixed by the number of points, so
    the result is the same on every run (and machine), a valid work list, and still better than the caller's order."""
    prob = synth.ba_problem(K=20, P=9000, obs_per_point=4, F=550, seed=8)
    P = prob["points"].shape[0]
    a = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=24)
    check_valid(prob, *a)
    for _ in range(3):
        b = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=24)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ident = api.ba_compose_chunks(prob["fixed"], P, prob["e_pose"], prob["e_point"], lookahead=1)
    assert np.array_equal(ident[0], np.arange(P))
    o1, d1 = slot_cost(prob, *a)
    o0, d0 = slot_cost(prob, *ident)
    assert o1 <= 0.8 * o0, (o1, o0)
