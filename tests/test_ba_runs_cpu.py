"""Host-only checks of the plan cms_ba_create makes for the run-major Schur kernel (cubemapslam_amd/csrc/cms_ba_schur_runs.hip): the
index arithmetic of the kernel -- which lane multiplies which pair of observations of which point, where its sum is flushed, which chunks a
producer / consumer pair walks -- is replayed here in Python on integer stand-ins for the 6x6 products, and has to reproduce the plain
double loop "for every point, for every pair of its free observations" exactly.  No GPU, no oracle: this is list-construction logic.
(The graph is what Optimizer::LocalBundleAdjustment hands to g2o, Optimizer.cpp:192-363; the Schur complement it feeds is block_solver.hpp:367-437.)"""
import numpy as np
import pytest

from cubemapslam_amd import api, synth

SSTRIDE, DSTRIDE, DCOPIES, PAIRS = 37, 33, 4, 4


def _opair(np_, s1, s2):
    return s1 * np_ - (s1 * (s1 + 1)) // 2 + (s2 - s1 - 1)


def _structure(prob):
    P, E = len(prob["points"]), len(prob["e_pose"])
    pl = api.ba_plan(prob["fixed"], P, prob["e_pose"], prob["e_point"], tables=True)
    assert pl["usable"]
    pinv, perm = pl["pinv"], pl["perm"]
    assert np.array_equal(np.sort(pinv), np.arange(P)) and np.array_equal(np.sort(perm), np.arange(E))
    prank = np.empty(P, np.int64); prank[pinv] = np.arange(P)
    s_pose = prob["e_pose"][perm]; s_pt = prank[prob["e_point"][perm]]
    assert np.all(np.diff(s_pt) >= 0)                                        # sorted by internal point ...
    same = np.diff(s_pt) == 0
    assert np.all(np.diff(s_pose)[same] > 0)                                 # ... and by key frame inside a point (no key frame twice)
    pt_off = np.searchsorted(s_pt, np.arange(P + 1))
    slot_of = np.full(len(prob["fixed"]), -1); slot_of[prob["fixed"] == 0] = np.arange(int((prob["fixed"] == 0).sum()))
    info = pl["info"]
    a = info & 31; k = (info >> 5) & 31; s = ((info >> 10) & 63).astype(int) - 1; kp = (info >> 19) & 255
    assert np.array_equal(a, np.arange(E) - pt_off[s_pt]) and np.array_equal(k, (pt_off[1:] - pt_off[:-1])[s_pt])
    assert np.array_equal(s, slot_of[s_pose]) and np.array_equal(kp, s_pose)
    c0 = pl["chunk_pt0"]
    assert c0[0] == 0 and c0[-1] == P and np.all(np.diff(c0) > 0)
    assert np.all(pt_off[c0[1:]] - pt_off[c0[:-1]] <= 64)                    # a chunk is one wavefront
    return pl, pt_off, s_pose, s_pt, slot_of, a, s


@pytest.mark.parametrize("K,P,views,seed", [(20, 22150, "track", 42), (12, 3000, "track", 3), (7, 900, "track", 5), (20, 4000, "random", 1)])
def test_run_plan_replays_the_schur_sum(K, P, views, seed):
    prob = synth.ba_problem(K=K, P=P, obs_per_point=4, F=550, seed=seed, views=views)
    pl, pt_off, s_pose, s_pt, slot_of, a_of, s_of = _structure(prob)
    E = len(s_pose); np_ = pl["np"]; NP2 = np_ * (np_ + 1) // 2
    n_rm, rmc, rl = pl["n_rm"], pl["rm_chunk"], pl["run_lane"]
    c0 = pl["chunk_pt0"]
    if views == "track" and P >= 3000:
        assert pl["rm_points"] > 0.5 * P and n_rm > 0                       # tracked points share their signatures: most of a large window is in runs
    # ---- run chunks: whole points of ONE signature, descriptor fields
    P_rm = pl["rm_points"]
    assert (c0[n_rm] if n_rm < len(c0) else P) == P_rm
    for c in range(n_rm):
        e0, word, run, pfirst = (int(v) for v in rmc[c])
        ne, k, m = word & 255, (word >> 8) & 255, word >> 16
        invk = (65536 + k - 1) // k
        p0, p1 = c0[c], c0[c + 1]
        assert e0 == pt_off[p0] and ne == pt_off[p1] - pt_off[p0] and m == p1 - p0 and ne == k * m and pfirst == p0
        assert np.array_equal(s_pt[e0:e0 + ne], pfirst + ((np.arange(ne) * invk) >> 16))      # a lane's point follows from the chunk's first point
        assert 0 <= run < pl["n_runs"] and m <= 32
        sig = s_pose[pt_off[p0]:pt_off[p0] + k]
        assert np.array_equal(s_pose[e0:e0 + ne].reshape(m, k), np.tile(sig, (m, 1)))
        if c > 0 and rmc[c - 1][2] == run:                                   # chunks of a run are consecutive and share the signature
            pe0 = int(rmc[c - 1][0])
            assert np.array_equal(s_pose[pe0:pe0 + k], sig)
        lanes = np.arange(ne)
        assert np.array_equal(((lanes - a_of[e0:e0 + ne]) * invk) >> 16, lanes // k)      # the kernel's point-of-lane arithmetic
    # ---- integer stand-ins: w per edge (0 for fixed key frames: W = 0 there), d per point, h per edge (the key frame's own block)
    rs = np.random.RandomState(seed)
    w = rs.randint(1, 50, E).astype(np.int64) * (s_of >= 0); dpt = rs.randint(1, 9, P).astype(np.int64); h = rs.randint(1, 1000, E).astype(np.int64) * (s_of >= 0)
    ref_S = np.zeros((np_, np_), np.int64); ref_h = np.zeros(np_, np.int64)
    for p in range(P_rm):
        es = np.arange(pt_off[p], pt_off[p + 1]); es = es[s_of[es] >= 0]
        for i, ea in enumerate(es):
            ref_h[s_of[ea]] += h[ea]
            for eb in es[i:]:
                ref_S[s_of[ea], s_of[eb]] += w[ea] * dpt[p] * w[eb]
    dg_off = ((NP2 - np_) * SSTRIDE + 1) & ~1
    for R_rm in sorted({max(pl["R_rm"], 1), 1, 7}):
        if n_rm == 0:
            break
        lds = np.zeros(dg_off + DCOPIES * np_ * DSTRIDE, np.int64)
        hsum = np.zeros((DCOPIES, np_), np.int64)
        total_pairs = R_rm * PAIRS
        covered = 0
        for g in range(total_pairs):
            cb, ce = g * n_rm // total_pairs, (g + 1) * n_rm // total_pairs
            covered += ce - cb
            acc = np.zeros(64, np.int64); hp = np.zeros(64, np.int64); hp_slot = np.full(64, -1)
            cur_run = -1; lt = None
            for c in range(cb, ce):
                e0, word, run, _ = (int(v) for v in rmc[c])
                ne, k, m = word & 255, (word >> 8) & 255, word >> 16
                invk = (65536 + k - 1) // k
                nxt = int(rmc[c + 1][2]) if c + 1 < ce else -1
                # producer
                for lane in range(ne):
                    e = e0 + lane
                    if s_of[e] >= 0:
                        hp_slot[lane] = s_of[e]; hp[lane] += h[e]
                if nxt != run:
                    for lane in range(64):
                        if hp_slot[lane] >= 0:
                            j = ((lane - (a_of[e0 + lane] if lane < ne else 0)) * invk) >> 16
                            hsum[j & (DCOPIES - 1), hp_slot[lane]] += hp[lane]
                    hp[:] = 0; hp_slot[:] = -1
                # consumer
                if run != cur_run:
                    cur_run = run; lt = rl[run]
                for lane in range(64):
                    x, y = int(lt[lane][0]), int(lt[lane][1])
                    if not (x >> 23) & 1:
                        continue
                    pa, pb, q0, Q = x & 31, (x >> 5) & 31, (x >> 10) & 63, (x >> 16) & 127
                    for j in range(q0, m, Q):
                        ea, eb = e0 + j * k + pa, e0 + j * k + pb
                        acc[lane] += w[ea] * dpt[s_pt[ea]] * w[eb]
                if nxt != cur_run:
                    for lane in range(64):
                        x, y = int(lt[lane][0]), int(lt[lane][1])
                        if (x >> 23) & 1:
                            lds[y] += acc[lane]
                    acc[:] = 0
        assert covered == n_rm
        got_S = np.zeros((np_, np_), np.int64)
        for s1 in range(np_):
            for s2 in range(s1 + 1, np_):
                got_S[s1, s2] = lds[_opair(np_, s1, s2) * SSTRIDE]
            got_S[s1, s1] = sum(lds[dg_off + (cp * np_ + s1) * DSTRIDE] for cp in range(DCOPIES))
        assert np.array_equal(got_S, ref_S), R_rm
        assert np.array_equal(hsum.sum(0), ref_h), R_rm
    # ---- lane tables: every tuple of a run's free observations exactly Q times (sequences 0 .. Q - 1), diagonal flag and targets consistent
    for r in range(pl["n_runs"]):
        c = int(np.flatnonzero(rmc[:, 2] == r)[0])
        e0, word = int(rmc[c][0]), int(rmc[c][1])
        k = (word >> 8) & 255
        free = [i for i in range(k) if s_of[e0 + i] >= 0]
        T = len(free) * (len(free) + 1) // 2
        seen = {}
        for lane in range(64):
            x, y = int(rl[r][lane][0]), int(rl[r][lane][1])
            if not (x >> 23) & 1:
                continue
            pa, pb, q0, Q, diag = x & 31, (x >> 5) & 31, (x >> 10) & 63, (x >> 16) & 127, (x >> 24) & 1
            assert pa in free and pb in free and pa <= pb and diag == (pa == pb) and Q == 64 // T and q0 < Q
            sa, sb = s_of[e0 + pa], s_of[e0 + pb]
            if diag:
                assert y == dg_off + ((q0 % DCOPIES) * np_ + sa) * DSTRIDE
            else:
                assert sa < sb and y == _opair(np_, sa, sb) * SSTRIDE
            seen.setdefault((pa, pb), []).append(q0)
        assert len(seen) == T and all(sorted(v) == list(range(64 // T)) for v in seen.values())


def test_range_split_and_left_over_chunks():
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=7, views="track")
    pl, pt_off, s_pose, s_pt, slot_of, a_of, s_of = _structure(prob)
    # default: the run-major body's workgroups take every chunk of the window (its wavefronts' ranges are cut by cost over all of them)
    assert pl["R_rm"] >= 1 and pl["R"] == 0 and pl["R_rm"] <= 128 and pl["n_rm"] < pl["n_chunks"]
    assert pl["R_rm"] * 8 <= max(pl["n_chunks"], 8) + 7                        # a wavefront has at least one chunk
    import os, subprocess, sys
    # CMS_BA_SPLIT_WORKGROUPS (child process: the switches are read once): separate workgroups for run chunks and left-over chunks
    code = ("import sys; sys.path.insert(0, %r); from cubemapslam_amd import api, synth; "
            "p = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=7, views='track'); "
            "pl = api.ba_plan(p['fixed'], len(p['points']), p['e_pose'], p['e_point']); print(pl['R_rm'], pl['R'], pl['n_rm'])"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    R_rm, R, n_rm = (int(v) for v in subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CMS_BA_SPLIT_WORKGROUPS="1"), capture_output=True, text=True,
                                                     check=True).stdout.split())
    assert R_rm >= 1 and R >= 1 and R_rm + R <= 128 and R_rm * 8 <= max(n_rm, 8) + 7
    # with CMS_BA_NO_RUNS every point is a left-over point and the plan is the old composition
    code = ("import sys; sys.path.insert(0, %r); from cubemapslam_amd import api, synth; "
            "p = synth.ba_problem(K=20, P=3000, obs_per_point=4, F=550, seed=7, views='track'); "
            "pl = api.ba_plan(p['fixed'], len(p['points']), p['e_pose'], p['e_point']); print(pl['n_rm'], pl['n_runs'], pl['rm_points'], pl['n_chunks'])"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CMS_BA_NO_RUNS="1"), capture_output=True, text=True, check=True).stdout.split()
    assert out[:3] == ["0", "0", "0"] and int(out[3]) > 0


@pytest.mark.parametrize("K,P,obs,seed", [(20, 6000, 4, 11), (16, 5000, 6, 12), (9, 2500, 2, 13)])
def test_mfma_tables_replay_the_schur_sum(K, P, obs, seed):
    """The MFMA variant of the run-major body (ba_schur_runs_mfma_body): a signature's W rows are stacked into a (6 kf) x (6 kf + 1) Gram-like
    product evaluated in 16 x 16 tiles.  run_mf says which W entry a lane feeds into row / column i of the stacked matrix, run_fl where each
    of a lane's accumulators is added in the LDS copy.  Replayed here with integer stand-ins: G = sum_j Y_j D_j Y_j^T per run from the table's
    row entries, the accumulator -> LDS map applied lane by lane, and the result compared with the plain double loop over tuples."""
    prob = synth.ba_problem(K=K, P=P, obs_per_point=obs, F=550, seed=seed, views="track", dropout=0.05)
    if seed == 12:
        prob["fixed"][3] = 1
    pl, pt_off, s_pose, s_pt, slot_of, a_of, s_of = _structure(prob)
    np_ = pl["np"]; NP2 = np_ * (np_ + 1) // 2
    n_rm, rmc, mf, fl = pl["n_rm"], pl["rm_chunk"], pl["run_mf"], pl["run_fl"]
    assert n_rm > 0 and pl["n_runs"] > 3
    dg_off = ((NP2 - np_) * SSTRIDE + 1) & ~1
    NONE, RHS = 0xFFFF, 0xFFFE
    tile_i, tile_j = [0, 0, 1, 0, 1, 2], [0, 1, 1, 2, 2, 2]
    rs = np.random.RandomState(seed)
    E = len(s_pose)
    Wst = rs.randint(1, 9, (E, 6, 3)).astype(np.int64) * (s_of >= 0)[:, None, None]      # stand-in for W (zero for fixed key frames)
    Dst = rs.randint(1, 5, (P, 3)).astype(np.int64); yst = rs.randint(1, 7, (P, 3)).astype(np.int64)
    lds = np.zeros(dg_off + DCOPIES * np_ * DSTRIDE, np.int64)
    ref_S = {}; ref_rhs = np.zeros((np_, 6), np.int64)
    for c in range(n_rm):
        e0, word, run, p0 = (int(v) for v in rmc[c])
        ne, k, m = word & 255, (word >> 8) & 255, word >> 16
        tab = mf[run]; kf = int(tab[56]); n6 = 6 * kf
        assert n6 + 1 <= 48 and int(tab[n6]) == RHS and all(int(tab[i]) == NONE for i in range(n6 + 1, 48))
        G = np.zeros((48, 48), np.int64)
        for j in range(m):
            pnt = p0 + j
            Y = np.zeros((48, 3), np.int64)
            for i in range(n6):
                ent = int(tab[i]); pos, r = ent // 18, (ent % 18) // 3
                assert ent % 3 == 0 and pos < k and r == i % 6
                e = e0 + j * k + pos
                assert s_of[e] == int(tab[48 + i // 6])              # the slot the table names is the edge's key frame
                Y[i] = Wst[e, r]
            Yc = Y.copy(); Yc[n6] = yst[pnt]                          # the rhs column
            G += (Y * Dst[pnt]) @ Yc.T
            # reference: the tuples of this point
            es = [e0 + j * k + a for a in range(k) if s_of[e0 + j * k + a] >= 0]
            for ia, ea in enumerate(es):
                ref_rhs[s_of[ea]] += (Wst[ea] * Dst[pnt]) @ yst[pnt]
                for eb in es[ia:]:
                    key = (s_of[ea], s_of[eb])
                    ref_S[key] = ref_S.get(key, 0) + (Wst[ea] * Dst[pnt]) @ Wst[eb].T
        # the accumulators of every lane -> LDS, as the kernel adds them (resident tiles at the run's end, the third tile column per chunk:
        # the sum over chunks is the same either way)
        for lane in range(64):
            for t in range(6):
                for g in range(4):
                    w = int(fl[run][lane][(4 * t + g) // 2]); off = (w >> 16) if (4 * t + g) & 1 else (w & 0xFFFF)
                    if off != 0xFFFF:
                        lds[off] += G[16 * tile_i[t] + (lane >> 4) + 4 * g, 16 * tile_j[t] + (lane & 15)]
    # read the LDS copy back the way the write-out does
    def off_el(r, q):
        up = lambda a, b: a * 5 - (a * (a - 1)) // 2 + (b - a - 1)
        return up(r, q) if r < q else 16 + up(q, r) if r > q else (15 if r == 0 else 31 if r == 1 else 30 + r)
    for (s1, s2), blk in ref_S.items():
        for r in range(6):
            for q in range(6):
                if s1 == s2:
                    if r > q:
                        continue
                    got = sum(lds[dg_off + (cp * np_ + s1) * DSTRIDE + (r * 6 - (r * (r - 1)) // 2 + (q - r))] for cp in range(DCOPIES))
                else:
                    got = lds[_opair(np_, s1, s2) * SSTRIDE + off_el(r, q)]
                assert got == blk[r, q], (s1, s2, r, q)
    for s1 in range(np_):
        for r in range(6):
            assert sum(lds[dg_off + (cp * np_ + s1) * DSTRIDE + 21 + r] for cp in range(DCOPIES)) == ref_rhs[s1, r]
    # nothing was added anywhere else
    used = np.zeros(len(lds), bool)
    for (s1, s2) in ref_S:
        if s1 != s2:
            used[_opair(np_, s1, s2) * SSTRIDE:_opair(np_, s1, s2) * SSTRIDE + 36] = True
    used[dg_off:] = True
    assert not lds[~used].any()
