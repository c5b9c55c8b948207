"""Host-only checks of the tables of the opt-in one-wavefront run workgroups of the Schur kernel (CMS_BA_RUN_WG=1,
cubemapslam_amd/csrc/cms_ba_schur_runwg.hip): where a lane's MFMA accumulators are added in the window's ONE global copy of the reduced system
(run_fg), the order of the runs by class and the cut points of the units.  No GPU, no oracle: list-construction logic, re-derived here from
the layout of v_mfma_f64_16x16x4_f64's accumulators and of BaSe::partial (42 doubles per pose pair s1 <= s2: the 6x6 block row major, then the
pair's right-hand side).  (block_solver.hpp:367-437 is what the sums feed.)"""
import numpy as np
import pytest

from cubemapslam_amd import api, synth

NONE = 0xFFFFFFFF
TILES = [(0, 0), (0, 1), (1, 1), (0, 2), (1, 2), (2, 2)]


def _pair(np_, s1, s2):
    return s1 * np_ - (s1 * (s1 - 1)) // 2 + (s2 - s1)


@pytest.mark.parametrize("K,P,views,seed", [(20, 22150, "track", 42), (12, 3000, "track", 3), (7, 900, "track", 5)])
def test_flush_table_covers_every_sum_of_a_signature_once(K, P, views, seed):
    prob = synth.ba_problem(K=K, P=P, obs_per_point=4, F=550, seed=seed, views=views)
    args = (prob["fixed"], len(prob["points"]), prob["e_pose"], prob["e_point"])
    host, fast = api.ba_run_fg(*args, fast=False), api.ba_run_fg(*args, fast=True)
    assert host is not None and host["n_runs"] > 0
    if fast is not None:                                      # both planners: the same tables
        assert np.array_equal(host["run_fg"], fast["run_fg"]) and np.array_equal(host["rm_cut"], fast["rm_cut"])
        assert (host["n_runs"], host["n_rm"], host["n_rmA"]) == (fast["n_runs"], fast["n_rm"], fast["n_rmA"])
    pl = api.ba_plan(*args, tables=True)
    np_, mf, fg = pl["np"], pl["run_mf"], host["run_fg"]
    assert host["np"] == np_ and host["n_runs"] == pl["n_runs"] and host["n_rm"] == pl["n_rm"]
    for r in range(pl["n_runs"]):
        kf = int(mf[r][56]); n6 = 6 * kf
        slots = [int(mf[r][48 + a]) for a in range(kf)]
        assert slots == sorted(slots) and len(set(slots)) == kf
        want = set()
        for a1 in range(kf):
            for a2 in range(a1, kf):
                base = _pair(np_, slots[a1], slots[a2]) * 42
                for r1 in range(6):
                    for r2 in range(6):
                        if a1 < a2 or r1 <= r2:
                            want.add(base + 6 * r1 + r2)             # diagonal pairs: the upper triangle only (the solve kernel mirrors it)
            for r1 in range(6):
                want.add(_pair(np_, slots[a1], slots[a1]) * 42 + 36 + r1)      # the right-hand side of the key frame's diagonal pair
        got = {}
        ntile = 3 if n6 + 1 <= 32 else 6
        for l in range(64):
            for idx in range(24):
                o = int(fg[r][l][idx])
                if o == NONE:
                    continue
                t, g = idx >> 2, idx & 3
                assert t < ntile                                     # class 0 never flushes a tile of the third tile column
                ti, tj = TILES[t]
                I, N = 16 * ti + (l >> 4) + 4 * g, 16 * tj + (l & 15)      # accumulator g of lane l holds G[I][N]
                assert I < n6 and N <= n6
                a1, r1 = divmod(I, 6)
                exp = _pair(np_, slots[a1], slots[a1]) * 42 + 36 + r1 if N == n6 else _pair(np_, slots[a1], slots[N // 6]) * 42 + 6 * r1 + N % 6
                assert o == exp and o not in got
                got[o] = (l, idx)
        assert set(got) == want, (r, kf, len(got), len(want))


@pytest.mark.parametrize("K,P,seed", [(20, 22150, 42), (12, 3000, 3)])
def test_runs_are_ordered_by_class_and_cut_by_cost(K, P, seed):
    prob = synth.ba_problem(K=K, P=P, obs_per_point=4, F=550, seed=seed, views="track")
    args = (prob["fixed"], len(prob["points"]), prob["e_pose"], prob["e_point"])
    t = api.ba_run_fg(*args)
    pl = api.ba_plan(*args, tables=True)
    kf_of_chunk = np.array([int(pl["run_mf"][int(c[2])][56]) for c in pl["rm_chunk"]])
    k_of_chunk = np.array([(int(c[1]) >> 8) & 255 for c in pl["rm_chunk"]])
    nA, n_rm = t["n_rmA"], t["n_rm"]
    assert np.all(6 * kf_of_chunk[:nA] + 1 <= 32) and np.all(6 * kf_of_chunk[nA:] + 1 > 32) and np.all(6 * kf_of_chunk + 1 <= 48)
    assert np.all(k_of_chunk <= 9)                                   # the own-block tasks of a chunk fit two rounds of 64 lanes
    assert np.all(np.diff(pl["rm_chunk"][:, 2]) >= 0)                # chunks of a run stay consecutive
    for cls, (lo, hi) in enumerate(((0, nA), (nA, n_rm))):
        cut = t["rm_cut"][cls]
        assert cut[0] == lo and cut[1024] == hi and np.all(np.diff(cut) >= 0)
        # units of any count tile the class's chunks without gaps or overlap
        for U in (7, 128, 192, 1024):
            b = [int(cut[(u * 1024) // U]) for u in range(U + 1)]
            assert b[0] == lo and b[-1] == hi and all(x <= y for x, y in zip(b, b[1:]))
        if hi - lo >= 64:                                            # ... of about equal estimated cost (ba_rm_chunk_cost, replayed)
            def cost(c):
                kf, m, k = int(kf_of_chunk[c]), int(pl["rm_chunk"][c][1]) >> 16, int(k_of_chunk[c])
                return 45 + (1 if kf <= 2 else 3 if kf <= 5 else 6) * 3 * ((m + 3) >> 2) + (0 if k in (2, 4) else 5)
            U = 32
            tot = [sum(cost(c) for c in range(int(cut[(u * 1024) // U]), int(cut[((u + 1) * 1024) // U]))) for u in range(U)]
            mean = sum(tot) / U
            assert max(tot) <= mean + 2 * max(cost(c) for c in range(lo, hi)), (cls, max(tot), mean)
