"""The fixed-order protocol of the deterministic Schur kernels (cms_ba_schur_edges.hip: ba_det_publish / ba_det_wait), as a model on the CPU.

The kernels give every set of LDS additions a key from the window's plan (estimated cost of the wavefront's chunks so far) and perform the sets in ascending
(key, wavefront) order: L[w] holds a lower bound of wavefront w's next key, published at every chunk start once the previous additions are through; a
wavefront adds when (L[w'], w') > (key, w) for every other wavefront.  This test runs that protocol as a discrete-event simulation with RANDOM chunk
durations (the thing that differs from run to run on the device) and checks what the design claims: no deadlock, and an order of additions that depends on
the keys alone -- for chunk sequences with and without additions (a run flushes only at its end), equal costs in several wavefronts, empty wavefronts.
The GPU tests hold the kernels to the same claim through bit-identical results (tests/test_gpu_parity.py::test_ba_deterministic_*)."""
import heapq
import numpy as np

DONE = 0xFFFFFFFF


def simulate(plan, rng):
    """plan[w] = [(cost, flushes)] per chunk of wavefront w.  Returns the order of the additions as (key, wavefront) pairs."""
    nw = len(plan)
    L = [0] * nw                                   # det_L_[w] = 0 before the first barrier
    order = []
    t_now = [0.0] * nw
    pos = [0] * nw                                 # next chunk of the wavefront
    acc = [0] * nw                                 # running cost = the key of the chunk being worked on
    state = ["start"] * nw
    ev = [(rng.uniform(0, 1), w) for w in range(nw)]
    heapq.heapify(ev)
    waiting = {}
    steps = 0

    def enc(t, w):
        return (t << 3) | w

    def try_release():
        for w in list(waiting):
            key = waiting[w]
            if all(v == w or L[v] > key for v in range(nw)):
                del waiting[w]
                order.append((key >> 3, w))
                heapq.heappush(ev, (max(t_now) + rng.uniform(0.01, 0.3), w))       # the additions take a while; the bound moves at the next chunk start
                state[w] = "added"

    while ev or waiting:
        steps += 1
        assert steps < 200000, "no progress: deadlock"
        if not ev:
            try_release()
            assert ev, "every wavefront waits: deadlock"
            continue
        t, w = heapq.heappop(ev)
        t_now[w] = t
        if state[w] in ("start", "added", "computed_noflush"):
            if pos[w] == len(plan[w]):
                L[w] = DONE                         # ba_det_publish(.., BA_DET_DONE)
                state[w] = "done"
            else:
                cost, _ = plan[w][pos[w]]
                acc[w] += cost
                L[w] = enc(acc[w], w)               # chunk start: "my next additions have at least this key"
                state[w] = "computing"
                heapq.heappush(ev, (t + cost * rng.uniform(0.5, 2.0), w))           # vector / matrix phase: timing varies from run to run
        elif state[w] == "computing":
            _, flushes = plan[w][pos[w]]
            pos[w] += 1
            if flushes:
                waiting[w] = enc(acc[w], w)         # ba_det_wait
            else:
                state[w] = "computed_noflush"
                heapq.heappush(ev, (t, w))
        try_release()
    assert all(v == DONE for v in L)
    return order


def make_plan(rng, nw=8, equal_costs=False):
    plan = []
    for w in range(nw):
        n = int(rng.integers(0, 12))
        if w == 3:
            n = 0                                   # a wavefront without a chunk
        chunks = []
        for i in range(n):
            cost = 60 if equal_costs else int(rng.integers(40, 120))
            chunks.append((cost, bool(rng.random() < 0.4) or i == n - 1))          # the last chunk of a range always flushes
        plan.append(chunks)
    return plan


def test_the_order_of_additions_follows_from_the_keys_alone():
    for seed in range(12):
        prng = np.random.default_rng(seed)
        plan = make_plan(prng, equal_costs=(seed % 3 == 0))
        want = []
        for w, chunks in enumerate(plan):
            a = 0
            for cost, fl in chunks:
                a += cost
                if fl:
                    want.append((a, w))
        want.sort()
        for timing in range(6):
            got = simulate(plan, np.random.default_rng(1000 * seed + timing))
            assert got == want, (seed, timing)


def test_chunk_index_keys_give_chunk_order_whatever_the_wavefront_count():
    """the edge-major body: wavefront w takes chunks w, w + nw, ...; key = the chunk's index, every chunk adds -- the order is chunk order for any nw"""
    for nw in (2, 4, 6, 8):
        nchunks = 37
        plan = [[] for _ in range(nw)]
        keys = [[] for _ in range(nw)]
        for c in range(nchunks):
            keys[c % nw].append(c)
        for w in range(nw):
            prev = -1
            for c in keys[w]:
                plan[w].append((c - prev if prev >= 0 else c + 1, True))            # running sum = c + 1: the model's key for chunk c
                prev = c
        got = simulate(plan, np.random.default_rng(nw))
        assert [k - 1 for k, _ in got] == list(range(nchunks)), nw
