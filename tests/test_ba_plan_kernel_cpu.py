"""The plan kernel's matching (cms_api_ba_devplan.hip: ba_dp_match, the augmenting-path search without recursion and with its state in nibbles of 64-bit
words) against the host planners' BaDiagMatch::run on the host -- no GPU needed: the function is __host__ __device__ and cms_ba_debug_match runs either."""
import ctypes as C

import numpy as np

from cubemapslam_amd import api


def _match(which, slots, npf):
    L = api.lib()
    L.cms_ba_debug_match.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    s = (C.c_int * 16)(*slots)
    out = (C.c_int * 16)()
    assert L.cms_ba_debug_match(which, len(slots), s, npf, out) == 0
    return list(out)[:len(slots)]


def test_register_only_matching_equals_the_recursive_one():
    rng = np.random.default_rng(11)
    searched = 0
    for case in range(20000):
        npf = int(rng.integers(1, 63))
        n = int(rng.integers(0, 17))
        # a few slots repeated many times force collisions (the augmenting path and the load-balancing tail)
        pool = rng.integers(0, npf, size=int(rng.integers(1, 6))) if case % 3 == 0 else np.arange(npf)
        slots = [int(pool[rng.integers(0, len(pool))]) for _ in range(n)]
        a, b = _match(0, slots, npf), _match(1, slots, npf)
        assert a == b, (case, npf, slots, a, b)
        assert all(0 <= c < 4 for c in a)
        banks = [(33 * (c * npf + s)) & 15 for c, s in zip(a, slots)]
        searched += len(set(banks)) < len(banks)
    assert searched > 500          # (cases in which lanes had to share a bank: the tail of the search ran)
