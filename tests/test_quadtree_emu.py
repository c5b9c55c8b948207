"""Product quadtree core (cms_quadtree_core.h) compiled for the host as a sequential emulation vs the oracle's
literal DistributeOctTree.  The HIP kernel runs the very same source with one workgroup per (frame, level)."""
import ctypes as C
import os
import subprocess
import numpy as np
import orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _emu():
    so = os.path.join(HERE, "emu", "libqt_emu.so")
    src = os.path.join(HERE, "emu", "qt_emu.cpp")
    hdr = os.path.join(ROOT, "cubemapslam_amd", "csrc", "cms_quadtree_core.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.dirname(hdr), src, "-o", so])
    return C.CDLL(so)


def _order_key(xys, wc):
    cx = (xys[:, 0] - 3) // wc; cy = (xys[:, 1] - 3) // wc
    return ((cy * 4096 + cx) * 4096 + xys[:, 1]) * 4096 + xys[:, 0]


def test_emulated_kernel_logic_equals_oracle():
    lib = _emu()
    rs = np.random.RandomState(12)
    for t in range(150):
        W = int(rs.choice([40, 97, 300, 1318, 1918]))
        n = int(rs.choice([0, 1, 2, 3, 5, 17, 200, 1500, 6000, 20000]))
        N = int(rs.choice([1, 5, 99, 122, 150, 434, 652, 1955]))
        wc = 31
        lo, hi = 3, W - 3          # FAST interior relative to minBorder
        n = min(n, ((hi - lo) ** 2) // 4)
        if t % 3 == 1:             # clustered
            cx, cy = rs.randint(lo, hi, 2)
            xs = np.clip(rs.normal(cx, W / 10, n).astype(int), lo, hi - 1); ys = np.clip(rs.normal(cy, W / 10, n).astype(int), lo, hi - 1)
        else:
            xs = rs.randint(lo, hi, n); ys = rs.randint(lo, hi, n)
        pos = np.unique(ys.astype(np.int64) * 8192 + xs)
        resp = rs.randint(7, 12 if t % 2 else 200, len(pos))     # few distinct responses -> many ties
        xys = np.stack([pos % 8192, pos // 8192, resp], 1).astype(np.int64)
        xys = xys[np.argsort(_order_key(xys, wc), kind="stable")].astype(np.int32)
        want = orc.distribute_octree(xys, 16, 16 + W, 16, 16 + W, N)
        out = np.zeros((N + 8, 3), np.int32)
        xin = np.ascontiguousarray(xys)
        S = lib.emu_quadtree(xin.ctypes.data_as(C.c_void_p), len(xin), W, W, N, wc, wc, out.ctypes.data_as(C.c_void_p))
        got = out[:S]
        assert got.shape == want.shape and np.array_equal(got, want), (t, W, len(xin), N)
