"""Trajectory output format: System::SaveKeyFrameTrajectoryTUM in the C++ mirror and the Python writer rank 0
uses after the RCCL gather must produce the reference's line format -- and the same bytes as each other."""
import ctypes as C
import os

import numpy as np

from cubemapslam_amd import build, dist


def _poses(n, seed):
    rs = np.random.RandomState(seed)
    T = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        a = rs.normal(size=3); a /= np.linalg.norm(a)
        ang = rs.uniform(-3.1, 3.1)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        T[i, :3, :3] = R; T[i, :3, 3] = rs.uniform(-5, 5, 3); T[i, 3, 3] = 1
    return T


def test_tum_writer_format_and_mirror_agreement(tmp_path):
    build.build(verbose=False)
    L = C.CDLL(build.HOST_LIB)
    L.hm_save_trajectory_tum.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    n = 40
    T = _poses(n, 5)
    ts = 1403636579.763556 + 0.05 * np.arange(n)
    p_cpp, p_py = str(tmp_path / "cpp.txt"), str(tmp_path / "py.txt")
    assert L.hm_save_trajectory_tum(p_cpp.encode(), n, ts.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p)) == n
    dist.write_trajectory_tum(p_py, ts, T)
    a, b = open(p_cpp).read(), open(p_py).read()
    assert a == b
    lines = a.strip().split("\n")
    assert len(lines) == n
    f0 = lines[0].split(" ")
    assert len(f0) == 8 and f0[0] == "1403636579.763556" and all(len(x.split(".")[1]) == 7 for x in f0[1:])
    # identity pose at t = 0: centre 0 (printed as -0.0000000 by "-acc", like the reference's -R^T t), unit quaternion
    I = np.eye(4, dtype=np.float32)[None]
    dist.write_trajectory_tum(p_py, [0.0], I)
    assert open(p_py).read() == "0.000000 -0.0000000 -0.0000000 -0.0000000 0.0000000 0.0000000 0.0000000 1.0000000\n"
    # the centre really is -R^T t
    vals = np.array([[float(x) for x in l.split(" ")] for l in lines])
    want = -np.einsum("nji,nj->ni", T[:, :3, :3].astype(np.float64), T[:, :3, 3].astype(np.float64))
    assert np.abs(vals[:, 1:4] - want).max() < 2e-5
