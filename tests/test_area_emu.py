"""Frame::GetFeaturesInArea: the oracle's branch-by-branch transcription (oracle/orc_area.cpp) against
 (a) brute force on the cases where the reference's rules are plain geometry, and
 (b) the product's table (cubemapslam_amd/csrc/cms_area_table.h) compiled for the host -- two independent transcriptions of the
     reference's 41 cases / 90 AddCells calls must agree on every query, candidate order included."""
import ctypes as C
import os
import subprocess

import numpy as np

import orc
from cubemapslam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORIGIN = {0: (1, 1), 1: (0, 1), 2: (2, 1), 3: (1, 0), 4: (1, 2)}     # face -> (col, row) on the cross


def _emu():
    so = os.path.join(HERE, "emu", "libarea_emu.so")
    src = os.path.join(HERE, "emu", "area_emu.cpp")
    hdr = os.path.join(ROOT, "cubemapslam_amd", "csrc", "cms_area_table.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so])
    L = C.CDLL(so)
    L.area_emu.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    return L


def _keypoints(F, n, seed):
    rs = np.random.RandomState(seed)
    face = rs.randint(0, 5, n)
    x = np.array([ORIGIN[f][0] for f in face]) * F + rs.uniform(0, F, n)
    y = np.array([ORIGIN[f][1] for f in face]) * F + rs.uniform(0, F, n)
    return x.astype(np.float32), y.astype(np.float32), rs.randint(0, 8, n).astype(np.int32)


def _queries(F, nq, seed, edge_bias=True):
    """query centres on the five faces, biased towards face edges and corners; radii up to a few cells"""
    rs = np.random.RandomState(seed)
    face = rs.randint(0, 5, nq)
    u = rs.uniform(0, F, (nq, 2))
    if edge_bias:
        near = rs.uniform(size=(nq, 2)) < 0.6
        side = rs.uniform(size=(nq, 2)) < 0.5
        d = rs.uniform(0, 40, (nq, 2))
        u = np.where(near, np.where(side, d, F - 1e-3 - d), u)
    qx = (np.array([ORIGIN[f][0] for f in face]) * F + u[:, 0]).astype(np.float32)
    qy = (np.array([ORIGIN[f][1] for f in face]) * F + u[:, 1]).astype(np.float32)
    qr = rs.choice([3.0, 7.5, 15.0, 22.3, 36.0, 60.0], nq).astype(np.float32)
    lo = rs.randint(-1, 6, nq).astype(np.int32)
    hi = np.where(rs.uniform(size=nq) < 0.3, -1, lo + rs.randint(0, 3, nq)).astype(np.int32)
    return qx, qy, qr, lo, hi, face


def test_table_transcription_equals_branch_transcription():
    L = _emu()
    for F, seed in ((550, 1), (450, 2), (650, 3), (150, 4)):
        cam = orc.make_camera(synth.camera("lafida", F))
        kx, ky, ko = _keypoints(F, 3000, seed)
        qx, qy, qr, lo, hi, _ = _queries(F, 6000, 10 + seed)
        off_o, idx_o = orc.features_in_area(cam, kx, ky, ko, qx, qy, qr, lo, hi)
        cap = len(idx_o) + 16
        off_e = np.zeros(len(qx) + 1, np.int32); idx_e = np.zeros(cap, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        tot = L.area_emu(F, len(kx), p(kx), p(ky), p(ko), len(qx), p(qx), p(qy), p(qr), p(lo), p(hi), p(off_e), p(idx_e), cap)
        assert tot == len(idx_o), (F, tot, len(idx_o))
        assert np.array_equal(off_e, off_o) and np.array_equal(idx_e[:tot], idx_o)
        assert tot > 3000          # the queries do find things


def test_oracle_equals_brute_force_where_the_rules_are_plain_geometry():
    """window inside one face, or crossing one edge between FRONT and a neighbour: the result set is exactly the key points inside
    the window on those faces (canvas distance test), listed cell-major (ix outer, iy inner, index order inside a cell)."""
    F = 550
    cam = orc.make_camera(synth.camera("lafida", F))
    kx, ky, ko = _keypoints(F, 4000, 7)
    rs = np.random.RandomState(8)
    qs = []
    for _ in range(1500):
        r = float(rs.choice([5.0, 12.0, 30.0]))
        kind = rs.randint(0, 3)
        if kind == 0:      # strictly inside a random face
            f = rs.randint(0, 5)
            u, v = rs.uniform(r + 1, F - 2 - r, 2)
        elif kind == 1:    # FRONT, crossing exactly one edge
            f = 0
            u, v = rs.uniform(r + 1, F - 2 - r, 2)
            e = rs.randint(0, 4)
            if e == 0: u = rs.uniform(0, r * 0.9)
            elif e == 1: u = F - 1 - rs.uniform(0, r * 0.9)
            elif e == 2: v = rs.uniform(0, r * 0.9)
            else: v = F - 1 - rs.uniform(0, r * 0.9)
        else:              # a side face, crossing its edge towards FRONT
            f = rs.randint(1, 5)
            u, v = rs.uniform(r + 1, F - 2 - r, 2)
            if f == 1: u = F - 1 - rs.uniform(0, r * 0.9)
            elif f == 2: u = rs.uniform(0, r * 0.9)
            elif f == 3: v = F - 1 - rs.uniform(0, r * 0.9)
            else: v = rs.uniform(0, r * 0.9)
        qs.append((ORIGIN[f][0] * F + u, ORIGIN[f][1] * F + v, r))
    qx, qy, qr = (np.array(c, np.float32) for c in zip(*qs))
    lo = np.full(len(qx), -1, np.int32); hi = np.full(len(qx), -1, np.int32)
    off, idx = orc.features_in_area(cam, kx, ky, ko, qx, qy, qr, lo, hi)
    for q in range(len(qx)):
        got = idx[off[q]:off[q + 1]]
        inside = np.nonzero((np.abs(kx - qx[q]) < qr[q]) & (np.abs(ky - qy[q]) < qr[q]))[0]
        assert sorted(got.tolist()) == inside.tolist(), (q, qx[q], qy[q], qr[q])
        assert len(set(got.tolist())) == len(got)
    # level filter: minLevel / maxLevel semantics of AddCells (Frame.cpp:52-62)
    lo2 = np.full(len(qx), 2, np.int32); hi2 = np.full(len(qx), 4, np.int32)
    off2, idx2 = orc.features_in_area(cam, kx, ky, ko, qx, qy, qr, lo2, hi2)
    for q in range(0, len(qx), 7):
        a = idx[off[q]:off[q + 1]]
        assert np.array_equal(idx2[off2[q]:off2[q + 1]], a[(ko[a] >= 2) & (ko[a] <= 4)])


def test_reference_quirks_are_kept():
    """UPPER face with the window leaving through its top edge: the reference searches the LOWER face (Frame.cpp:373) -> nothing;
    a query centre on a corner block of the cross -> nothing."""
    F = 550
    cam = orc.make_camera(synth.camera("lafida", F))
    kx, ky, ko = _keypoints(F, 4000, 9)
    qx = np.array([F + 200.0, 30.0], np.float32); qy = np.array([5.0, 40.0], np.float32); qr = np.array([20.0, 20.0], np.float32)
    off, idx = orc.features_in_area(cam, kx, ky, ko, qx, qy, qr, np.array([-1, -1], np.int32), np.array([-1, -1], np.int32))
    assert off.tolist() == [0, 0, 0]
    assert ((np.abs(kx - qx[0]) < 20) & (np.abs(ky - qy[0]) < 20)).sum() > 0      # brute force would have found some
