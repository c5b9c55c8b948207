"""Oracle extractor: table known-answers, structural invariants, and the array-form octree."""
import numpy as np
import orc
import npref
from cubemapslam_amd import synth


def test_tables_match_survey():
    t = orc.Orb(nfeatures=2000).tables()
    assert list(t["quota"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(orc.Orb(nfeatures=3000).tables()["quota"]) == [652, 543, 452, 377, 314, 262, 218, 182]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert np.float32(t["scale"][1]) == np.float32(1.2) and abs(t["scale"][7] - 1.2 ** 7) < 1e-5


def test_level_sizes_match_survey():
    cam = orc.make_camera(synth.camera("lafida", 450))
    img = synth.texture(1350, 1350, 3)
    mask = synth.cubemap_valid_mask(synth.camera("lafida", 450))
    o = orc.Orb(nfeatures=2000)
    kps, desc = o.extract(cam, img, mask)
    assert [o.level(l).shape[1] for l in range(8)] == [1350, 1125, 937, 781, 651, 543, 452, 377]
    assert 500 < len(kps) <= 2000 and desc.shape == (len(kps), 32)
    # every survivor is on a face, under the mask, octaves ascending (level-major output)
    assert np.all(np.diff(kps["octave"]) >= 0)
    assert np.all(mask[(kps["y"] + 0.5).astype(int), (kps["x"] + 0.5).astype(int)] != 0)
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360.0001))
    assert desc.any(axis=1).all()
    # per level: no more than quota(+3) survivors before the cull, all inside [19, w-19)
    q = o.tables()["quota"]
    for l in range(8):
        d = o.distributed(l)
        w = o.level(l).shape[1]
        assert len(d) <= q[l] + 3
        assert np.all((d["x"] >= 19) & (d["x"] < w - 19) & (d["y"] >= 19) & (d["y"] < w - 19))


def test_octree_arrayform_equals_literal():
    rs = np.random.RandomState(7)
    for t in range(60):
        W = int(rs.choice([40, 97, 300, 1318]))
        n = int(rs.choice([0, 1, 2, 3, 5, 17, 200, 1500, 6000]))
        n = min(n, (W * W) // 4)
        N = int(rs.choice([1, 5, 99, 122, 150, 434, 652]))
        pos = rs.choice(W * W, size=n, replace=False)
        if t % 2:   # clustered
            cx, cy = rs.randint(0, W, 2)
            xs = np.clip((rs.normal(cx, W / 8, n)).astype(int), 0, W - 1); ys = np.clip((rs.normal(cy, W / 8, n)).astype(int), 0, W - 1)
            pos = np.unique(ys * W + xs)
            rs.shuffle(pos)
        xys = np.stack([pos % W, pos // W, rs.randint(7, 40, len(pos))], 1).astype(np.int32)
        a = orc.distribute_octree(xys, 16, 16 + W, 16, 16 + W, N)
        b = npref.octree_arrayform(xys, W, W, N)
        assert a.shape == b.shape and np.array_equal(a, b), (t, W, n, N)
