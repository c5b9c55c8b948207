"""Oracle restatements of the OpenCV primitives vs independent brute-force numpy versions (tests/npref.py)."""
import numpy as np
import orc
import npref
from cubemapslam_amd import synth


def test_cvround_half_even():
    L = orc.lib()
    assert [L.orc_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_atan2():
    rs = np.random.RandomState(1)
    L = orc.lib()
    for _ in range(3000):
        y, x = rs.randint(-50000, 50000, 2)
        a = L.orc_fast_atan2(float(y), float(x))
        assert np.float32(a) == npref.fast_atan2(y, x)
        if x or y:
            true = np.degrees(np.arctan2(float(y), float(x))) % 360
            assert min(abs(a - true), 360 - abs(a - true)) < 0.02
    assert L.orc_fast_atan2(0.0, 0.0) == 0.0


def test_fast_literal_vs_definition():
    rs = np.random.RandomState(2)
    for t in range(40):
        h, w = rs.randint(7, 45, 2)
        if t % 3 == 0:
            img = rs.randint(0, 256, (h, w)).astype(np.uint8)                 # dense corners, plateaus rare
        elif t % 3 == 1:
            img = (rs.randint(0, 4, (h, w)) * 60 + rs.randint(0, 3, (h, w))).astype(np.uint8)  # plateaus / ties
        else:
            img = synth.texture(h + 20, w + 20, t)[10:10 + h, 10:10 + w]
        for th in (7, 20):
            a = orc.fast(img, th)
            b = npref.fast_nms(img, th)
            assert a.shape == b.shape and np.array_equal(a, b), (t, th)


def test_resize_blur_remap_vs_numpy():
    img = synth.texture(300, 340, 5)
    for (dw, dh) in ((283, 250), (340, 300), (170, 150), (57, 40)):
        assert np.array_equal(orc.resize(img, dw, dh), npref.resize_linear(img, dw, dh))
    assert np.array_equal(orc.blur7(img), npref.blur7(img))
    small = img[:9, :11]
    assert np.array_equal(orc.blur7(small), npref.blur7(small))
    rs = np.random.RandomState(3)
    m1 = rs.uniform(-3, 343, (64, 80)).astype(np.float32)
    m2 = rs.uniform(-3, 303, (64, 80)).astype(np.float32)
    m1[0, :8] = [0, 339, 339.5, 340, -1, -0.5, 12.5, 12.015625]
    m2[0, :8] = [0, 299, 299.5, 300, -1, -0.5, 7.25, 7.984375]
    assert np.array_equal(orc.remap(img, m1, m2), npref.remap_bilinear(img, m1, m2))


def test_scoremap_formulation_equals_per_cell_fast():
    """The derivation the HIP FAST kernel relies on: one score map + per-cell NMS + ini/min
    fallback reproduces the reference's thousands of per-cell cv::FAST calls, including order."""
    camd = synth.camera("lafida", 150)
    cam = orc.make_camera(camd)
    img = synth.texture(450, 450, 11)
    img[100:220, 60:300] = (img[100:220, 60:300] // 16) + 100   # low-contrast area -> exercises the minTh fallback
    mask = np.full((450, 450), 255, np.uint8)
    o = orc.Orb(nfeatures=1000)
    o.extract(cam, img, mask)
    n_fallback = 0
    for l in range(8):
        lv = o.level(l)
        got = npref.level_candidates_scoremap(lv, 20, 7)
        want = o.candidates(l)
        assert got.shape == want.shape and np.array_equal(got, want), l
        n_fallback += int((want[:, 2] < 20).sum())
    assert n_fallback > 0


def test_gaussian_float_column_mode_differs_only_on_even_ties():
    """the SSE2 column functor of OpenCV <= 3.2 evaluates the 8-bit Gaussian's column pass in float and converts with
    round-half-to-even.  The oracle restates that literally (float products and sums, nearbyintf); every product and partial sum is exact in
    binary32, so the result must equal the integer formula everywhere except on ties with an even quotient, for columns x < width & ~3 --
    the rule the device kernel applies (cms_set_gaussian_mode)."""
    rng = np.random.default_rng(5)
    n_ties = 0
    for w, h in ((61, 47), (128, 64), (203, 77), (1650, 400)):
        # piecewise-constant images make exact ties frequent: 257 * v / 65536 ... force some by hand as well
        img = (rng.integers(0, 4, (h, w)) * 64 + rng.integers(0, 2, (h, w)) * 63).astype(np.uint8)
        img[: h // 3] = rng.integers(0, 256, (h // 3, w))
        a = orc.blur7(img, 0); b = orc.blur7(img, 1)
        # integer sums (kernel [18, 34, 49, 55, 49, 34, 18], REFLECT_101), straight from the definition
        k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
        pad = np.pad(img.astype(np.int64), 3, mode="reflect")
        rows = sum(k[i] * pad[:, i:i + w] for i in range(7))
        s = sum(k[i] * rows[i:i + h] for i in range(7))
        want0 = np.clip((s + 32768) >> 16, 0, 255)
        tie_even = ((s & 0xFFFF) == 0x8000) & (((s >> 16) & 1) == 0) & (np.arange(w)[None, :] < (w & ~3))
        want1 = np.where(tie_even, np.clip(s >> 16, 0, 255), want0)
        assert np.array_equal(a, want0.astype(np.uint8))
        assert np.array_equal(b, want1.astype(np.uint8))
        assert np.array_equal(a != b, tie_even)
        n_ties += int(tie_even.sum())
    assert n_ties >= 1            # about one pixel in 10^5: "1 LSB on rare pixels"
