"""The reference's own shipped inputs -- Masks/*.png and Config/*.yaml -- as committed fixtures: they exist, they say what
the synthetic harness assumes, and they agree with the oracle's camera model.  The last point is one of the few places where the oracle meets
reference-held data: the masks were drawn by the reference's authors around the part of the cubemap their camera model fills."""
import ctypes as C
import os
import numpy as np
import pytest
import orc
import refdata
from cubemapslam_amd import api, synth

REF = "/root/reference"


@pytest.mark.parametrize("name,F", [("lafida", 450), ("lafida", 550), ("lafida", 650), ("front", 650)])
def test_reference_masks_lie_inside_the_oracle_camera_models_cubemap(name, F):
    """Every non-zero pixel of the reference's mask is a cubemap pixel the oracle's CubemapToFisheye maps INTO the fisheye image (20 pixels of slack
    on the 450 mask, none elsewhere), the mask is a strict erosion of that region (it covers 70-96 % of it), and it is zero on the four corner
    blocks of the cross (cubemap_lafida.cpp:110-111) -- but for one 20-pixel run on the 450 mask."""
    m = refdata.reference_mask(name, F) != 0
    camd = synth.camera(name, F); ocam = orc.make_camera(camd)
    m1, m2 = orc.build_lut(ocam)
    inside = (m1 > 0) | (m2 > 0)                      # LUT entries never written stay (0, 0) (System.cpp:305-322)
    assert m.shape == inside.shape == (3 * F, 3 * F)
    assert int((m & ~inside).sum()) <= (20 if F == 450 else 0), int((m & ~inside).sum())
    assert 0.70 <= m.sum() / inside.sum() <= 0.97, m.sum() / inside.sum()
    faces = np.zeros_like(m)
    for (ox, oy) in synth._FACE_ORIGIN.values():
        faces[oy * F:(oy + 1) * F, ox * F:(ox + 1) * F] = True
    # (the 450 mask's 20 stray pixels sit in row 900 = the first row of a corner block: FaceInCubemap calls them UNKNOWN_FACE and the cull at
    # ORBExtractor.cpp:891-892 drops a key point there before it ever looks at the mask)
    assert int((m & ~faces).sum()) == (20 if F == 450 else 0)


def test_reference_configs_are_what_the_harness_assumes():
    """Config/lafida_cam0_params.yaml and Config/front_cam_params.yaml, parsed by the product's reader of the reference's settings format
    (io_formats.cpp), give exactly synth.LAFIDA / synth.FRONT; left_cam differs from front_cam (another camera of the same rig)."""
    for key, want in (("lafida_cam0", synth.LAFIDA), ("front_cam", synth.FRONT)):
        cam_b, orb_b, misc = refdata.reference_config(key)
        cam = api.Camera.from_buffer_copy(cam_b); orb = api.OrbParams.from_buffer_copy(orb_b)
        assert cam.face == 650                                      # the shipped files are set up for the 650 masks
        camd = synth.camera("lafida" if key == "lafida_cam0" else "front", 650)
        assert bytes(api.make_camera(camd)) == cam_b, key           # every double identical
        assert (orb.nfeatures, orb.nlevels, orb.ini_th_fast, orb.min_th_fast) == (want["nfeatures"], 8, 20, 7) and abs(orb.scale_factor - 1.2) < 1e-7
        assert list(misc) == [30.0, 1.0, 1.0]                       # fps, withFisheyeMask, RGB
    assert refdata.reference_config("left_cam")[0] != refdata.reference_config("front_cam")[0]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")
def test_fixtures_are_current_with_the_reference_checkout():
    """the committed fixtures against the files they were made from (build container only)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_reference_fixtures as mk
    from PIL import Image
    for (name, F), key in refdata.MASK_KEYS.items():
        a = np.asarray(Image.open(os.path.join(REF, mk.MASKS[key])))
        assert np.array_equal(refdata.reference_mask(name, F) != 0, a != 0), key
    for key, rel in mk.CONFIGS.items():
        cam, orb, fps, wm, rgb = mk.load_settings(os.path.join(REF, rel))
        cam_b, orb_b, misc = refdata.reference_config(key)
        assert bytes(cam) == cam_b and bytes(orb) == orb_b and list(misc) == [fps, wm, rgb], key


def test_oracle_extraction_culls_with_the_reference_mask():
    """ORBExtractor.cpp:887-904 on the real, irregular mask edge: no surviving key point sits on a zero pixel (rounded like the reference rounds),
    and the real mask removes key points the model-derived synthetic mask would keep or vice versa (the two are different shapes)."""
    camd = synth.camera("lafida", 450); ocam = orc.make_camera(camd)
    real = refdata.reference_mask("lafida", 450)
    m1, m2 = orc.build_lut(ocam)
    cube = orc.fisheye_to_cubemap(ocam, m1, m2, synth.texture(camd["Ih"], camd["Iw"], 5))
    k, d = orc.Orb(nfeatures=2000).extract(ocam, cube, real)
    assert len(k) > 800
    xi = (k["x"] + 0.5).astype(int); yi = (k["y"] + 0.5).astype(int)
    assert np.all(real[yi, xi] != 0)
    k2, _ = orc.Orb(nfeatures=2000).extract(ocam, cube, synth.cubemap_valid_mask(camd))
    assert len(k2) != len(k) or not np.array_equal(k2["x"], k["x"])
