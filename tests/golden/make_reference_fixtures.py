"""Reference-held INPUT DATA as test fixtures (build container only: /root/reference does not exist on the GPU box).

    python tests/golden/make_reference_fixtures.py

Reads, from the reference checkout,
  * Masks/gray_lafida_cubemap_mask_{450,550,650}.png, Masks/gray_cubemap_{front,left}_mask_650.png -- the cubemap masks ORBextractor culls
    key points with (ORBExtractor.cpp:887-904; loaded by Examples/cubemap_lafida.cpp / cubemap_fangshan.cpp) -- with Pillow, and
  * Config/{lafida_cam0,front_cam,left_cam}_params.yaml with the product's own reader of the reference's settings format
    (cubemapslam_amd/host/io_formats.cpp, through libcubemapslam_host.so),
checks the parsed intrinsics against the constants the synthetic harness carries (cubemapslam_amd/synth.py LAFIDA / FRONT) and writes

    tests/golden/reference_masks.npz      the masks, bit-packed rows (1 = non-zero pixel) + their shapes + the set of distinct pixel values
    tests/golden/reference_configs.npz    the parsed values: cms_camera / cms_orb_params bytes, fps, withFisheyeMask, RGB per file

DATA only: pixels and numbers.  No reference source text is stored.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

MASKS = {
    "lafida_450": "Masks/gray_lafida_cubemap_mask_450.png",
    "lafida_550": "Masks/gray_lafida_cubemap_mask_550.png",
    "lafida_650": "Masks/gray_lafida_cubemap_mask_650.png",
    "front_650": "Masks/gray_cubemap_front_mask_650.png",
    "left_650": "Masks/gray_cubemap_left_mask_650.png",
}
CONFIGS = {"lafida_cam0": "Config/lafida_cam0_params.yaml", "front_cam": "Config/front_cam_params.yaml", "left_cam": "Config/left_cam_params.yaml"}


def load_settings(path):
    """the product's reader of the reference's settings files -> (Camera struct, OrbParams struct, fps, withFisheyeMask, RGB)"""
    from cubemapslam_amd import api, build
    build.build(verbose=False)
    L = C.CDLL(build.HOST_LIB)
    L.hm_last_error.restype = C.c_char_p
    L.hm_settings_load.argtypes = [C.c_char_p] + [C.c_void_p] * 5
    cam = api.Camera(); orb = api.OrbParams(); fps = C.c_float(); wm = C.c_int(-1); rgb = C.c_int(-1)
    rc = L.hm_settings_load(path.encode(), C.byref(cam), C.byref(orb), C.byref(fps), C.byref(wm), C.byref(rgb))
    assert rc == 0, L.hm_last_error()
    return cam, orb, float(fps.value), int(wm.value), int(rgb.value)


def main():
    from PIL import Image
    from cubemapslam_amd import synth
    out_m, out_c = {}, {}
    for key, rel in MASKS.items():
        im = Image.open(os.path.join(REF, rel))
        a = np.asarray(im)
        assert a.ndim == 2 and a.dtype == np.uint8 and a.shape[0] == a.shape[1] and a.shape[0] % 3 == 0, (key, a.shape, a.dtype)
        out_m[key + "_bits"] = np.packbits(a != 0, axis=1)
        out_m[key + "_shape"] = np.array(a.shape, np.int32)
        out_m[key + "_values"] = np.unique(a)             # (the cull tests "== 0": only zero / non-zero matters; kept for the record)
        print("%-12s %4d x %4d, %5.1f %% non-zero, values %s" % (key, a.shape[1], a.shape[0], 100.0 * (a != 0).mean(), np.unique(a)[:8]))
    for key, rel in CONFIGS.items():
        cam, orb, fps, wm, rgb = load_settings(os.path.join(REF, rel))
        out_c[key + "_camera"] = np.frombuffer(bytes(cam), np.uint8).copy()
        out_c[key + "_orb"] = np.frombuffer(bytes(orb), np.uint8).copy()
        out_c[key + "_misc"] = np.array([fps, wm, rgb], np.float64)
        # the synthetic harness's constants are these files' values
        want = synth.LAFIDA if key == "lafida_cam0" else synth.FRONT if key == "front_cam" else None
        if want is not None:
            assert [cam.c, cam.d, cam.e, cam.u0, cam.v0] == [want[k] for k in ("c", "d", "e", "u0", "v0")], key
            assert list(cam.invpol)[:len(want["invpol"])] == list(want["invpol"]) and all(v == 0 for v in list(cam.invpol)[len(want["invpol"]):]), key
            assert list(cam.pol)[:len(want["pol"])] == list(want["pol"]), key
            assert (cam.Iw, cam.Ih, cam.fov_deg, orb.nfeatures) == (want["Iw"], want["Ih"], want["fov_deg"], want["nfeatures"]), key
        print("%-12s %d x %d, face %d, nFeatures %d, fov %.0f, fps %.0f, withFisheyeMask %d" % (key, cam.Iw, cam.Ih, cam.face, orb.nfeatures, cam.fov_deg, fps, wm))
    np.savez_compressed(os.path.join(HERE, "reference_masks.npz"), **out_m)
    np.savez_compressed(os.path.join(HERE, "reference_configs.npz"), **out_c)
    for f in ("reference_masks.npz", "reference_configs.npz"):
        print("wrote tests/golden/%s (%d bytes)" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
