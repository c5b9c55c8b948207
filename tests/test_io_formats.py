"""The reference's file formats around the hot path, host only: settings YAML, image lists, tracking-time summary."""
import ctypes as C
import numpy as np
from cubemapslam_amd import api, build, synth


def _host():
    build.build(verbose=False)
    L = C.CDLL(build.HOST_LIB)
    L.hm_last_error.restype = C.c_char_p
    L.hm_settings_load.argtypes = [C.c_char_p] + [C.c_void_p] * 5
    L.hm_load_image_list.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.hm_write_tracking_summary.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def _yaml(cam, nfeat=2000, fps=None, extra=""):
    lines = ["%YAML:1.0", "# camera", "Camera.Iw: %d" % cam["Iw"], "Camera.Ih: %d   # rows" % cam["Ih"], "", "Camera.nrpol: 5", "Camera.nrinvpol: 12"]
    lines += ["Camera.a%d: %r" % (i, float(v)) for i, v in enumerate(cam["pol"])]
    lines += ["Camera.pol%d: %r" % (i, float(v)) for i, v in enumerate(cam["invpol"])]
    lines += ["Camera.%s: %r" % (k, float(cam[k])) for k in ("c", "d", "e", "u0", "v0")]
    lines += ["Camera.fov: %r" % float(cam["fov_deg"]), "Camera.RGB: 1", "Camera.withFisheyeMask: 0", "CubeFace.w: %d" % cam["face"], "CubeFace.h: %d" % cam["face"],
              "ORBextractor.nFeatures: %d" % nfeat, "ORBextractor.scaleFactor: 1.2", "ORBextractor.nLevels: 8", "ORBextractor.iniThFAST: 20",
              "ORBextractor.minThFAST: 7", 'Viewer.Title: "a: b # c"']
    if fps is not None:
        lines.append("Camera.fps: %r" % fps)
    return "\n".join(lines) + "\n" + extra


def test_settings_yaml_round_trip(tmp_path):
    L = _host()
    camd = synth.camera("lafida", 550)
    p = tmp_path / "cam.yaml"
    p.write_text(_yaml(camd, fps=0.0))
    cam = api.Camera(); orb = api.OrbParams(); fps = C.c_float(); wm = C.c_int(-1); rgb = C.c_int(-1)
    assert L.hm_settings_load(str(p).encode(), C.byref(cam), C.byref(orb), C.byref(fps), C.byref(wm), C.byref(rgb)) == 0, L.hm_last_error()
    want = api.make_camera(camd)
    assert bytes(cam) == bytes(want)                               # every double identical: strtod round trip of repr()
    assert (orb.nfeatures, orb.nlevels, orb.ini_th_fast, orb.min_th_fast) == (2000, 8, 20, 7) and abs(orb.scale_factor - 1.2) < 1e-7
    assert fps.value == 30.0 and wm.value == 0 and rgb.value == 1  # fps 0 -> 30 (Tracking.cpp:66-68)
    # shorter polynomials are zero padded (System.cpp:67-72); a missing key reads as 0
    short = _yaml(camd).replace("Camera.nrinvpol: 12", "Camera.nrinvpol: 9").replace("Camera.nrpol: 5", "Camera.nrpol: 3")
    p.write_text(short.replace("Camera.fov", "Camera.nofov"))
    assert L.hm_settings_load(str(p).encode(), C.byref(cam), None, C.byref(fps), None, None) == 0
    assert list(cam.invpol[9:]) == [0.0] * 3 and list(cam.pol[3:]) == [0.0] * 2 and cam.invpol[8] == want.invpol[8] and cam.fov_deg == 0.0
    assert fps.value == 30.0
    assert L.hm_settings_load(str(tmp_path / "missing.yaml").encode(), C.byref(cam), None, None, None, None) != 0
    assert b"Failed to open settings file" in L.hm_last_error()


def test_image_lists(tmp_path):
    L = _host()
    laf = tmp_path / "images.lst"
    laf.write_text("1409666701.100000 imgs/cam0/00000001.png\n1409666701.150000 00000002.png\n1409666701.25 /abs/dir/x_3.png\n")
    names = np.zeros((8, 64), np.uint8); ts = np.zeros(8, np.float64)
    n = L.hm_load_image_list(str(laf).encode(), 0, 8, 64, names.ctypes.data, ts.ctypes.data)
    got = [bytes(names[i]).split(b"\0")[0].decode() for i in range(n)]
    assert n == 3 and got == ["00000001.png", "00000002.png", "x_3.png"]
    assert list(ts[:3]) == [1409666701.1, 1409666701.15, 1409666701.25]
    fan = tmp_path / "fangshan.lst"
    fan.write_text("1500000000.125_front.jpg\n1500000000.250_front.jpg\n")
    n = L.hm_load_image_list(str(fan).encode(), 1, 8, 64, names.ctypes.data, ts.ctypes.data)
    got = [bytes(names[i]).split(b"\0")[0].decode() for i in range(n)]
    assert n == 2 and got == ["1500000000.125_front.jpg", "1500000000.250_front.jpg"] and list(ts[:2]) == [1500000000.125, 1500000000.25]


def test_tracking_summary(tmp_path):
    L = _host()
    t = np.array([0.031, 0.012, 0.044, 0.020, 0.027, 0.019], np.float32)
    buf = C.create_string_buffer(512)
    out = tmp_path / "perf.txt"
    times = t.copy()
    assert L.hm_write_tracking_summary(str(out).encode(), times.ctypes.data, len(t), 5, buf, 512) == 0
    s = np.sort(t)
    assert np.array_equal(times, s)                                # sorted in place like the reference
    tot = np.float32(0)
    for v in s:
        tot = np.float32(tot + v)
    median, mean = s[len(s) // 2], np.float32(tot / len(s))
    # the file is written with std::fixed (6 decimals), the console text with the default format
    assert out.read_text() == ("-------\n\nmedian tracking time: %.6f\nmean tracking time: %.6f\ntracking frames/ total frames: 5/ 6 %.6f\n"
                               % (median, mean, np.float32(5) / np.float32(6)))
    assert buf.value.decode() == "-------\n\nmedian tracking time: %g\nmean tracking time: %g\n" % (median, mean)
