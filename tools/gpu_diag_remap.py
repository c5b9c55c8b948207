import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, orc
from cubemapslam_amd import api, synth
for name, F, Ih, seed in (("front", 650, 1024, 21), ("lafida", 450, None, 1)):
    camd = synth.camera(name, F, Ih=Ih); ocam = orc.make_camera(camd)
    ctx = api.Context(camd, nfeatures=500)
    m1, m2 = orc.build_lut(ocam)
    fish = synth.texture(camd["Ih"], camd["Iw"], seed)
    for rep in range(2):
        got = ctx.remap(fish)
        ref = orc.fisheye_to_cubemap(ocam, m1, m2, fish)
        ys, xs = np.nonzero(got != ref)
        print(name, "rep", rep, "ndiff", len(ys))
        lut = ctx.debug_lut()
        for y, x in list(zip(ys, xs))[:25]:
            e = int(lut[y, x]); X = e & 0x7FF; Y = (e >> 11) & 0x7FF; ax = (e >> 22) & 31; ay = e >> 27
            nb = fish[Y:Y+2, X:X+2].tolist()
            print("  (y=%d,x=%d) got %d want %d  X=%d Y=%d ax=%d ay=%d nb=%s m=(%r,%r)" % (y, x, got[y, x], ref[y, x], X, Y, ax, ay, nb, float(m1[y,x]), float(m2[y,x])))
    ctx.close()
