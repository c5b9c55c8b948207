"""Developer aid: latency of ONE local-BA call as the reference issues it (create + optimize + read + destroy), small and large windows."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
for name, prob in (("closed-loop size (K=6, E~2k)", synth.ba_problem(K=6, P=550, obs_per_point=4, F=550, seed=3)),
                   ("configs[3] size (K=20, E~80k)", synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42))):
    ts = {"create": [], "optimize": [], "again after reset": [], "read+destroy": []}
    for rep in range(7):
        t0 = time.perf_counter(); ba = api.BundleAdjuster(prob); t1 = time.perf_counter()
        ba.optimize((5, 10)); t2 = time.perf_counter()
        ba.reset(); ta = time.perf_counter(); ba.optimize((5, 10)); tb = time.perf_counter()
        ba.read(); ba.close(); t3 = time.perf_counter()
        if rep >= 2:
            ts["create"].append(t1 - t0); ts["optimize"].append(t2 - t1); ts["again after reset"].append(tb - ta); ts["read+destroy"].append(t3 - tb)
    print("%-32s E %6d: " % (name, len(prob["e_pose"])) + "  ".join("%s %.2f ms" % (k, 1e3 * np.median(v)) for k, v in ts.items()))
