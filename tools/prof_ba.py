import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import api, synth
prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42)
ba = api.BundleAdjuster(prob)
for i in range(3):
    ba.reset(); t = time.perf_counter(); rc, st = ba.optimize((5, 10)); dt = time.perf_counter() - t
    print("BA window %.2f ms its %s" % (dt * 1e3, list(st.iterations_done)))
import ctypes, numpy as np
clk = (ctypes.c_longlong * 16)()
api.lib().cms_ba_debug_clocks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
api.lib().cms_ba_debug_clocks(ba.h, clk)
c = np.array(list(clk)[:5], dtype=np.int64)
c8 = np.array(list(clk), dtype=np.int64)
print("column 0 (us): diag0+panel %.2f  trailing(1,1) %.2f  factor(1,1) %.2f" % ((c8[5] - c8[1]) / 100.0, (c8[6] - c8[5]) / 100.0, (c8[7] - c8[6]) / 100.0))
print("trial_solve phases (us): assemble %.1f factor %.1f backsub %.1f poses+tail %.1f total %.1f" % tuple(list(np.diff(c) / 100.0) + [(c[4] - c[0]) / 100.0]))
for n in (4, 8, 16):
    bas = [api.BundleAdjuster(prob) for _ in range(n)]
    for i in range(3):
        for b in bas:
            b.reset()
        t = time.perf_counter(); api.ba_optimize_many(bas, (5, 10)); dt = time.perf_counter() - t
    print("%d windows lock-step: %.2f ms (%.2f ms / window)" % (n, dt * 1e3, dt * 1e3 / n))
    del bas
