import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import api, synth
prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42)
ba = api.BundleAdjuster(prob)
for i in range(3):
    ba.reset(); t = time.perf_counter(); rc, st = ba.optimize((5, 10)); dt = time.perf_counter() - t
    print("BA window %.2f ms its %s" % (dt * 1e3, list(st.iterations_done)))
