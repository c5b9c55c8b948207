import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from cubemapslam_amd import api, synth
import orc
prob = synth.ba_problem(K=20, P=1105, obs_per_point=4, F=550, seed=42)
print("edges", len(prob["e_pose"]))
ba = api.BundleAdjuster(prob)
for i in range(3):
    ba.reset(); t = time.perf_counter(); rc, st = ba.optimize((5, 10)); dt = time.perf_counter() - t
print("GPU 1 window %.2f ms" % (dt * 1e3), list(st.iterations_done))
bas = [api.BundleAdjuster(prob) for _ in range(8)]
for i in range(3):
    for b in bas: b.reset()
    t = time.perf_counter(); api.ba_optimize_many(bas, (5, 10)); dt = time.perf_counter() - t
print("GPU 8 windows %.2f ms" % (dt * 1e3))
t = time.perf_counter(); orc.ba_run(prob); print("CPU oracle %.2f ms" % ((time.perf_counter() - t) * 1e3))
