"""Developer aid: n windows of BASELINE configs[3] in lock-step (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42)
bas = [api.BundleAdjuster(prob) for _ in range(n)]
for i in range(4):
    for b in bas:
        b.reset()
    t = time.perf_counter(); api.ba_optimize_many(bas, (5, 10)); dt = time.perf_counter() - t
print("%d windows lock-step: %.2f ms" % (n, dt * 1e3))
