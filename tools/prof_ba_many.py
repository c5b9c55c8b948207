"""Developer aid: n windows of BASELINE configs[3] in lock-step (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42)
for rep in range(2):          # the second set of windows reuses the first one's streams
    t = time.perf_counter()
    bas = [api.BundleAdjuster(prob) for _ in range(n)]
    print("cms_ba_create: %.2f ms per window (host work list + uploads%s)" % ((time.perf_counter() - t) * 1e3 / n, ", first windows of the process" if rep == 0 else ""))
    if rep == 0:
        for b in bas:
            b.close()
for i in range(4):
    for b in bas:
        b.reset()
    t = time.perf_counter(); api.ba_optimize_many(bas, (5, 10)); dt = time.perf_counter() - t
print("%d windows lock-step: %.2f ms" % (n, dt * 1e3))
