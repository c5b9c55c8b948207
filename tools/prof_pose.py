"""Developer aid: time Optimizer::PoseOptimization on the GPU for a batch of frames (one workgroup per frame, one launch)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from cubemapslam_amd import api, synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
probs = [synth.pose_problem(N=N, seed=100 + f, outlier_frac=0.1) for f in range(nf)]
po = api.PoseOptimizer(nf, sum(len(p["Xw"]) for p in probs))
po.upload(probs)
for it in range(3):
    t = time.perf_counter(); po.launch(); r = po.fetch(); dt = time.perf_counter() - t
print("%d frames x %d edges: %.3f ms per launch (%.1f us / frame), inliers %s" % (nf, N, dt * 1e3, dt * 1e6 / nf, r[0][:4]))
po1 = api.PoseOptimizer(1, N); po1.upload(probs[:1])
for it in range(3):
    t = time.perf_counter(); po1.launch(); po1.fetch(); dt = time.perf_counter() - t
print("1 frame: %.3f ms" % (dt * 1e3))
import orc
t = time.perf_counter()
for p in probs[:8]:
    orc.pose_optimize(p)
print("CPU oracle: %.3f ms / frame" % ((time.perf_counter() - t) * 1e3 / 8))
