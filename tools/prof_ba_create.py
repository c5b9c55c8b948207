"""developer helper (GPU box): where the host side of cms_ba_create goes, phase by phase (CMS_BA_CREATE_TIMING), for a tracked configs[3] window.
usage: CMS_BA_CREATE_TIMING=1 python tools/prof_ba_create.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMS_BA_CREATE_TIMING", "1")
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=7 + i, views="track") for i in range(2)]
for i in range(n):
    t = time.perf_counter()
    ba = api.BundleAdjuster(probs[i & 1])
    dt = 1e3 * (time.perf_counter() - t)
    ba.close()
    print("window %d: %.2f ms" % (i, dt), file=sys.stderr)
