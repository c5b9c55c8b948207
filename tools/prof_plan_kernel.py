"""developer helper (GPU box): the phases of k_ba_plan_many for bench-sized windows (CMS_BA_DP_CLK=1 prints them to stderr)"""
import os, sys, time
os.environ["CMS_BA_DP_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubemapslam_amd import api, synth
probs = [dict(synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i, views="track"), _plan_on_device=True) for i in range(8)]
for rep in range(3):
    t0 = time.perf_counter()
    many = api.ba_create_many(probs, threads=2)
    print("create_many of 8: %.2f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
    for b in many:
        b.close()
