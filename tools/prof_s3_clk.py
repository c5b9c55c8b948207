"""Developer aid (needs the -DBA_S3_CLK build: tools/ab_build.sh s3clk -DBA_S3_CLK, CMS_HIP_LIB=.../ab_s3clk.so): cycle stamps of thread 0 of
kb_ba_trial_solve3r (window 0), phase by phase, of the last launch that had work."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i, views="track") for i in range(n)]
bas = [api.BundleAdjuster(p) for p in probs]
api.ba_optimize_many(bas, (5, 10))
for b in bas:
    b.reset()
api.ba_optimize_many(bas, (1, 0))
out = np.zeros(16, np.int64)
rc = api.lib().cms_ba_debug_s3_clocks(api._p(out))
names = ["assembly + first diagonal block", "-", "panels (all steps)", "trailing updates + next diagonal block (all steps)", "-", "back substitution", "pose update"]
print("rc", rc, "free key frames", int(out[10]), "cycles (s_memtime, 100 MHz constant clock x ?): total %d" % out[:7].sum())
for i, nm in enumerate(names):
    print("  %-34s %8d" % (nm, out[i]))
for b in bas:
    b.reset()
bas[0].profile_kernel(5)
api.ba_optimize_many(bas, (5, 10))
ms, nl = bas[0].profile_get()
print("solve kernel %.1f us average over %d rounds" % (1e3 * ms / max(nl, 1), nl))
