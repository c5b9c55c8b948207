"""Developer aid (needs the -DBA_RM_CLK=<unit> build: tools/ab_build.sh rwclk -DBA_RM_CLK=7, CMS_HIP_LIB=.../ab_rwclk.so): cycle stamps of one
unit (one wavefront) of the one-wavefront run workgroups (cms_ba_schur_runwg.hip, class 0), phase by phase."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i, views="track") for i in range(n)]
bas = [api.BundleAdjuster(p) for p in probs]
api.ba_optimize_many(bas, (5, 10))
for b in bas:
    b.reset()
api.ba_optimize_many(bas, (1, 0))
out = np.zeros(16, np.int64)
rc = api.lib().cms_ba_debug_rm_clocks(api._p(out))
names = ["loop", "operands ready", "residual+jacobian", "own sums via buffer", "row exchange", "factor+W", "published+next loads", "matrix phase", "flush", "unit start"]
nch = max(int(out[12]), 1)
print("rc", rc, "chunks of the unit", nch, "cycles: total %.0f, per chunk %.0f" % (out[:10].sum(), out[:9].sum() / nch))
for i, nm in enumerate(names):
    print("  %-22s %8.0f per chunk   (%d in all)" % (nm, out[i] / nch, out[i]))
