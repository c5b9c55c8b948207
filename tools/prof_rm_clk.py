"""Developer aid (needs the -DBA_RM_CLK build: tools/ab_build.sh rmclk -DBA_RM_CLK, CMS_HIP_LIB=.../ab_rmclk.so): cycle stamps of one
wavefront of the run-major MFMA body, phase by phase."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i, views="track") for i in range(n)]
bas = [api.BundleAdjuster(p) for p in probs]
api.ba_optimize_many(bas, (5, 10))
for b in bas:
    b.reset()
api.ba_optimize_many(bas, (1, 0))          # a single round (plus idle ones): the stamps are those of the last launch that had work
out = np.zeros(16, np.int64)
rc = api.lib().cms_ba_debug_rm_clocks(api._p(out))
names = ["loop", "operands ready", "residual+jacobian", "row exchange", "factor+W+own sums", "rows published", "next loads issued", "matrix phase", "flush"]
nch = max(int(out[12]), 1)
print("rc", rc, "chunks of the wavefront", nch, "cycles per chunk: total %.0f" % (out[:9].sum() / nch))
for i, nm in enumerate(names):
    print("  %-22s %8.0f" % (nm, out[i] / nch))
