"""Developer aid (needs tools/ab_build.sh rwts -DBA_RW_TS, CMS_HIP_LIB=.../ab_rwts.so): when the units of the one-wavefront run workgroups start and end."""
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
api.ba_optimize_many(bas, (1, 0))
out = np.zeros(2 * 4096 * 2, np.int64)
lib = api.lib()
lib.cms_ba_debug_rw_ts.argtypes = [C.c_void_p]
rc = lib.cms_ba_debug_rw_ts(out.ctypes.data)
out = out.reshape(2, 4096, 2)
for cls in range(2):
    t = out[cls]; t = t[t[:, 0] > 0]
    if len(t) == 0: continue
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0; du = en - st
    q = lambda v: " ".join("%6.1f" % np.percentile(v, p) for p in (0, 10, 50, 90, 99, 100))
    print("class %d: %d units (us; percentiles 0 10 50 90 99 100)\n  start    %s\n  end      %s\n  duration %s" % (cls, len(t), q(st), q(en), q(du)))
# per unit of a window (all windows hold the same problem with "same"): mean duration against the unit's chunks
if len(sys.argv) > 2 and sys.argv[2] == "same":
    pass
U0 = len(out[0][out[0][:, 0] > 0]) // n
d0 = (out[0][:U0 * n, 1] - out[0][:U0 * n, 0]).reshape(n, U0) / 100.0
pl = api.ba_plan(probs[0]["fixed"], len(probs[0]["points"]), probs[0]["e_pose"], probs[0]["e_point"], tables=True)
rmc = pl["rm_chunk"]; kf_run = pl["run_mf"].reshape(-1, 64)[:, 56]
print("window 0, class 0: unit, us, then (k_run, kf, points) of the unit's chunks [cut by the library's cost model, replayed here]")
def cost(k, kf, m): return 45 + (1 if kf <= 2 else 3 if kf <= 5 else 6) * 3 * ((m + 3) >> 2) + (0 if k in (2, 4) else 5)
cs = [cost((int(w) >> 8) & 255, int(kf_run[r]), int(w) >> 16) for (_, w, r, _) in rmc]
nA = sum(1 for (_, w, r, _) in rmc if 6 * int(kf_run[r]) + 1 <= 32)
cum = np.concatenate([[0], np.cumsum(cs[:nA])])
def cut(i):
    target = (cum[-1] * i + 1023) // 1024
    return int(np.searchsorted(cum, target, side="left"))
for u in list(range(0, U0, max(1, U0 // 24))):
    cb, ce = cut(u * 1024 // U0), cut((u + 1) * 1024 // U0)
    print("  %4d %6.1f  %s" % (u, d0[0, u], " ".join("(%d,%d,%d)" % ((int(rmc[c][1]) >> 8) & 255, int(kf_run[int(rmc[c][2])]), int(rmc[c][1]) >> 16) for c in range(cb, ce))))
