"""Developer aid: n windows of BASELINE configs[3] in lock-step (for rocprofv3 --kernel-trace --stats).

    python tools/prof_ba_many.py [n windows] [track|random] [same|diff] [kernel of the round to time: 3 Schur (default), 5 solve, 6 trial, 7 reduce2]
"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cubemapslam_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
views = sys.argv[2] if len(sys.argv) > 2 else "track"
diff = len(sys.argv) > 3 and sys.argv[3] == "diff"
kid = int(sys.argv[4]) if len(sys.argv) > 4 else 3
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + (i if diff else 0), views=views) for i in range(n if diff else 1)]
probs = probs if diff else probs * n
pl = api.ba_plan(probs[0]["fixed"], len(probs[0]["points"]), probs[0]["e_pose"], probs[0]["e_point"])
print("views %s: E %d, %d signature runs hold %.0f %% of the points, %d of %d chunks are run chunks" % (
    views, len(probs[0]["e_pose"]), pl["n_runs"], 100.0 * pl["rm_points"] / len(probs[0]["points"]), pl["n_rm"], pl["n_chunks"]))
for rep in range(2):          # the second set of windows reuses the first one's streams
    t = time.perf_counter()
    bas = [api.BundleAdjuster(p) for p in probs]
    print("cms_ba_create: %.2f ms per window (host work list + uploads%s)" % ((time.perf_counter() - t) * 1e3 / n, ", first windows of the process" if rep == 0 else ""))
    if rep == 0:
        for b in bas:
            b.close()
ts = []
for i in range(5):
    for b in bas:
        b.reset()
    bas[0].profile_kernel(kid)
    t = time.perf_counter(); _, stats = api.ba_optimize_many(bas, (5, 10)); dt = time.perf_counter() - t
    ms, nl = bas[0].profile_get()
    ts.append(dt)
print("%d windows lock-step: %.2f ms (best of 5: %.2f), iterations %s; %s kernel %.1f us average over %d rounds" % (
    n, dt * 1e3, min(ts) * 1e3, list(stats[0].iterations_done), {3: "Schur", 5: "solve", 6: "trial", 7: "reduce2"}.get(kid, str(kid)), 1e3 * ms / max(nl, 1), nl))
t = time.perf_counter()
outs = [b.read() for b in bas]
print("cms_ba_read: %.3f ms per window" % ((time.perf_counter() - t) * 1e3 / n))
