"""LocalMapping::CreateNewMapPoints timing: python tools/prof_tri.py [jobs] [neighbours] [reps]
GPU: whole call (flatten + upload + kernel + fetch); CPU: the oracle on the same jobs, one thread."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
torch.cuda.init()
from cubemapslam_amd import api, synth
import orc

J = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NN = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
F = 550
camd = synth.camera("lafida", F)
ocam = orc.make_camera(camd)
ctx = api.Context(camd, nfeatures=2000, max_batch=1)
jobs_g, jobs_o, keep = [], [], []
for j in range(J):
    S = synth.keyframe_set(F, n_kf=NN + 1, n_pts=2400, seed=200 + j)
    oks = [orc.make_keyframe(ocam, k) for k in S["kfs"]]
    gks = [api.make_keyframe(k) for k in S["kfs"]]
    keep.append((S, oks, gks))
    jobs_g.append((gks[0][0], [k for k, _ in gks[1:]]))
    jobs_o.append((S, oks))
got = api.create_new_map_points(ctx, jobs_g)
t0 = time.perf_counter()
for _ in range(reps):
    got = api.create_new_map_points(ctx, jobs_g)
t_gpu = (time.perf_counter() - t0) / reps
store = api.KeyframeStore(ctx, max_keyframes=J * (NN + 1), max_features=2048, max_nodes=1024)
slot_jobs = []
for j, (S, oks, gks) in enumerate(keep):
    base = j * (NN + 1)
    for i, (K, _) in enumerate(gks):
        store.put(base + i, K)
    slot_jobs.append((base, list(range(base + 1, base + NN + 1))))
res = store.create_new_map_points(slot_jobs)
assert all(np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) for a, b in zip(res, got))
t0 = time.perf_counter()
for _ in range(reps):
    res = store.create_new_map_points(slot_jobs)
t_res = (time.perf_counter() - t0) / reps
print("resident key frames: %.3f ms per call (%.3f ms/job)" % (1e3 * t_res, 1e3 * t_res / J))
t0 = time.perf_counter()
tot = 0
for S, oks in jobs_o:
    cur_mp = S["kfs"][0]["mp"].copy()
    w = orc.create_new_map_points(ocam, oks[0][0], [k for k, _ in oks[1:]], S["scale_factors"], S["level_sigma2"], cur_mp)
    tot += len(w[0])
t_cpu = time.perf_counter() - t0
same = all(np.array_equal(g[1], None) is False for g in got)
print("jobs %d x %d neighbours, %d features/KF: GPU call %.3f ms (%.3f ms/job), CPU oracle %.3f ms (%.3f ms/job), new points/job %.0f" %
      (J, NN, jobs_g[0][0].n, 1e3 * t_gpu, 1e3 * t_gpu / J, 1e3 * t_cpu, 1e3 * t_cpu / J, tot / J))
