"""Per-frame latency of the tracking chain for ONE camera stream through the host-buffer entries (what the reference's Tracking thread
would call per frame): remap + extract, frame grid, SearchByProjection(Cur, Last), PoseOptimization, SearchLocalPoints, PoseOptimization.
python tools/prof_track_latency.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
torch.cuda.init()
from cubemapslam_amd import api, synth
import orc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = 550
camd = synth.camera("lafida", F)
ctx = api.Context(camd, nfeatures=camd["nfeatures"], max_batch=1)
ctx.set_mask(synth.cubemap_valid_mask(camd))
fish = synth.texture(camd["Ih"], camd["Iw"], 3)
k, d = ctx.remap_extract(fish)
ctx.area_grid(1)
mm = synth.motion_model_problem(F, k["x"], k["y"], k["octave"], k["angle"], d, seed=1)
lm = synth.local_map_problem(F, k["x"], k["y"], k["octave"], d, seed=2)
pp = synth.pose_problem(N=600, F=F, seed=3)
po = api.PoseOptimizer(1, len(pp["Xw"]))
t = {n: 0.0 for n in ("extract", "grid", "search_by_projection", "pose_opt", "search_local", "total")}
for r in range(reps + 2):
    t0 = time.perf_counter()
    k, d = ctx.remap_extract(fish)
    t1 = time.perf_counter()
    ctx.area_grid(1); ctx.sync()
    t2 = time.perf_counter()
    kp = np.full(len(k), -1, np.int32)
    m, nm = ctx.search_by_projection(0, mm["pose12"], mm["valid"], mm["Xw"], mm["octave"], mm["angle"], mm["desc"], kp)
    t3 = time.perf_counter()
    po.optimize([pp])
    t4 = time.perf_counter()
    res = ctx.search_local_points(0, lm["pose15"], lm["pos"], lm["normal"], lm["min_dist"], lm["max_dist"], lm["desc"], kp)
    t5 = time.perf_counter()
    po.optimize([pp])
    t6 = time.perf_counter()
    if r >= 2:
        t["extract"] += t1 - t0; t["grid"] += t2 - t1; t["search_by_projection"] += t3 - t2; t["pose_opt"] += (t4 - t3) + (t6 - t5)
        t["search_local"] += t5 - t4; t["total"] += t6 - t0
print("one stream, host-buffer entries, ms per frame:", {n: round(1e3 * v / reps, 3) for n, v in t.items()}, "| %d key points, %d + %d matches" % (len(k), nm, res["n_matches"]))
# the same chain on the CPU oracle, one thread
ocam = orc.make_camera(camd)
m1, m2 = orc.build_lut(ocam)
o = orc.Orb(nfeatures=camd["nfeatures"])
mask = synth.cubemap_valid_mask(camd)
t0 = time.perf_counter()
cube = orc.fisheye_to_cubemap(ocam, m1, m2, fish)
kc, dc = o.extract(ocam, cube, mask)
t1 = time.perf_counter()
kp = np.full(len(kc), -1, np.int32)
orc.search_by_projection_frames(ocam, mm["pose12"][:9], mm["pose12"][9:], kc["x"], kc["y"], kc["octave"], kc["angle"], dc, mm["scale_factors"], mm["valid"], mm["Xw"],
                                mm["octave"], mm["angle"], mm["desc"], kp)
orc.pose_optimize(pp)
fr = orc.is_in_frustum(ocam, lm["pose15"], lm["pos"], lm["normal"], lm["min_dist"], lm["max_dist"])
orc.search_local_points(ocam, kc["x"], kc["y"], kc["octave"], dc, lm["scale_factors"], fr, lm["desc"], kp)
orc.pose_optimize(pp)
t2 = time.perf_counter()
print("CPU oracle, one thread: extract %.1f ms, tracking searches + 2 pose optimisations %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
