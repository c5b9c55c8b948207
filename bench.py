#!/usr/bin/env python3
"""bench.py -- frames/sec of the CubemapSLAM hot path (remap + ORB extract + Hamming match + local BA) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json: "frames/sec (extract+match+localBA) on Lafida cam0", configs[2] geometry): synthetic Lafida cam0
stream, 754x480 fisheye, cube face F=550 (1650^2 cross), nFeatures 2000 / 8 levels / 1.2 / FAST 20-7.
One step = one batch of B frames (default 256 = 32 camera streams x 8 consecutive frames, cf. BASELINE.json configs[4]), inputs
resident in HBM:
    remap -> pyramid -> FAST cells -> octree -> cull -> orientation + rBRIEF     (ORBextractor::operator(), all B frames per launch)
    Frame::AssignFeaturesToGrid + GetFeaturesInArea windows + Hamming best/second-best of every key point of frame b-1 in
        frame b (ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th=15)), all on the device, every step
    Optimizer::PoseOptimization of every frame (one launch for the batch, own stream)
    one local BA window (K=20 key frames, ~80k cubemap edges, BASELINE.json configs[3]) per `--ba-every` frames, on its own
        stream / host thread like the reference's LocalMapping thread.
Weak scaling over GPUs: every rank runs its own stream(s); the only exchange is an RCCL gather of the per-frame trajectory
records [ts, t(3), q(4)] (System.cpp:261-262 order) to rank 0.

Prints ONE JSON line (rank 0) incl. `roofline` for the dominant kernel (HIP-event timing on the library's stream) and
`cpu_baseline` (the CPU oracle timed on this box's host cores, single thread, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

# The frame path and every local-BA window run on their own HIP stream; the ROCm runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), which serialises the BA windows behind each other.  Must be set before the
# HIP runtime initialises (measured: 4 -> 8 queues = +28 % frames/s with 4 concurrent windows).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_frames(camd, B, seed):
    """B consecutive frames of one synthetic stream: a large seeded texture drifting 3 px / frame under the fisheye."""
    from cubemapslam_amd import synth
    Ih, Iw = camd["Ih"], camd["Iw"]
    big = synth.texture(Ih + 4 * B + 8, Iw + 4 * B + 8, seed)
    return np.stack([big[2 * b:2 * b + Ih, 3 * b:3 * b + Iw] for b in range(B)]).copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (default: 32 streams x 8 consecutive frames)")
    ap.add_argument("--face", type=int, default=550)
    ap.add_argument("--ba-every", type=int, default=8, help="one local-BA window per this many frames")
    ap.add_argument("--ba-groups", type=int, default=4, help="host threads / streams the local-BA windows of a step are split over")
    ap.add_argument("--pose-edges", type=int, default=600, help="matched map points per frame for the pose-only optimisation")
    ap.add_argument("--save-trajectory", default="", help="rank 0 writes the gathered trajectory of the last step here (TUM format)")
    ap.add_argument("--force-gather", action="store_true", help="run the trajectory gather code path even with one rank (self-test)")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the CPU-oracle baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from cubemapslam_amd import api, build, synth
    from cubemapslam_amd import dist as cdist
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()

    B, F = args.batch, args.face
    camd = synth.camera("lafida", F)
    nfeat = camd["nfeatures"]
    ctx = api.Context(camd, nfeatures=nfeat, max_batch=B, device=local_rank)
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    frames = make_frames(camd, B, seed=100 + rank)
    ctx.upload(frames)          # inputs resident in HBM before the timed region
    ctx.process(B, True)
    ctx.sync()
    fetched = [ctx.fetch(b) for b in range(B)]
    kps = [f[0] for f in fetched]
    g = ctx.geom
    kp_cap = g.kp_cap
    scales = [g.scale[l] for l in range(g.nlevels)]
    dev = torch.device("cuda", local_rank)
    # ---- frame-to-frame matching (ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th = 15), Tracking::TrackWithMotionModel): per
    # frame a predicted pose and the ~1400 map points its last frame holds; projection, windows, greedy best match and the rotation
    # histogram run on the device every step (the key points the windows are answered from are the ones this step extracts)
    ctx.area_grid(B)
    mm = [synth.motion_model_problem(F, k["x"], k["y"], k["octave"], k["angle"], d, seed=5000 + 100 * rank + b) for b, (k, d) in enumerate(fetched)]
    mm_off = np.concatenate([[0], np.cumsum([len(p["valid"]) for p in mm])]).astype(np.int32)
    nq = int(mm_off[-1])
    mcat = lambda key, dt: torch.from_numpy(np.concatenate([p[key] for p in mm]).astype(dt)).to(dev)
    d_mm_pose = torch.from_numpy(np.stack([p["pose12"] for p in mm])).to(dev)
    d_mm_frame = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), np.diff(mm_off))).to(dev)
    d_mm_valid, d_mm_xw, d_mm_oct, d_mm_ang, d_mm_desc = mcat("valid", np.uint8), mcat("Xw", np.float32), mcat("octave", np.int32), mcat("angle", np.float32), mcat("desc", np.uint8)
    d_mm_q = [torch.zeros(nq, dtype=torch.float32, device=dev) for _ in range(3)] + [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(2)]
    d_cnt = torch.zeros(max(nq, 1), dtype=torch.int32, device=dev); d_off = torch.zeros(nq + 1, dtype=torch.int32, device=dev)
    d_tot = torch.zeros(1, dtype=torch.int32, device=dev)
    d_mm_match = torch.zeros(max(nq, 1), dtype=torch.int32, device=dev); d_mm_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_mm_mpoff = torch.from_numpy(mm_off).to(dev)

    def mm_windows(d_idx_buf, cap):
        ctx.project_last_frame_device(nq, d_mm_frame.data_ptr(), d_mm_pose.data_ptr(), d_mm_valid.data_ptr(), d_mm_xw.data_ptr(), d_mm_oct.data_ptr(), 15.0,
                                      [t.data_ptr() for t in d_mm_q])
        ctx.features_in_area_batch_device(nq, d_mm_frame.data_ptr(), [t.data_ptr() for t in d_mm_q], d_cnt.data_ptr(), d_off.data_ptr(),
                                          d_idx_buf.data_ptr(), cap, d_tot.data_ptr())

    mm_windows(torch.zeros(1, dtype=torch.int32, device=dev), 0)       # dry run: size the candidate buffer
    ctx.sync()
    n_pairs = int(d_tot.item())
    cand_cap = n_pairs + 4096
    d_idx = torch.zeros(cand_cap, dtype=torch.int32, device=dev); d_mm_pd = torch.zeros(cand_cap, dtype=torch.int16, device=dev)

    # ---- track local map (Tracking::SearchLocalPoints): every frame has its own pose and local map (~2000 points, about two thirds in
    # view, a quarter competing for a key point); projection, windows and the greedy search run on the device every step
    lm = [synth.local_map_problem(F, k["x"], k["y"], k["octave"], d, seed=7000 + 100 * rank + b) for b, (k, d) in enumerate(fetched)]
    lm_off = np.concatenate([[0], np.cumsum([len(p["pos"]) for p in lm])]).astype(np.int32)
    n_mp = int(lm_off[-1])
    lcat = lambda key, dt: torch.from_numpy(np.concatenate([p[key] for p in lm]).astype(dt)).to(dev)
    d_lm_pose = torch.from_numpy(np.stack([p["pose15"] for p in lm])).to(dev)
    d_lm_frame = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), np.diff(lm_off))).to(dev)
    d_lm_in = [lcat("pos", np.float32), lcat("normal", np.float32), lcat("min_dist", np.float32), lcat("max_dist", np.float32)]
    d_lm_desc = lcat("desc", np.uint8)
    d_lm_vis = torch.zeros(n_mp, dtype=torch.uint8, device=dev)
    d_lm_f = [torch.zeros(n_mp, dtype=torch.float32, device=dev) for _ in range(4)]          # proj_x, proj_y, view_cos, qr
    d_lm_i = [torch.zeros(n_mp, dtype=torch.int32, device=dev) for _ in range(5)]            # level, qmin, qmax, cnt, match
    d_lm_off = torch.zeros(n_mp + 1, dtype=torch.int32, device=dev); d_lm_tot = torch.zeros(1, dtype=torch.int32, device=dev)
    d_lm_mpoff = torch.from_numpy(lm_off).to(dev)
    d_kpmp0 = torch.full((B * kp_cap,), -1, dtype=torch.int32, device=dev); d_kpmp = d_kpmp0.clone()
    ext_stream = torch.cuda.ExternalStream(ctx.stream, device=dev)

    def lm_windows(d_idx_buf, cap):
        ctx.is_in_frustum_device(n_mp, d_lm_frame.data_ptr(), d_lm_pose.data_ptr(), *[t.data_ptr() for t in d_lm_in], 0.5, 1.0,
                                 [d_lm_vis.data_ptr(), d_lm_f[0].data_ptr(), d_lm_f[1].data_ptr(), d_lm_i[0].data_ptr(), d_lm_f[2].data_ptr()],
                                 [d_lm_f[3].data_ptr(), d_lm_i[1].data_ptr(), d_lm_i[2].data_ptr()])
        ctx.features_in_area_batch_device(n_mp, d_lm_frame.data_ptr(), [d_lm_f[0].data_ptr(), d_lm_f[1].data_ptr(), d_lm_f[3].data_ptr(),
                                                                        d_lm_i[1].data_ptr(), d_lm_i[2].data_ptr()],
                                          d_lm_i[3].data_ptr(), d_lm_off.data_ptr(), d_idx_buf.data_ptr(), cap, d_lm_tot.data_ptr())

    lm_windows(torch.zeros(1, dtype=torch.int32, device=dev), 0)       # dry run: size the candidate buffer
    ctx.sync()
    lm_pairs = int(d_lm_tot.item())
    lm_cap = lm_pairs + 4096
    d_lm_idx = torch.zeros(lm_cap, dtype=torch.int32, device=dev); d_lm_pd = torch.zeros(lm_cap, dtype=torch.int16, device=dev)

    n_ba = max(1, B // args.ba_every)
    prob = synth.ba_problem(K=20, P=22150, obs_per_point=4, F=F, seed=42 + rank)
    # the local-BA windows of this step are independent LM problems: one host thread (LocalMapping-like) drives them as a batch,
    # concurrent with the frame path on its own stream
    bas = [api.BundleAdjuster(prob, device=local_rank) for _ in range(n_ba)]
    # tracking's pose-only optimisation (Optimizer::PoseOptimization, once per frame here; the reference calls it 1-3 times):
    # one problem per frame, ~600 matched map points with 10 % mismatches, resident on the device, one launch per step
    pose_probs = [synth.pose_problem(N=args.pose_edges, F=F, seed=1000 * rank + b, outlier_frac=0.1) for b in range(B)]
    po = api.PoseOptimizer(B, sum(len(p["Xw"]) for p in pose_probs), device=local_rank)
    po.upload(pose_probs)
    ba_err = []

    ba_ms = [0.0, 0]

    # the windows are split into `--ba-groups` groups, each advanced in lock-step by its own host thread on its own stream:
    # while one group waits for a Levenberg step (single-workgroup solve kernel, host decision), the other keeps the chip busy
    n_grp = max(1, min(args.ba_groups, n_ba))
    groups = [bas[g::n_grp] for g in range(n_grp)]

    # LocalMapping::CreateNewMapPoints in front of every window's BA: the window's key frame against its 20 best covisible neighbours
    # (~1650 features each, FeatureVectors of ~400 nodes), key frames resident on the device; one store + context (stream) per group
    tri_nn = 20
    tri_sets = [synth.keyframe_set(F, n_kf=tri_nn + 1, n_pts=2400, seed=300 + 10 * rank + w) for w in range(min(n_ba, 2))]
    for S in tri_sets:
        for k in S["kfs"]:           # mvKeyRays: the pixel's ray with unit depth on its face (CamModelGeneral::TransformCubemapToRays)
            _, r = synth.pixel_to_ray(F, k["x"].astype(np.float64), k["y"].astype(np.float64))
            k["rays"] = r.astype(np.float32)
    tri_ctx, tri_store, tri_jobs = [], [], []
    for gi, grp in enumerate(groups):
        cg = api.Context(camd, nfeatures=nfeat, max_batch=1, device=local_rank)
        st = api.KeyframeStore(cg, max_keyframes=len(grp) * (tri_nn + 1), max_features=2048, max_nodes=1024)
        jobs_g = []
        for wi in range(len(grp)):
            S = tri_sets[(gi + wi) % len(tri_sets)]
            base = wi * (tri_nn + 1)
            for i, k in enumerate(S["kfs"]):
                K, keep = api.make_keyframe(k)
                st.put(base + i, K)
            jobs_g.append((base, list(range(base + 1, base + tri_nn + 1))))
        tri_ctx.append(cg); tri_store.append(st); tri_jobs.append(jobs_g)
    tri_new = [0]

    def ba_worker(grp, gi):
        t_ba0 = time.perf_counter()
        try:
            res = tri_store[gi].create_new_map_points(tri_jobs[gi])
            tri_new[0] = sum(len(r[0]) for r in res)
            api.ba_optimize_many(grp, (5, 10))   # the group's windows share every launch (kb_ba_* kernels, one window per blockIdx.z)
            for ba in grp:                       # benchmark plumbing: put the initial estimate back for the next step (asynchronous; done
                ba.reset()                       # here, where the chip is quiet, rather than in front of the next step's first kernel)
        except Exception as e:  # surfaced after join
            ba_err.append(e)
        ba_ms[0] += 1e3 * (time.perf_counter() - t_ba0) / n_grp; ba_ms[1] += 1.0 / n_grp

    last_traj = [None]

    part = os.environ.get("CMS_BENCH_PART", "")      # developer knob: "ba" / "frames" times one half of the step alone (not a bench line)

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=n_grp)       # one standing host thread per window group (LocalMapping-like)

    def step(i):
        ths = [pool.submit(ba_worker, grp, gi) for gi, grp in enumerate(groups)] if part != "frames" else []   # the groups' standing host threads
        if part == "ba":
            for th in ths:
                th.result()
            return
        po.launch()                 # own stream, overlaps the frame path
        ctx.process(B, True)
        ctx.area_grid(B)            # Frame::AssignFeaturesToGrid of the B frames
        with torch.cuda.stream(ext_stream):
            d_kpmp.copy_(d_kpmp0)
        # TrackWithMotionModel's matcher: projection + windows + greedy best match (Hamming inside) + rotation histogram ...
        mm_windows(d_idx, cand_cap)
        ctx.search_local_points_device(B, d_mm_mpoff.data_ptr(), d_mm_desc.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), d_mm_pd.data_ptr(),
                                       -1.0, 100, d_kpmp.data_ptr(), d_mm_match.data_ptr())
        ctx.rotation_filter_device(B, d_mm_mpoff.data_ptr(), d_mm_ang.data_ptr(), d_kpmp.data_ptr(), d_mm_match.data_ptr(), d_mm_n.data_ptr(), True)
        # ... then TrackLocalMap's search over the key points that are still free
        lm_windows(d_lm_idx, lm_cap)
        ctx.search_local_points_device(B, d_lm_mpoff.data_ptr(), d_lm_desc.data_ptr(), d_lm_off.data_ptr(), d_lm_idx.data_ptr(), d_lm_pd.data_ptr(),
                                       0.8, 100, d_kpmp.data_ptr(), d_lm_i[4].data_ptr())
        ctx.sync()
        _, frame_poses, _, _ = po.fetch()
        for th in ths:
            th.result()
        if ba_err:
            raise ba_err[0]
        if world > 1 or args.force_gather:   # trajectory assembly on rank 0 over RCCL (72 B / frame, latency only)
            rec = cdist.make_records(rank, i * B + np.arange(B), frame_poses)   # the frames' optimised poses, TUM order
            traj = cdist.gather_trajectory(rec, device=dev, dst=0)
            if traj is not None:
                last_traj[0] = traj

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    ctx.profile(True)
    for i in range(args.warmup):
        step(i)
    stage_ms = {}
    barrier()
    ba_ms[0], ba_ms[1] = 0.0, 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        for k, v in (ctx.profile_ms().items() if part != "ba" else ()):
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    pool.shutdown()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    for k in stage_ms:
        stage_ms[k] /= max(args.steps, 1)
    if part:
        if rank == 0:
            print(json.dumps({"developer_part": part, "ms_per_step": round(1e3 * dt / args.steps, 3), "config": {"ba_ms_per_step": round(ba_ms[0] / max(ba_ms[1], 1), 3), "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()}}}))
        return

    # ---- roofline of the dominant extraction kernel (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
    sumP = sum(g.level_w[l] * g.level_h[l] for l in range(g.nlevels))
    P = [g.level_w[l] * g.level_h[l] for l in range(g.nlevels)]
    nkp = float(np.mean([len(k) for k in kps]))
    alg = {
        "remap": camd["Iw"] * camd["Ih"] + 5 * F * F * (1 + 4),                     # source + dest + one packed u32 LUT entry / px
        "pyramid": sum(P[l - 1] + P[l] for l in range(1, g.nlevels)),               # read level l-1, write level l
        "fast": sumP,                                                                # every pyramid pixel read once
        "describe": nkp * (43 * 43 + 32 + 24),                                       # raw patch + descriptor + key-point record
    }
    # dominant kernel of the extraction path = the one with the longest average launch (k_resize is 7 short launches)
    nl = {"pyramid": g.nlevels - 1}
    per_launch = {k: stage_ms.get(k, 0.0) / nl.get(k, 1) for k in ("remap", "pyramid", "fast", "describe")}
    dom = max(per_launch, key=per_launch.get)
    kname = {"remap": "k_remap", "pyramid": "k_resize", "fast": "k_fast_cells", "describe": "k_describe"}[dom]
    peak = 8000.0
    launches = nl.get(dom, 1)
    ach = alg[dom] * B / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms.get(dom, 0) > 0 else None
    # measured HBM traffic of that kernel: committed rocprofv3 PMC pass (FETCH_SIZE and WRITE_SIZE collected in separate runs,
    # tools/run_profiles.sh), valid for the batch size the passes were taken at (the default, F = 550)
    traffic = None
    pj = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    if os.path.exists(pj) and F == 550 and json.load(open(pj)).get("frames_per_dispatch", 64) == B:
        kk = json.load(open(pj))["kernels"].get(kname)
        if kk and "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
            # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section) -> doubled; unit KB
            traffic = int((2.0 * kk["FETCH_SIZE"]["per_dispatch"] + kk["WRITE_SIZE"]["per_dispatch"]) * 1024)
    # vector-ALU issue time of the same kernel from the committed SQ-counter pass (VALU instructions x 4 cycles per wave64 instruction /
    # (256 CUs x 4 SIMDs x 2.4 GHz)): how much of the launch is spent just issuing its vector instructions
    valu_us = None
    pm = os.path.join(ROOT, "profiles", "r01_pmc_instruction_mix.json")
    if os.path.exists(pm) and F == 550 and json.load(open(pm)).get("frames_per_dispatch", 64) == B:
        kk = json.load(open(pm))["kernels"].get(kname)
        if kk:
            valu_us = kk["valu_issue_bound_us"]
    roof = {"kernel": kname, "bound": "hbm", "achieved": None if ach is None else round(ach, 1), "peak": peak, "unit": "GB/s",
            "frac": None if ach is None else round(ach / peak, 4), "traffic": traffic,
            "ms_per_launch": round(stage_ms.get(dom, 0.0) / launches, 4), "valu_issue_bound_ms": None if valu_us is None else round(valu_us / 1e3, 4), "algorithmic_bytes_per_launch": int(alg[dom] * B / launches),
            "all_stages_GBps": {k: round(alg[k] * B / (stage_ms[k] * 1e-3) / 1e9, 1) for k in alg if stage_ms.get(k, 0) > 0}}
    fast_gbs = alg["fast"] * B / (stage_ms["fast"] * 1e-3) / 1e9 if stage_ms.get("fast", 0) > 0 else None
    # whole ORBextractor::operator() against SURVEY.md 8(d)'s compulsory traffic at the reference's stage granularity
    # (B_pyr + B_fast + B_blur + B_kp; the full-level blur pass is counted although the product never runs it)
    b_extract = (2 * P[0] + sum(P[l - 1] + P[l] for l in range(1, g.nlevels))) + sumP + 2 * sumP + nkp * (709 + 961 + 32 + 28)
    ext_ms = sum(stage_ms.get(k, 0.0) for k in ("pyramid", "fast", "octree", "cull", "describe"))
    extractor = None
    if ext_ms > 0:
        ext_gbs = b_extract * B / (ext_ms * 1e-3) / 1e9
        extractor = {"survey_bytes_per_frame": int(b_extract), "us_per_frame": round(1e3 * ext_ms / B, 2), "GBps": round(ext_gbs, 1),
                     "frac_of_8TBps": round(ext_gbs / peak, 4)}

    # ---- CPU baseline: the oracle, single thread, on a bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        import orc
        ocam = orc.make_camera(camd)
        m1, m2 = orc.build_lut(ocam)
        o = orc.Orb(nfeatures=nfeat)
        n = min(args.cpu_frames, B)
        t1 = time.perf_counter()
        descs, cpu_kps = [], []
        for b in range(n):
            cube = orc.fisheye_to_cubemap(ocam, m1, m2, frames[b])
            k, d = o.extract(ocam, cube, mask)
            descs.append(d); cpu_kps.append(k)
        t_ext = time.perf_counter() - t1
        # TrackWithMotionModel's SearchByProjection, then TrackLocalMap's SearchLocalPoints on what is left, per frame like the GPU leg
        t1 = time.perf_counter()
        kp_after = []
        for b in range(n):
            kb, db = fetched[b]
            kpm = np.full(len(kb), -1, np.int32)
            orc.search_by_projection_frames(ocam, mm[b]["pose12"][:9], mm[b]["pose12"][9:], kb["x"], kb["y"], kb["octave"], kb["angle"], db, mm[b]["scale_factors"],
                                            mm[b]["valid"], mm[b]["Xw"], mm[b]["octave"], mm[b]["angle"], mm[b]["desc"], kpm, th=15.0, check_ori=True)
            kp_after.append(kpm)
        t_match = time.perf_counter() - t1
        pairs = n
        t1 = time.perf_counter()
        for b in range(n):
            kb, db = fetched[b]
            fr = orc.is_in_frustum(ocam, lm[b]["pose15"], lm[b]["pos"], lm[b]["normal"], lm[b]["min_dist"], lm[b]["max_dist"])
            orc.search_local_points(ocam, kb["x"], kb["y"], kb["octave"], db, lm[b]["scale_factors"], fr, lm[b]["desc"], kp_after[b])
        t_local = (time.perf_counter() - t1) / n
        t1 = time.perf_counter()
        S0 = tri_sets[0]
        oks = [orc.make_keyframe(ocam, k) for k in S0["kfs"]]
        for _ in range(2):
            orc.create_new_map_points(ocam, oks[0][0], [k for k, _ in oks[1:]], S0["scale_factors"], S0["level_sigma2"], S0["kfs"][0]["mp"].copy())
        t_tri = (time.perf_counter() - t1) / 2
        t1 = time.perf_counter()
        n_cpu_ba = 3
        for _ in range(n_cpu_ba):
            orc.ba_run(prob)
        t_ba = (time.perf_counter() - t1) / n_cpu_ba
        t1 = time.perf_counter()
        for b in range(min(n, 8)):
            orc.pose_optimize(pose_probs[b])
        t_pose = (time.perf_counter() - t1) / min(n, 8)
        per_frame = t_ext / n + t_match / max(pairs, 1e-9) + t_local + t_pose + (t_ba + t_tri) / args.ba_every
        cpu = {"value": round(1.0 / per_frame, 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames remap+extract (%.1f ms/frame), frame-to-frame SearchByProjection over %.1f frames (%.2f ms/frame), local-map search "
                         "(%.2f ms/frame), pose-only optimisation (%.2f ms/frame), CreateNewMapPoints (%.1f ms per key frame), %d local-BA windows (%.1f ms each, 1 per %d frames); "
                         "oracle/liborc.so, single thread" %
                         (n, 1e3 * t_ext / n, pairs, 1e3 * t_match / max(pairs, 1e-9), 1e3 * t_local, 1e3 * t_pose, 1e3 * t_tri, n_cpu_ba, 1e3 * t_ba,
                          args.ba_every),
               "host_cores_available": os.cpu_count()}

    if rank == 0 and args.save_trajectory and last_traj[0] is not None:
        # rank 0's assembled trajectory of the last step in the reference's TUM format (System.cpp:238-268)
        tr = last_traj[0]
        q = tr[:, 5:9]
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Tcw = np.zeros((len(tr), 4, 4), np.float32)
        Tcw[:, 0, 0] = 1 - 2 * (y * y + z * z); Tcw[:, 0, 1] = 2 * (x * y - z * w); Tcw[:, 0, 2] = 2 * (x * z + y * w)
        Tcw[:, 1, 0] = 2 * (x * y + z * w); Tcw[:, 1, 1] = 1 - 2 * (x * x + z * z); Tcw[:, 1, 2] = 2 * (y * z - x * w)
        Tcw[:, 2, 0] = 2 * (x * z - y * w); Tcw[:, 2, 1] = 2 * (y * z + x * w); Tcw[:, 2, 2] = 1 - 2 * (x * x + y * y)
        Tcw[:, :3, 3] = tr[:, 2:5]; Tcw[:, 3, 3] = 1
        cdist.write_trajectory_tum(args.save_trajectory, tr[:, 1], Tcw)
    if rank == 0:
        total_frames = B * args.steps * world
        out = {
            "metric": "frames/sec (extract+match+localBA) on Lafida cam0",
            "value": round(total_frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (extract/match), f64 (BA)", "data": "synthetic",
            "config": {"workload": "Lafida cam0 synthetic stream, 754x480 fisheye, face=%d (%dx%d cross), nFeatures %d; per step %d frames: "
                                   "remap+ORB extract, frame grids, frame-to-frame SearchByProjection (projection + GetFeaturesInArea windows + greedy Hamming match + rotation histogram: "
                                   "%d map points, %d candidate pairs), "
                                   "local-map search (isInFrustum + SearchByProjection, %d map points, %d candidate pairs), pose-only optimisation (%d edges/frame), "
                                   "%d key frames x (CreateNewMapPoints against 20 neighbours + local BA window K=20, E=%d)"
                                   % (F, 3 * F, 3 * F, nfeat, B, nq, n_pairs, n_mp, lm_pairs, args.pose_edges, n_ba, len(prob["e_pose"])),
                       "frames_per_step_per_gpu": B, "keypoints_per_frame": round(nkp, 1), "ba_every_frames": args.ba_every,
                       "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                       "fast_kernel_GBps": None if fast_gbs is None else round(fast_gbs, 1),
                       "extractor_vs_survey_bytes": extractor,
                       "ba_windows_per_step": n_ba, "ba_ms_per_step": round(ba_ms[0] / max(ba_ms[1], 1), 3)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
