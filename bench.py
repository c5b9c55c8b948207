#!/usr/bin/env python3
"""bench.py -- frames/sec of the CubemapSLAM hot path (remap + ORB extract + matching + pose optimisation + local BA) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: bench.py starts its N ranks itself (one process per GPU through torch.distributed.run, rendezvous on 127.0.0.1) unless it
already runs under a launcher (RANK / WORLD_SIZE in the environment, e.g. the driver's own torch.distributed.run line); it fails if
the world it ends up in is not --gpus.

Workloads
  default (BASELINE.json metric, configs[2] geometry): synthetic Lafida cam0 streams, 754x480 fisheye, cube face F = 550 (1650^2
      cross), nFeatures 2000 / 8 levels / 1.2 / FAST 20-7.  Per GPU 32 camera streams x 8 consecutive frames = 256 frames per step.
      Weak scaling: every rank owns its own 32 streams.
  --camera front (configs[4]): 8 independent front_cam streams (1280x720, F = 650, nFeatures 3000), stream s -> GPU s mod G,
      --frames-per-stream consecutive frames per step.  Strong scaling: the 8 streams are split over the ranks.

One step = one batch of frames through, all on the device:
    remap -> pyramid -> FAST cells -> octree -> cull -> orientation + rBRIEF     (ORBextractor::operator(), all frames per launch)
    Frame::AssignFeaturesToGrid; TrackWithMotionModel's SearchByProjection(Cur, Last, th = 15): projection, GetFeaturesInArea windows,
        greedy Hamming match, rotation histogram; TrackLocalMap's SearchLocalPoints (isInFrustum + windows + greedy match)
    Optimizer::PoseOptimization of every frame (one launch for the batch, own stream)
    per `--ba-every` frames one key frame on the mapping side: LocalMapping::CreateNewMapPoints against 20 neighbours, then one
        Optimizer::LocalBundleAdjustment call (K = 20 key frames, ~80k cubemap edges, BASELINE.json configs[3]) with its WHOLE life cycle
        inside the step, as the reference has it (Optimizer.cpp:192-358 builds the graph per call, :419-450 writes the result back):
        cms_ba_create from the problem's host arrays (host work lists, one pinned upload), optimisation, cms_ba_read of poses / points /
        outlier flags into host arrays, cms_ba_destroy.  EVERY window of a step is a different problem (own seed; map points tracked over
        consecutive key frames, synth.ba_problem(views="track")) and two sets of problems alternate from step to step.  The windows advance
        in lock-step groups on their own streams / host threads (the reference's LocalMapping thread); a pool of host threads builds the
        windows of step s + 1 while step s runs and reads back / destroys the windows of step s -- all of it inside the timed region.
Two frame batches (different streams) alternate from step to step, so no step replays the previous step's inputs.

`value` is measured with both batches resident in HBM (a device-to-device copy into the staging buffer opens the step).  A second
timed pass streams the same batches from pinned host memory (cms_frames_upload_async: the copy of step s + 1 overlaps step s) and
is reported as `config.with_input_streaming`.

After the timed passes, outside any timed region: the local-BA results of a sample of windows are compared with the CPU oracle
(iterations of both stages, outlier flags, updates), the CPU baseline is taken (rank 0, N = 1 only) and ONE JSON line is printed
(rank 0) with `roofline` (dominant kernel of the step, HIP-event timing inside the timed region) and `cpu_baseline`.
"""
import argparse
import collections
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

# The frame path, every local-BA window group, the mapping contexts and the window pool run on their own HIP streams; the ROCm runtime multiplexes
# streams onto GPU_MAX_HW_QUEUES hardware queues (its default: 4).  Must be set before the HIP runtime initialises.  Round 1 (every window a stream of its
# own, four concurrent windows): 8 queues = +28 % frames/s over 4.  Round 5 (two window groups on long-lived streams, set-ups on pool streams, the mapping
# side's calls): measured again -- 2 / 3 / 4 / 6 / 8 / 16 / 24 queues = 18.6 / 19.2 / 20.2-21.6 / 18.9-19.7 / 19.7-20.4 / 19.8-20.2 / 14.8-16.4 k frames/s
# (profiles/r05_experiments.txt): with fewer queues the frame path's kernels queue behind each other instead of competing with the Levenberg chain for
# compute units (extractor inside the step 0.35-0.36 of its byte roofline against 0.31-0.32), with two or three the chains themselves serialise.
# Round 6: WHICH streams share a queue is what counts (the runtime hands its queues to new streams in a fixed cycle; tools/stream_queues.py on a profiled run): with four, one
# window group's rounds share a queue with a set-up stream -- harmless while the host plans (short uploads), but a rank that plans on the device (<= 2 cores) has its
# plan kernels there, in front of that group's rounds: 1-ms gaps inside the rounds, 20.2 k frames/s; with six the set-up streams sit with the pose optimiser's / alone:
# 21.4 k (three runs each, 2 cores).  With cores to spare four stays best (23.0 against 22.0 / 22.6 / 21.9 / 22.5 k with 5 / 6 / 10 / 12).  main() picks once the budget is known.
_HWQ_FROM_USER = "GPU_MAX_HW_QUEUES" in os.environ
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cbd_types():
    """ctypes descriptions of cubemapslam_amd/host/batch_driver.cpp's plan structures (cbd_frame_set, cbd_group, cbd_plan, cbd_stats); bench.py checks their sizes
    against the library's own (cbd_sizes) before it uses them, tests/test_batch_driver_cpu.py does it without a GPU"""
    import ctypes as C_
    VP = C_.c_void_p
    class CbdFrameSet(C_.Structure):
        _fields_ = [("d_frames", VP), ("nq", C_.c_int), ("d_mm_frame", VP), ("d_mm_pose", VP), ("d_mm_valid", VP), ("d_mm_xw", VP), ("d_mm_oct", VP), ("d_mm_q", VP * 5),
                    ("d_cnt", VP), ("d_off", VP), ("d_idx", VP), ("cand_cap", C_.c_int), ("d_tot", VP),
                    ("d_mm_mpoff", VP), ("d_mm_desc", VP), ("d_mm_pd", VP), ("d_mm_match", VP), ("d_mm_ang", VP), ("d_mm_n", VP),
                    ("n_mp", C_.c_int), ("d_lm_frame", VP), ("d_lm_pose", VP), ("d_lm_in", VP * 4), ("d_lm_vis", VP), ("d_lm_f", VP * 4), ("d_lm_i", VP * 5),
                    ("d_lm_off", VP), ("d_lm_idx", VP), ("lm_cap", C_.c_int), ("d_lm_tot", VP), ("d_lm_mpoff", VP), ("d_lm_desc", VP), ("d_lm_pd", VP),
                    ("d_kpmp", VP), ("d_kpmp0", VP), ("kpmp_bytes", C_.c_size_t), ("put_items", VP * 8), ("put_n", C_.c_int * 8)]
    class CbdGroup(C_.Structure):
        _fields_ = [("store", VP), ("ba_stream", VP),
                    ("njobs", C_.c_int), ("cur_slot", VP), ("neigh_off", VP), ("neigh_slot", VP), ("cap", C_.c_int), ("n_new", VP), ("o_neigh", VP), ("o_idx1", VP), ("o_idx2", VP), ("o_x3d", VP),
                    ("nsets", C_.c_int), ("set_off", VP), ("pos", VP), ("normal", VP), ("min_d", VP), ("max_d", VP), ("desc", VP),
                    ("nfjobs", C_.c_int), ("job_slot", VP), ("job_set", VP), ("skip", VP), ("th", C_.c_float), ("best_idx", VP), ("best_dist", VP),
                    ("n_upd", C_.c_int), ("upd_slots", VP), ("upd_R", VP), ("upd_t", VP), ("upd_Ow", VP),
                    ("nwin", C_.c_int), ("windows", VP * 2)]
    STEP_DONE = C_.CFUNCTYPE(None, VP, C_.c_int, C_.POINTER(C_.c_double), C_.c_int)
    class CbdPlan(C_.Structure):
        _fields_ = [("ctx", VP), ("po", VP), ("B", C_.c_int), ("device", C_.c_int), ("ngroups", C_.c_int), ("create_threads", C_.c_int), ("mapping_full", C_.c_int),
                    ("ahead", C_.c_int), ("n_pose_edges", C_.c_int), ("groups", CbdGroup * 8), ("sets", CbdFrameSet * 2), ("step_done", STEP_DONE), ("user", VP)]
    class CbdStats(C_.Structure):
        _fields_ = [("ba_ms_sum", C_.c_double), ("ba_jobs", C_.c_long), ("schur_ms", C_.c_double), ("schur_launches", C_.c_long), ("create_ms_sum", C_.c_double),
                    ("create_windows", C_.c_long), ("tri_ms_sum", C_.c_double), ("fuse_ms_sum", C_.c_double), ("put_ms_sum", C_.c_double), ("upd_ms_sum", C_.c_double),
                    ("tri_calls", C_.c_long), ("fuse_calls", C_.c_long), ("put_calls", C_.c_long), ("upd_calls", C_.c_long),
                    ("wait_windows_ms", C_.c_double), ("wait_tri_ms", C_.c_double), ("optimize_ms", C_.c_double),
                    ("new_map_points_last_step", C_.c_long), ("fused_last_call", C_.c_long), ("stage_ms", C_.c_float * 7), ("steps", C_.c_long)]
    return VP, CbdFrameSet, CbdGroup, STEP_DONE, CbdPlan, CbdStats


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--camera", choices=("lafida", "front"), default="lafida", help="front = BASELINE.json configs[4]: 8 front_cam streams, 1280x720, F=650")
    ap.add_argument("--face", type=int, default=0, help="cube face size (default 550 for lafida, 650 for front)")
    ap.add_argument("--batch", type=int, default=256, help="lafida: frames per step per GPU (default 32 streams x 8 consecutive frames)")
    ap.add_argument("--streams", type=int, default=8, help="front: camera streams in total (stream s -> GPU s mod G)")
    ap.add_argument("--frames-per-stream", type=int, default=16, help="front: consecutive frames of every stream per step")
    ap.add_argument("--ba-every", type=int, default=8, help="one local-BA window per this many frames")
    ap.add_argument("--ba-groups", type=int, default=2, help="host threads / streams the local-BA windows of a step are split over (2 groups of 16 windows: "
                    "11.5 ms per step against 11.8 with 3 groups and 12.3 with 4, interleaved runs; the extraction kernels keep 41 %% of their byte roofline either way)")
    ap.add_argument("--pose-edges", type=int, default=600, help="matched map points per frame for the pose-only optimisation")
    ap.add_argument("--save-trajectory", default="", help="rank 0 writes the gathered trajectory of the last step here (TUM format)")
    ap.add_argument("--force-gather", action="store_true", help="run the trajectory gather code path even with one rank (self-test)")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-streaming-pass", action="store_true", help="skip the second timed pass with inputs streamed from pinned host memory")
    ap.add_argument("--verify-windows", type=int, default=8, help="local-BA windows whose estimates are checked against the CPU oracle after the timed region (rank 0); "
                    "the iteration counts and outlier counts of ALL windows of that step are checked as well (0 = no check)")
    ap.add_argument("--ba-views", choices=("track", "random"), default="track", help="how the synthetic windows' observations are drawn (synth.ba_problem)")
    ap.add_argument("--window-threads", type=int, default=0, help="host threads that build / read back / destroy local-BA windows (0 = this rank's share of the "
                    "node's usable cores minus two, between 4 and 32: host_budget())")
    ap.add_argument("--extract-only-steps", type=int, default=10, help="steps of the extra pass that runs the frame path alone (config.extract_only; 0 = skip)")
    ap.add_argument("--random-views-steps", type=int, default=8, help="steps of the extra pass on windows with RANDOM views (no signature runs: round 2's windows), "
                    "reported as config.ba_views_random next to the headline (0 = skip)")
    ap.add_argument("--optimise-only-steps", type=int, default=10, help="steps of the extra pass that keeps the windows and only resets them between steps "
                    "(round 2's headline, reported as config.optimise_only; 0 = skip)")
    ap.add_argument("--unpipelined-steps", type=int, default=10, help="steps of the loop that waits for every step's own mapping side (config.unpipelined; 0 = skip)")
    ap.add_argument("--deterministic-steps", type=int, default=6, help="steps of the extra pass with cms_ba_set_deterministic(1) (config.deterministic; 0 = skip)")
    ap.add_argument("--mapping-only-steps", type=int, default=6, help="steps of the mapping side alone (CreateNewMapPoints + local BA, no frame path): config.mapping_only, and the Schur kernel's launch time without the frame path next to it in roofline (0 = skip)")
    ap.add_argument("--closed-loop-frames", type=int, default=24, help="frames of the single-stream closed-loop run reported next to the batch figure (rank 0, N = 1; 0 = skip)")
    ap.add_argument("--confined-steps", type=int, default=40, help="steps of the two child runs of the same step with the process confined to 2 and to 4 host cores "
                    "(config.host.confined_2_cores / _4_cores: what a rank gets when eight of them share a 16-core box; rank 0, N = 1; 0 = skip)")
    ap.add_argument("--launcher-selftest", action="store_true", help="no GPU work: every rank reports its rendezvous (gloo) and exits")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks, one per GPU (one process per GPU)."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not _HWQ_FROM_USER:
        env.pop("GPU_MAX_HW_QUEUES", None)      # (this process's default, not the user's wish: every rank picks its own from ITS core budget)
    env["CMS_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def host_budget(world, local_rank, window_threads_arg=0, pin=True):
    """Host cores of THIS rank: the ranks of a node split the cores the process may use (scheduler affinity, bounded by the cgroup's CPU quota)
    into disjoint slices -- every rank builds, reads back and destroys its local-BA windows on host threads, and eight ranks that each size
    their pool for the whole machine throttle each other (10 cores per rank on a 16-core quota).  With more than one rank on the
    node the process is pinned to its slice (CMS_BENCH_NO_PIN=1: not).  Returns a dict that goes into the JSON line."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    usable = len(cores) if quota is None else min(len(cores), quota)
    budget = max(2, usable // local_world)
    # a rank's slice sits in the MIDDLE of its share of the visible cores when the quota leaves room (256 visible, 16 usable: rank 0 gets cores 15-16, not 0-1 --
    # the first cores of a box carry its interrupts and other tenants' spill-over: a two-core process there measured 16-19 k frames/s against 19.3-19.6 k elsewhere)
    stride = len(cores) // local_world
    lo = (local_rank % local_world) * stride + max(0, (stride - max(budget, 1)) // 2)
    mine = cores[lo:lo + max(budget, 1)] if local_world > 1 else cores
    pinned = False
    if pin and local_world > 1 and mine and os.environ.get("CMS_BENCH_NO_PIN", "") == "":
        try:
            os.sched_setaffinity(0, mine)
            pinned = True
        except (AttributeError, OSError):
            pass
    # pool threads that build, read back and destroy windows: one per TWO cores of the slice since round 5 (between 4 and 16; one per core before).  A
    # window's set-up is 0.6 ms of CPU now (device-side planner) instead of 2.5, and the threads no longer spin while the set-up waits for its turn
    # on the device (cms_ba_set_stream's event): eight threads build a step's 32 windows in ~3 ms, sixteen only add waiters on the runtime's locks
    # (measured 22.9 k against 22.0-22.3 k frames/s with round 4's step).  CMS_BA_RELAXED_WAIT=1 (read once by the library) lets the remaining waits
    # sleep instead of spin; with it 1.5x the threads
    relaxed = os.environ.get("CMS_BA_RELAXED_WAIT", "") != ""
    wthreads = window_threads_arg or max(4, min(16, (3 * budget) // 4 if relaxed else budget // 2))
    return {"cores_visible": len(cores), "cpu_quota_cores": quota, "local_world_size": local_world, "thread_budget": budget,
            "core_slice": [mine[0], mine[-1]] if mine else None, "pinned": pinned, "window_threads": wthreads, "host_waits": "sleep" if relaxed else "spin"}


def plans_on_device(thread_budget, env=None):
    """Does this rank let k_ba_plan_many plan its local-BA windows (cms_ba_window.flags |= CMS_BA_PLAN_ON_DEVICE)?  With at most two host cores (eight ranks on a
    16-core box): yes -- the plans are half of such a rank's CPU time per step; with more the host's plan overlaps the GPU better.  CMS_BENCH_PLAN_ON_DEVICE=1 / 0 overrides."""
    v = (os.environ if env is None else env).get("CMS_BENCH_PLAN_ON_DEVICE", "")
    return (v == "1") if v in ("0", "1") else thread_budget <= 2


# ---- the same step with the whole process confined to 2 and to 4 host cores (child processes, run before the parent touches the GPU): the driver's box gives
# 8 ranks a 16-core quota, two cores each -- SCALE_rNN cannot be measured on a one-GPU lease, this is the part of it that can
def confined_leg(args, ncores):
    try:
        mask = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if len(mask) < ncores:
        return None
    lo = max(0, min(len(mask) - ncores, len(mask) // 2))      # from the middle of the mask: the first cores of a box are where its interrupts and other tenants' spill-over land
    cores = mask[lo:lo + ncores]
    env = dict(os.environ); env["CMS_BENCH_AFFINITY"] = ",".join(str(c) for c in cores)
    if not _HWQ_FROM_USER:
        env.pop("GPU_MAX_HW_QUEUES", None)      # (this process's default, not the user's wish: the child picks its own from ITS core budget)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.confined_steps), "--warmup", "3", "--cpu-frames", "0", "--no-streaming-pass", "--verify-windows", "0",
           "--extract-only-steps", "0", "--random-views-steps", "0", "--optimise-only-steps", "0", "--unpipelined-steps", "0", "--deterministic-steps", "0",
           "--mapping-only-steps", "0", "--closed-loop-frames", "0", "--confined-steps", "0", "--camera", args.camera, "--batch", str(args.batch), "--ba-every", str(args.ba_every)]
    import shutil
    if shutil.which("taskset"):      # the whole child process from its first instruction on (threads that exist before main() runs keep their own mask otherwise)
        cmd = ["taskset", "-c", env["CMS_BENCH_AFFINITY"]] + cmd
    t0_ = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"cores": ncores, "state": "timed out"}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"cores": ncores, "state": "failed (rc %d)" % r.returncode, "stderr_tail": r.stderr[-300:]}
    j = json.loads(lines[-1])
    h = j["config"].get("host") or {}
    return {"cores": ncores, "core_list": cores, "value": j["value"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "host_cores_used": h.get("host_cores_used"),
            "host_waits": h.get("host_waits"), "window_threads": h.get("window_threads"), "window_plans": h.get("window_plans"), "gpu_max_hw_queues": h.get("gpu_max_hw_queues"),
            "child_wall_s": round(time.perf_counter() - t0_, 1)}


def launcher_selftest(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank], dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        seen = sorted(int(g.item()) for g in got)
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = [0]
    # one write per record: the ranks share the launcher's stdout, and a record and its newline written separately can interleave
    hb = host_budget(world, local_rank, args.window_threads)
    line = json.dumps({"launcher_selftest": True, "rank": rank, "local_rank": local_rank, "world": world, "gpus_arg": args.gpus,
                       "ranks_seen": seen, "pid": os.getpid(), "spawned_by_bench": os.environ.get("CMS_BENCH_SPAWNED") == "1", "host": hb,
                       "affinity_now": sorted(os.sched_getaffinity(0))[:4] + ["..."] if hasattr(os, "sched_getaffinity") else None}) + "\n"
    sys.stdout.flush()
    os.write(sys.stdout.fileno(), line.encode())


def make_stream_frames(camd, n, seed):
    """n consecutive frames of one synthetic camera stream: a seeded texture drifting 3 px / frame under the fisheye."""
    from cubemapslam_amd import synth
    Ih, Iw = camd["Ih"], camd["Iw"]
    big = synth.texture(Ih + 4 * n + 8, Iw + 4 * n + 8, seed)
    return np.stack([big[2 * b:2 * b + Ih, 3 * b:3 * b + Iw] for b in range(n)])


class TrackSet:
    """Everything the tracking side of a step needs for ONE batch of frames, resident on the device: the frames (staging layout), and --
    built from the key points this batch extracts to -- per frame a predicted pose with the ~1400 map points of its last frame
    (TrackWithMotionModel) and its local map (~2150 points, TrackLocalMap)."""

    def __init__(self, ctx, torch, dev, camd, F, stream_seeds, frames_per_stream, seed0):
        from cubemapslam_amd import api, synth
        self.ctx, self.torch = ctx, torch
        g = ctx.geom
        B = len(stream_seeds) * frames_per_stream
        self.B = B
        Ih, Iw, fs = camd["Ih"], camd["Iw"], g.fisheye_stride
        self.pinned = api.PinnedArray((B, Ih, fs))          # host copy in the device staging layout (input streaming)
        self.pinned.array[:] = 0
        self.stream_of_frame = np.repeat(np.arange(len(stream_seeds)), frames_per_stream)
        for si, sd in enumerate(stream_seeds):
            self.pinned.array[si * frames_per_stream:(si + 1) * frames_per_stream, :, :Iw] = make_stream_frames(camd, frames_per_stream, sd)
        self.d_frames = torch.from_numpy(self.pinned.array).to(dev)        # resident copy
        ctx.upload_device(self.d_frames.data_ptr(), B)
        ctx.process(B, True)
        ctx.sync()
        self.fetched = [ctx.fetch(b) for b in range(B)]
        fetched = self.fetched
        ctx.area_grid(B)
        # ---- frame-to-frame matching (ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th = 15), Tracking::TrackWithMotionModel)
        self.mm = mm = [synth.motion_model_problem(F, k["x"], k["y"], k["octave"], k["angle"], d, seed=seed0 + 5000 + b) for b, (k, d) in enumerate(fetched)]
        mm_off = np.concatenate([[0], np.cumsum([len(p["valid"]) for p in mm])]).astype(np.int32)
        self.nq = nq = int(mm_off[-1])
        mcat = lambda key, dt: torch.from_numpy(np.concatenate([p[key] for p in mm]).astype(dt)).to(dev)
        self.d_mm_pose = torch.from_numpy(np.stack([p["pose12"] for p in mm])).to(dev)
        self.d_mm_frame = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), np.diff(mm_off))).to(dev)
        self.d_mm_valid, self.d_mm_xw, self.d_mm_oct, self.d_mm_ang, self.d_mm_desc = (mcat("valid", np.uint8), mcat("Xw", np.float32), mcat("octave", np.int32),
                                                                                      mcat("angle", np.float32), mcat("desc", np.uint8))
        self.d_mm_q = [torch.zeros(nq, dtype=torch.float32, device=dev) for _ in range(3)] + [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(2)]
        self.d_cnt = torch.zeros(max(nq, 1), dtype=torch.int32, device=dev); self.d_off = torch.zeros(nq + 1, dtype=torch.int32, device=dev)
        self.d_tot = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_mm_match = torch.zeros(max(nq, 1), dtype=torch.int32, device=dev); self.d_mm_n = torch.zeros(B, dtype=torch.int32, device=dev)
        self.d_mm_mpoff = torch.from_numpy(mm_off).to(dev)
        self.mm_windows(torch.zeros(1, dtype=torch.int32, device=dev), 0)       # dry run: size the candidate buffer
        ctx.sync()
        self.n_pairs = int(self.d_tot.item())
        self.cand_cap = self.n_pairs + 4096
        self.d_idx = torch.zeros(self.cand_cap, dtype=torch.int32, device=dev); self.d_mm_pd = torch.zeros(self.cand_cap, dtype=torch.int16, device=dev)
        # ---- track local map (Tracking::SearchLocalPoints): every frame has its own pose and local map
        self.lm = lm = [synth.local_map_problem(F, k["x"], k["y"], k["octave"], d, seed=seed0 + 7000 + b) for b, (k, d) in enumerate(fetched)]
        lm_off = np.concatenate([[0], np.cumsum([len(p["pos"]) for p in lm])]).astype(np.int32)
        self.n_mp = n_mp = int(lm_off[-1])
        lcat = lambda key, dt: torch.from_numpy(np.concatenate([p[key] for p in lm]).astype(dt)).to(dev)
        self.d_lm_pose = torch.from_numpy(np.stack([p["pose15"] for p in lm])).to(dev)
        self.d_lm_frame = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), np.diff(lm_off))).to(dev)
        self.d_lm_in = [lcat("pos", np.float32), lcat("normal", np.float32), lcat("min_dist", np.float32), lcat("max_dist", np.float32)]
        self.d_lm_desc = lcat("desc", np.uint8)
        self.d_lm_vis = torch.zeros(n_mp, dtype=torch.uint8, device=dev)
        self.d_lm_f = [torch.zeros(n_mp, dtype=torch.float32, device=dev) for _ in range(4)]          # proj_x, proj_y, view_cos, qr
        self.d_lm_i = [torch.zeros(n_mp, dtype=torch.int32, device=dev) for _ in range(5)]            # level, qmin, qmax, cnt, match
        self.d_lm_off = torch.zeros(n_mp + 1, dtype=torch.int32, device=dev); self.d_lm_tot = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_lm_mpoff = torch.from_numpy(lm_off).to(dev)
        self.lm_windows(torch.zeros(1, dtype=torch.int32, device=dev), 0)       # dry run: size the candidate buffer
        ctx.sync()
        self.lm_pairs = int(self.d_lm_tot.item())
        self.lm_cap = self.lm_pairs + 4096
        self.d_lm_idx = torch.zeros(self.lm_cap, dtype=torch.int32, device=dev); self.d_lm_pd = torch.zeros(self.lm_cap, dtype=torch.int16, device=dev)
        kp_cap = g.kp_cap
        self.d_kpmp0 = torch.full((B * kp_cap,), -1, dtype=torch.int32, device=dev); self.d_kpmp = self.d_kpmp0.clone()

    def mm_windows(self, d_idx_buf, cap):
        c = self.ctx
        c.project_last_frame_device(self.nq, self.d_mm_frame.data_ptr(), self.d_mm_pose.data_ptr(), self.d_mm_valid.data_ptr(), self.d_mm_xw.data_ptr(),
                                    self.d_mm_oct.data_ptr(), 15.0, [t.data_ptr() for t in self.d_mm_q])
        c.features_in_area_batch_device(self.nq, self.d_mm_frame.data_ptr(), [t.data_ptr() for t in self.d_mm_q], self.d_cnt.data_ptr(), self.d_off.data_ptr(),
                                        d_idx_buf.data_ptr(), cap, self.d_tot.data_ptr())

    def lm_windows(self, d_idx_buf, cap):
        c = self.ctx
        c.is_in_frustum_device(self.n_mp, self.d_lm_frame.data_ptr(), self.d_lm_pose.data_ptr(), *[t.data_ptr() for t in self.d_lm_in], 0.5, 1.0,
                               [self.d_lm_vis.data_ptr(), self.d_lm_f[0].data_ptr(), self.d_lm_f[1].data_ptr(), self.d_lm_i[0].data_ptr(), self.d_lm_f[2].data_ptr()],
                               [self.d_lm_f[3].data_ptr(), self.d_lm_i[1].data_ptr(), self.d_lm_i[2].data_ptr()])
        c.features_in_area_batch_device(self.n_mp, self.d_lm_frame.data_ptr(), [self.d_lm_f[0].data_ptr(), self.d_lm_f[1].data_ptr(), self.d_lm_f[3].data_ptr(),
                                                                                  self.d_lm_i[1].data_ptr(), self.d_lm_i[2].data_ptr()],
                                        self.d_lm_i[3].data_ptr(), self.d_lm_off.data_ptr(), d_idx_buf.data_ptr(), cap, self.d_lm_tot.data_ptr())

    def enqueue_tracking(self, ext_stream):
        """everything after the extraction for this batch, on the ctx stream"""
        c, B, torch = self.ctx, self.B, self.torch
        c.area_grid(B)            # Frame::AssignFeaturesToGrid of the B frames
        with torch.cuda.stream(ext_stream):
            self.d_kpmp.copy_(self.d_kpmp0)
        # TrackWithMotionModel's matcher: projection + windows + greedy best match (Hamming inside) + rotation histogram ...
        self.mm_windows(self.d_idx, self.cand_cap)
        c.search_local_points_device(B, self.d_mm_mpoff.data_ptr(), self.d_mm_desc.data_ptr(), self.d_off.data_ptr(), self.d_idx.data_ptr(), self.d_mm_pd.data_ptr(),
                                     -1.0, 100, self.d_kpmp.data_ptr(), self.d_mm_match.data_ptr())
        c.rotation_filter_device(B, self.d_mm_mpoff.data_ptr(), self.d_mm_ang.data_ptr(), self.d_kpmp.data_ptr(), self.d_mm_match.data_ptr(), self.d_mm_n.data_ptr(), True)
        # ... then TrackLocalMap's search over the key points that are still free
        self.lm_windows(self.d_lm_idx, self.lm_cap)
        c.search_local_points_device(B, self.d_lm_mpoff.data_ptr(), self.d_lm_desc.data_ptr(), self.d_lm_off.data_ptr(), self.d_lm_idx.data_ptr(), self.d_lm_pd.data_ptr(),
                                     0.8, 100, self.d_kpmp.data_ptr(), self.d_lm_i[4].data_ptr())


def main():
    # developer knob: every hipStreamSynchronize / hipEventSynchronize of this process blocks instead of spinning (hipDeviceScheduleBlockingSync);
    # must be set before the first HIP call of the process
    args = parse_args()
    if os.environ.get("CMS_BENCH_AFFINITY", "") and hasattr(os, "sched_setaffinity"):      # a confined child leg (see confined_leg): before any thread exists
        os.sched_setaffinity(0, {int(c) for c in os.environ["CMS_BENCH_AFFINITY"].split(",")})
    maybe_spawn(args)
    # The confined child legs run FIRST, while this process has not touched the GPU yet: an idle parent's context (its hardware queues, its memory) on the same
    # GPU cost a two-core child ~6 % (19.0-19.7 k frames/s inside a default run against 21.3 k for the same command on its own, profiles/r06_bench_runs.txt)
    early_confined = {}
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and args.confined_steps > 0 and args.batch // max(args.ba_every, 1) > 0
            and os.environ.get("CMS_BENCH_AFFINITY", "") == "" and not args.launcher_selftest):
        early_confined["confined_2_cores"] = confined_leg(args, 2)
        early_confined["confined_4_cores"] = confined_leg(args, 4)
        early_confined["confined_note"] = ("child processes of this bench.py with sched_setaffinity to 2 / 4 cores from the middle of the parent's affinity mask (CMS_BENCH_AFFINITY), same step, "
                                           "no extra passes; they run before the parent's first HIP call, alone on the GPU")
    # A rank with few host cores (8 ranks on a node whose container has a 16-core quota: two each) cannot afford the runtime's spinning waits: with
    # <= 4 cores the process blocks in its synchronisations and the library's window threads sleep between stream queries (CMS_BA_RELAXED_WAIT).
    # Measured on one GPU with the process confined by taskset: 2 cores 12.5 k -> 13.2-13.8 k frames/s, 4 cores 17.7 k -> 19.1 k; with cores to spare
    # the spinning waits stay (sleeping costs 3-16 % there, round 4).  Both switches must be thrown before the first HIP call of the process.
    early_budget = host_budget(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")), pin=False)["thread_budget"]
    if not _HWQ_FROM_USER:
        os.environ["GPU_MAX_HW_QUEUES"] = "6" if plans_on_device(early_budget) else "4"      # (see the top of this file; before the first HIP call)
    few_cores = early_budget <= 4 and os.environ.get("CMS_BENCH_SPIN_ANYWAY", "") == ""
    if few_cores:
        os.environ.setdefault("CMS_BA_RELAXED_WAIT", "1")
    blocking_sync = os.environ.get("CMS_BENCH_BLOCKING_SYNC", "") != "" or few_cores
    if blocking_sync:
        import ctypes
        ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(ctypes.c_uint(0x4))
    # ~25 Python threads drive this process (window pool, window groups, mapping threads, the frame path); every library call releases the interpreter
    # lock and has to take it again when it returns.  With the default 5-ms switch interval a returning thread can wait milliseconds for a thread that
    # is merely running Python glue -- on the mapping side's critical path (CreateNewMapPoints -> Fuse -> local BA) that wait was most of the calls'
    # measured time.  Bench plumbing (a C++ host has no such lock); CMS_BENCH_SWITCH_INTERVAL_US overrides.
    sys.setswitchinterval(1e-6 * float(os.environ.get("CMS_BENCH_SWITCH_INTERVAL_US", "200")))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (RANK/WORLD_SIZE environment)" % (args.gpus, world))
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world, local_rank)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # developer smoke test of the N > 1 control flow on a box with ONE GPU (CMS_BENCH_SHARED_GPU_GLOO=1): every rank drives device 0 and the
    # collectives run over gloo on host tensors.  Not a measurement (the ranks share the chip) -- it exists because the multi-rank branches cannot
    # otherwise run before the driver's scaling bench does.
    shared_gpu_debug = os.environ.get("CMS_BENCH_SHARED_GPU_GLOO", "") != "" and world > 1
    if shared_gpu_debug:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu_debug:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: RCCL sees %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    from cubemapslam_amd import api, build, synth
    from cubemapslam_amd import dist as cdist
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()

    # ---- workload
    front = args.camera == "front"
    F = args.face or (650 if front else 550)
    camd = synth.camera(args.camera, F)
    nfeat = camd["nfeatures"]
    if front:       # configs[4]: the streams are split over the ranks (strong scaling)
        my_streams = cdist.streams_of_rank(args.streams, world, rank)
        if not my_streams:
            raise SystemExit("bench.py: rank %d owns no stream (%d streams on %d GPUs)" % (rank, args.streams, world))
        fps = args.frames_per_stream
        total_frames_per_step = args.streams * fps
        scaling = "strong"
    else:           # every rank has its own 32 streams (weak scaling)
        fps = 8
        n_str = max(1, args.batch // fps)
        my_streams = [rank * n_str + s for s in range(n_str)]
        total_frames_per_step = n_str * fps * world
        scaling = "weak"
    B = len(my_streams) * fps
    dev = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if shared_gpu_debug else dev       # where the collectives' tensors live (RCCL: the GPU)
    # The frame path's queue runs at LOW dispatch priority (CMS_FRAME_STREAM_PRIORITY, read by cms_ctx_create): its kernels are few and
    # chip-filling, the mapping side's are a long chain of short dependent launches -- when both have workgroups ready the chain goes first.
    # Measured 13.9-14.1 against 14.6-15.5 ms per step (tools/experiments_r03/r03_run14.sh; "high" does the same: what counts is that the queue classes
    # differ).  CMS_BENCH_FRAME_PRIORITY=normal|high|low overrides; the mapping side's contexts keep the default.
    # Round 4: with the steps pipelined (the mapping side of step s next to the frame path of step s + 1) the step takes the same 11.0-11.3 ms
    # with either class, and the extractor keeps 0.35 instead of 0.30 of its byte roofline inside the step at normal priority: the default again.
    fprio = os.environ.get("CMS_BENCH_FRAME_PRIORITY", "normal")
    fprio = "" if fprio == "normal" else fprio
    if fprio:
        os.environ["CMS_FRAME_STREAM_PRIORITY"] = fprio
    # CMS_BENCH_CU_SPLIT=n (developer experiment): the frame path's streams (this context, the pose optimiser) confined to n compute units, the window groups'
    # streams to the other 256 - n (hipExtStreamCreateWithCUMask through CMS_CTX_CU_MASK; bit i of the mask = compute unit i of the runtime's enumeration)
    cu_split = int(os.environ.get("CMS_BENCH_CU_SPLIT", "0") or 0)
    def cu_mask_env(lo, hi):
        bits = sum(1 << i for i in range(lo, hi))
        return ",".join("%08x" % ((bits >> (32 * j)) & 0xFFFFFFFF) for j in range(8))
    if 0 < cu_split < 256:
        os.environ["CMS_CTX_CU_MASK"] = cu_mask_env(0, cu_split)
    ctx = api.Context(camd, nfeatures=nfeat, max_batch=B, device=local_rank)
    if fprio:
        del os.environ["CMS_FRAME_STREAM_PRIORITY"]
    mask = synth.cubemap_valid_mask(camd)
    ctx.set_mask(mask)
    g = ctx.geom
    # two batches of frames (all streams at time t and at time t + fps): consecutive steps never see the same inputs
    sets = [TrackSet(ctx, torch, dev, camd, F, [100 + 977 * j + s for s in my_streams], fps, seed0=100000 * j + 100 * rank) for j in range(2)]
    ext_stream = torch.cuda.ExternalStream(ctx.stream, device=dev)
    kps = [f[0] for S in sets for f in S.fetched]

    # ---- mapping side: one key frame per `ba_every` frames = CreateNewMapPoints + one local-BA window, every window its own problem
    n_ba = max(1, B // args.ba_every)
    from concurrent.futures import ThreadPoolExecutor, wait as fut_wait
    with ThreadPoolExecutor(max_workers=8) as tp:       # two sets of window problems alternate from step to step, like the frame batches
        all_probs = list(tp.map(lambda w: synth.ba_problem(K=20, P=22150, obs_per_point=4, F=F, seed=42 + 1000 * rank + w, views=args.ba_views), range(2 * n_ba)))
    prob_sets = [all_probs[:n_ba], all_probs[n_ba:]]
    probs = prob_sets[0]
    for p in all_probs:        # contiguous arrays of the C-ABI's types once, so that a window's creation is nothing but the cms_ba_create call
        p["poses"] = np.ascontiguousarray(p["poses"], np.float64); p["points"] = np.ascontiguousarray(p["points"], np.float64)
        p["e_obs"] = np.ascontiguousarray(p["e_obs"], np.float64)
    # the windows' observation-sized arrays in pinned host memory (cms_ba_window.flags = CMS_BA_INPUTS_PINNED): cms_ba_create_many then copies them to the device from
    # where they lie instead of through a staging memcpy -- a host assembles a window's observations (Optimizer.cpp:246-357) into buffers of its choice.
    # CMS_BENCH_PAGEABLE_WINDOWS=1: ordinary arrays, staged by the library (rounds 1-5)
    if os.environ.get("CMS_BENCH_PAGEABLE_WINDOWS", "") == "":
        all_probs = [api.pin_problem(p) for p in all_probs]
        prob_sets = [all_probs[:n_ba], all_probs[n_ba:]]
        probs = prob_sets[0]
    host = host_budget(world, local_rank, args.window_threads)      # the ranks of a node split its usable cores (and are pinned to their slice)
    host["blocking_sync"] = bool(blocking_sync)
    host["gpu_max_hw_queues"] = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
    # A rank short of cores lets the device plan its windows (cms_ba_window.flags |= CMS_BA_PLAN_ON_DEVICE: k_ba_plan_many, byte-identical device arrays): the plans are
    # half of a two-core rank's CPU time per step.  With cores to spare the host's plan stays: its windows become ready spread over the step and the Levenberg rounds
    # overlap the frame path better (profiles/r06_bench_runs.txt).  CMS_BENCH_PLAN_ON_DEVICE=1 / 0 overrides.
    plan_on_device = plans_on_device(host["thread_budget"])
    host["window_plans"] = "device (k_ba_plan_many)" if plan_on_device else "host threads"
    if plan_on_device:
        frac = float(os.environ.get("CMS_BENCH_PLAN_ON_DEVICE_FRACTION", "1"))      # developer knob: only this share of the windows (the flag is per window)
        for i, p in enumerate(all_probs):
            if (i % 8) < round(8 * frac):
                p["_plan_on_device"] = True
    n_wthreads = host["window_threads"]
    # threads of one cms_ba_create_many call (a group's 16 windows): every one of them sets its windows up on a stream of its own, and set-up streams compete with
    # the frame path and the Levenberg rounds for the hardware queues -- ONE thread per group keeps up (16 plans of ~0.35 ms per ~11.5-ms step) and measured best
    # (profiles/r06_bench_runs.txt: 1 / 2 / 3 / 4 threads per group = 23.0 / 22.2 / 22.1 / 21.7 k frames/s).  CMS_BENCH_CREATE_THREADS overrides.
    create_threads = max(1, int(os.environ.get("CMS_BENCH_CREATE_THREADS", "1")))
    host["create_threads_per_group"] = create_threads
    wpool = ThreadPoolExecutor(max_workers=n_wthreads)        # builds, reads back and destroys windows next to the running step
    group_stream = []           # one long-lived stream per window group (filled below): CreateNewMapPoints and the group's BA rounds
    cpu_acc = {"create": 0.0, "finish": 0.0, "n": 0}      # thread CPU seconds (developer knob CMS_BENCH_THREAD_CPU)
    def make_window(p, gi=-1):
        t1 = time.perf_counter(); c1 = time.thread_time()
        ba = api.BundleAdjuster(p, device=local_rank)      # uploads on a stream from the library's pool ...
        if gi >= 0 and not own_streams:
            ba.set_stream(group_stream[gi])                # ... and runs on its group's stream.  (Every window on a stream of its own meant ~100
        cpu_acc["create"] += time.thread_time() - c1; cpu_acc["n"] += 1
        return ba, 1e3 * (time.perf_counter() - t1)        # streams on 8 hardware queues: the two groups' chains and CreateNewMapPoints queued behind each other)
    own_streams = os.environ.get("CMS_BENCH_WINDOW_STREAMS", "") != ""      # developer knob: the old behaviour
    # Round 6: a window GROUP's set-up and read-back are one library call each (cms_ba_create_many: host parts on a few threads of the call, one expansion
    # launch per eight windows; cms_ba_read_many: one gather launch), and the group's 16 x 20 poses go back to the store in one cms_kfstore_update_poses
    # -- 2 x 3 pool tasks and ~8 small launches per step instead of 32 + 32 tasks and ~92 launches.  CMS_BENCH_PER_WINDOW_CALLS=1: round 5's calls (A/B)
    per_window_calls = os.environ.get("CMS_BENCH_PER_WINDOW_CALLS", "") != ""
    per_window_create = per_window_calls or os.environ.get("CMS_BENCH_PER_WINDOW_CREATE", "") != ""      # (each half on its own, A/B)
    per_window_finish = per_window_calls or os.environ.get("CMS_BENCH_PER_WINDOW_FINISH", "") != ""
    win_arrays = {}
    def make_group(j, gi, ids):
        t1 = time.perf_counter(); c1 = time.thread_time()
        mine = [prob_sets[j][w] for w in ids]
        key = tuple(id(p_) for p_ in mine)         # (the random-views pass swaps the problem sets: the descriptions belong to the problem OBJECTS)
        if key not in win_arrays:
            win_arrays[key] = (api.ba_window_array(mine), mine)
        det_points = api.ba_get_deterministic() and os.environ.get("CMS_BA_DET_POINTS", "") != ""      # (round 5's deterministic windows: the host's full plan, 4 ms each -- the window threads' share)
        grp = api.ba_create_many(mine, device=local_rank, threads=create_threads if not det_points else max(1, n_wthreads // max(1, n_grp)),
                                 windows=win_arrays[key][0])
        if not own_streams:
            for ba in grp:
                ba.set_stream(group_stream[gi])
        cpu_acc["create"] += time.thread_time() - c1; cpu_acc["n"] += len(grp)
        ms = 1e3 * (time.perf_counter() - t1) / max(len(grp), 1)
        return [(ba, ms) for ba in grp]
    def finish_group(grp, gi):
        c1 = time.thread_time()
        outs = api.ba_read_many(grp)
        if life.get("mapping_full"):
            for wi in range(len(grp)):
                write_back(gi, wi)
        for ba in grp:
            ba.close()
        cpu_acc["finish"] += time.thread_time() - c1
        return outs
    def finish_window(ba, gw=None):
        c1 = time.thread_time()
        out = ba.read()             # poses, points, outlier flags -> host arrays (Optimizer.cpp:419-450)
        if gw is not None and life.get("mapping_full"):
            write_back(*gw)         # the window's key frames' poses back into the store (Optimizer.cpp:419-431; see mapping_side)
        ba.close()                  # cms_ba_destroy
        cpu_acc["finish"] += time.thread_time() - c1
        return out
    # warm the per-device pools (streams, slabs, pinned blocks) to the steady state's high-water mark: the running step's windows, the two sets under
    # construction and the set being read back are alive at once -- growing a pool inside a timed step costs a device allocation (a 20-25 ms step)
    for ba, _ in [f.result() for f in [wpool.submit(make_window, p) for _ in range(4) for p in probs]]:
        ba.close()
    t_setup = time.perf_counter()
    bas = [api.BundleAdjuster(p, device=local_rank) for p in probs]
    ba_setup_ms = 1e3 * (time.perf_counter() - t_setup) / max(len(bas), 1)      # cms_ba_create per window, one after the other, pools warm
    # tracking's pose-only optimisation (Optimizer::PoseOptimization, once per frame here; the reference calls it 1-3 times):
    # one problem per frame, ~600 matched map points with 10 % mismatches, resident on the device, one launch per step
    pose_probs = [synth.pose_problem(N=args.pose_edges, F=F, seed=1000 * rank + b, outlier_frac=0.1) for b in range(B)]
    po = api.PoseOptimizer(B, sum(len(p["Xw"]) for p in pose_probs), device=local_rank)
    os.environ.pop("CMS_CTX_CU_MASK", None)
    po.upload(pose_probs)

    # the windows are split into `--ba-groups` groups, each advanced in lock-step by its own host thread on its own stream
    n_grp = max(1, min(args.ba_groups, n_ba))
    groups = [bas[gi::n_grp] for gi in range(n_grp)]            # the standing windows of the optimise-only pass
    group_ids = [list(range(n_ba))[gi::n_grp] for gi in range(n_grp)]

    # LocalMapping::CreateNewMapPoints in front of every window's BA: the window's key frame against its 20 best covisible neighbours
    # (~1650 features each, FeatureVectors of ~400 nodes), key frames resident on the device; one store + context (stream) per group
    tri_nn = 20
    tri_sets = [synth.keyframe_set(F, n_kf=tri_nn + 1, n_pts=2400, seed=300 + 10 * rank + w) for w in range(min(n_ba, 2))]
    for S in tri_sets:
        for k in S["kfs"]:           # mvKeyRays: the pixel's ray with unit depth on its face (CamModelGeneral::TransformCubemapToRays)
            _, r = synth.pixel_to_ray(F, k["x"].astype(np.float64), k["y"].astype(np.float64))
            k["rays"] = r.astype(np.float32)
    tri_ctx, tri_store, tri_jobs = [], [], []
    kf_max_features = max(2048, (max(len(k) for k in kps) + 255) // 256 * 256)
    # CreateNewMapPoints of a window group's key frames runs on a host thread and a queue of its own (tri_pool below): the group's local BA waits
    # for it (LocalMapping's order for a key frame: LocalMapping.cpp:80-110), but the NEXT step's CreateNewMapPoints -- other camera streams' key
    # frames -- no longer queues behind this step's Levenberg rounds (tri_pools below).  CMS_BENCH_TRI_INLINE=1: called by the BA worker itself, on the group's queue
    tri_inline = os.environ.get("CMS_BENCH_TRI_INLINE", "") != ""
    split_tri = os.environ.get("CMS_BENCH_SPLIT_TRI_STREAM", "") != "" or not tri_inline; ba_streams = []
    for gi, grp in enumerate(groups):
        mprio = os.environ.get("CMS_BENCH_MAP_PRIORITY", "")       # developer knob: the same for the mapping side's queues (one per window group)
        if mprio:
            os.environ["CMS_FRAME_STREAM_PRIORITY"] = mprio
        if 0 < cu_split < 256:
            os.environ["CMS_CTX_CU_MASK"] = cu_mask_env(cu_split, 256)
        elif os.environ.get("CMS_BENCH_CU_GROUPS", "") != "":      # developer experiment: every window group on its own share of the chip, the frame path everywhere
            per = 256 // max(1, n_grp)
            os.environ["CMS_CTX_CU_MASK"] = cu_mask_env(gi * per, (gi + 1) * per)
        cg = api.Context(camd, nfeatures=nfeat, max_batch=1, device=local_rank)
        os.environ.pop("CMS_CTX_CU_MASK", None)
        if mprio:
            del os.environ["CMS_FRAME_STREAM_PRIORITY"]
        # one more slot per window: where the key frame made from this step's frame lands (cms_kfstore_put_from_frame, see mapping_side below)
        st = api.KeyframeStore(cg, max_keyframes=len(grp) * (tri_nn + 2), max_features=kf_max_features, max_nodes=1024)
        jobs_g = []
        for wi in range(len(grp)):
            S = tri_sets[(gi + wi) % len(tri_sets)]
            base = wi * (tri_nn + 1)
            for i, k in enumerate(S["kfs"]):
                K, keep = api.make_keyframe(k)
                st.put(base + i, K)
            jobs_g.append((base, list(range(base + 1, base + tri_nn + 1))))
        tri_ctx.append(cg); tri_store.append(st); tri_jobs.append(jobs_g)
        if split_tri:          # developer knob: the group's BA rounds on a stream of their own, CreateNewMapPoints alone on the context's
            # (CMS_BENCH_BA_PRIORITY=high: the rounds' queue in the high-priority class -- hardware queues of its own, not shared with the
            # window pool's upload streams, whose multi-megabyte transfers otherwise sit in front of the chain's launches)
            bprio = {"high": -1, "low": 1}.get(os.environ.get("CMS_BENCH_BA_PRIORITY", ""), 0)
            bs = torch.cuda.Stream(device=dev, priority=bprio); ba_streams.append(bs); group_stream.append(bs.cuda_stream)
        else:
            group_stream.append(cg.stream)
        if os.environ.get("CMS_BENCH_SHARED_STREAMS", "") != "":
            for ba in grp:                   # developer knob: the whole mapping side of a group on ONE stream (cms_ba_set_stream).  Measured
                ba.set_stream(cg.stream)     # slower (15.9 against 14.4 ms per step): the resets and CreateNewMapPoints then queue behind the group's BA

    # ---- the rest of LocalMapping's per-key-frame sequence (LocalMapping::Run, LocalMapping.cpp:52-117, 388-466), inside the timed region since round 5:
    #   ProcessNewKeyFrame   the key frame made from this step's frame enters the store: cms_kfstore_put_from_frame -- key points, descriptors, rays and
    #                        the frame grid device to device out of the frame context, FeatureVector + map-point slots from a pinned block; enqueued by
    #                        the frame path's thread behind the frame's tracking, like the KeyFrame constructor on the Tracking thread
    #   CreateNewMapPoints   as before (against 20 resident neighbours)
    #   SearchInNeighbors    both Fuse directions as jobs of ONE cms_kfstore_fuse_search_sets call per window group: the key frame's map points into each
    #                        of its 20 neighbours, the neighbours' map points (union) into the key frame; map points already seen by the target skipped
    #   LocalBundleAdjustment as before
    #   pose write-back      cms_kfstore_update_poses for the window's 20 key frames (Optimizer.cpp:419-431), after the window's read-back
    # The synthetic streams are not one consistent scene (frames are drifting textures, the CreateNewMapPoints scenes and the BA windows are separate
    # synthetic problems), so the inserted key frame lands in a slot of its own next to the scene's key frames, and the write-back writes the scene's
    # key frames' own poses: every call's COST is in the step, CreateNewMapPoints' input geometry stays consistent.  CMS_BENCH_MAPPING_MINIMAL=1 (and
    # config.mapping_side.minimal): CreateNewMapPoints + local BA only, round 4's step.
    mapping_full_default = os.environ.get("CMS_BENCH_MAPPING_MINIMAL", "") == ""
    import ctypes as C_
    L_ = api.lib()
    def kf_info_of(S, b):
        k, d = S.fetched[b]
        n = len(k)
        node = (d[:, 0].astype(np.int32) * 4 + (d[:, 1] >> 6)) % 1024           # FeatureVector stand-in: features binned by descriptor bits (KeyFrame::ComputeBoW is DBoW2 on the host)
        order = np.lexsort((np.arange(n), node)).astype(np.int32)
        ids, starts = np.unique(node[order], return_index=True)
        pose = S.lm[b]["pose15"]
        a = dict(n=n, R=np.ascontiguousarray(pose[:9], np.float32), t=np.ascontiguousarray(pose[9:12], np.float32), Ow=np.ascontiguousarray(pose[12:15], np.float32),
                 mp=np.where(np.arange(n) % 3 == 0, -1, np.arange(n)).astype(np.int32), nid=ids.astype(np.int32),
                 noff=np.concatenate([starts, [n]]).astype(np.int32), nfeat=order)
        a["b"] = b
        return a
    kf_frame = [min(B - 1, w * args.ba_every + args.ba_every - 1) for w in range(n_ba)]      # the frame of every window's stream that becomes its key frame
    kf_infos = [[kf_info_of(S, kf_frame[w]) for w in range(n_ba)] for S in sets]
    class KfFromFrame(C_.Structure):      # cms_kf_from_frame (include/cubemapslam_hip.h)
        _fields_ = [("slot", C_.c_int), ("b", C_.c_int), ("n", C_.c_int), ("Rcw", C_.c_void_p), ("tcw", C_.c_void_p), ("Ow", C_.c_void_p), ("median_depth", C_.c_float),
                    ("mp", C_.c_void_p), ("nnodes", C_.c_int), ("node_id", C_.c_void_p), ("node_off", C_.c_void_p), ("node_feat", C_.c_void_p)]
    L_.cms_kfstore_put_from_frames.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_int, C_.c_void_p]
    put_items = []                                      # [set][group] -> array of items: one library call per window group and step (a Python loop of 32
    for j_ in range(len(sets)):                         # calls re-took the interpreter lock 64 times next to 20 busy threads: 6 ms per step)
        per_g = []
        for gi, ids in enumerate(group_ids):
            arr = (KfFromFrame * len(ids))()
            for wi, w in enumerate(ids):
                a = kf_infos[j_][w]
                q = arr[wi]
                q.slot = len(ids) * (tri_nn + 1) + wi; q.b = a["b"]; q.n = a["n"]; q.Rcw = a["R"].ctypes.data; q.tcw = a["t"].ctypes.data; q.Ow = a["Ow"].ctypes.data
                q.median_depth = 2.0; q.mp = a["mp"].ctypes.data; q.nnodes = len(a["nid"]); q.node_id = a["nid"].ctypes.data; q.node_off = a["noff"].ctypes.data
                q.node_feat = a["nfeat"].ctypes.data
            per_g.append(arr)
        put_items.append(per_g)
    def put_keyframes(set_idx):
        """ProcessNewKeyFrame for the n_ba key frames of this step's batch: behind the batch's tracking on the frame path's stream"""
        for gi, ids in enumerate(group_ids):
            rc = L_.cms_kfstore_put_from_frames(tri_store[gi].h, ctx.h, len(ids), put_items[set_idx][gi])
            if rc < 0:
                raise RuntimeError("cms_kfstore_put_from_frames: %s" % L_.cms_last_error().decode())
    def fuse_jobs_of(S, base):
        """SearchInNeighbors for the scene's current key frame (slot `base`) and its neighbours (base + 1 ...): (slot, map points) jobs"""
        kfs, X, sf = S["kfs"], S["X"], S["scale_factors"]
        def mps(kf):
            sel = np.flatnonzero(kf["mp"] >= 0)
            pid = kf["point"][sel]
            pos = X[pid].astype(np.float32)
            v = pos - np.asarray(kf["Ow"], np.float32)[None, :]
            dist = np.linalg.norm(v, axis=1).astype(np.float32)
            maxd = (dist * sf[kf["octave"][sel]]).astype(np.float32)            # MapPoint::UpdateNormalAndDepth (MapPoint.cpp:332-373)
            return pid, pos, (v / dist[:, None]).astype(np.float32), (maxd / sf[-1]).astype(np.float32), maxd, kf["desc"][sel]
        cur = mps(kfs[0])
        have = [set(k["point"][k["mp"] >= 0].tolist()) for k in kfs]
        jobs = []
        for i in range(1, len(kfs)):                                           # the key frame's map points into neighbour i
            skip = np.fromiter((p in have[i] for p in cur[0]), np.uint8, len(cur[0]))
            jobs.append((base + i, skip) + cur[1:])
        seen, parts = set(), []
        for i in range(1, len(kfs)):                                           # the neighbours' map points (union) into the key frame
            m = mps(kfs[i])
            new = np.array([p not in seen for p in m[0]], bool)
            seen.update(m[0][new].tolist())
            parts.append(tuple(a[new] for a in m))
        u = tuple(np.concatenate([q[j] for q in parts]) for j in range(6))
        jobs.append((base, np.fromiter((p in have[0] for p in u[0]), np.uint8, len(u[0]))) + u[1:])
        return jobs
    fuse_prep = []
    L_.cms_kfstore_fuse_search_sets.argtypes = [C_.c_void_p, C_.c_int] + [C_.c_void_p] * 6 + [C_.c_int] + [C_.c_void_p] * 3 + [C_.c_float, C_.c_void_p, C_.c_void_p]
    def pin(arr):                                                              # pinned host copies: the call's uploads are asynchronous DMA transfers
        arr = np.ascontiguousarray(arr)
        pa = api.PinnedArray((max(arr.nbytes, 16),))
        v = pa.array[:arr.nbytes].view(arr.dtype).reshape(arr.shape); v[...] = arr
        return pa, v
    for gi, ids in enumerate(group_ids):
        # per window two SETS of map points (the key frame's; the union of its neighbours') uploaded once each, and 21 jobs that refer to them
        # (cms_kfstore_fuse_search_sets): the key frame's set goes to 20 neighbours
        sets_g, jobs_g2 = [], []
        for wi in range(len(ids)):
            jl = fuse_jobs_of(tri_sets[(gi + wi) % len(tri_sets)], wi * (tri_nn + 1))
            sets_g.append(jl[0][2:]); sets_g.append(jl[-1][2:])                   # (pos, normal, min, max, desc) of the key frame's points / of the neighbours' union
            for j in jl[:-1]:
                jobs_g2.append((j[0], 2 * wi, j[1]))
            jobs_g2.append((jl[-1][0], 2 * wi + 1, jl[-1][1]))
        set_off = np.concatenate([[0], np.cumsum([len(q[0]) for q in sets_g])]).astype(np.int32)
        cols = [pin(np.concatenate([q[c] for q in sets_g])) for c in range(5)]    # pos, normal, min, max, desc per SET point
        slots = np.array([j[0] for j in jobs_g2], np.int32); jset = np.array([j[1] for j in jobs_g2], np.int32)
        pskip, skip = pin(np.concatenate([j[2] for j in jobs_g2]).astype(np.uint8))
        n_mp = len(skip)
        pbi, bi = pin(np.zeros(n_mp, np.int32)); pbd, bd = pin(np.zeros(n_mp, np.int32))
        fuse_prep.append(dict(keep=(cols, slots, jset, set_off, pskip, pbi, pbd), njobs=len(jobs_g2), n_mp=n_mp, n_set_points=int(set_off[-1]), best_idx=bi,
                              args=(len(sets_g), api._p(set_off)) + tuple(api._p(v) for _, v in cols) + (len(jobs_g2), api._p(slots), api._p(jset), api._p(skip),
                                                                                                          C_.c_float(3.0), api._p(bi), api._p(bd))))
    store_lock = [threading.Lock() for _ in range(n_grp)]         # a store's calls one at a time (its mapping thread; the optimise-only pass's worker)
    wb_queue = [collections.deque() for _ in range(n_grp)]        # pose write-backs of read-back windows, applied by the group's mapping thread before its next key frames
    upd_prep = []
    for gi, ids in enumerate(group_ids):
        per_w = []
        for wi in range(len(ids)):
            S = tri_sets[(gi + wi) % len(tri_sets)]
            sl = np.arange(wi * (tri_nn + 1), wi * (tri_nn + 1) + 20, dtype=np.int32)
            R = np.ascontiguousarray(np.stack([np.asarray(k["R"], np.float32).reshape(9) for k in S["kfs"][:20]])); t = np.ascontiguousarray(np.stack([k["t"] for k in S["kfs"][:20]]), np.float32)
            Ow = np.ascontiguousarray(np.stack([k["Ow"] for k in S["kfs"][:20]]), np.float32)
            per_w.append(dict(keep=(sl, R, t, Ow), args=(20, api._p(sl), api._p(R), api._p(t), api._p(Ow))))
        upd_prep.append(per_w)
    upd_all = []                                                  # a whole step's windows of a group in one call (what the queue usually holds)
    for gi in range(n_grp):
        ks = [q["keep"] for q in upd_prep[gi]]
        sl = np.ascontiguousarray(np.concatenate([k_[0] for k_ in ks])); R = np.ascontiguousarray(np.concatenate([k_[1] for k_ in ks]))
        t = np.ascontiguousarray(np.concatenate([k_[2] for k_ in ks])); Ow = np.ascontiguousarray(np.concatenate([k_[3] for k_ in ks]))
        upd_all.append(dict(keep=(sl, R, t, Ow), n=len(ks), args=(len(sl), api._p(sl), api._p(R), api._p(t), api._p(Ow))))
    map_acc = {"fuse_ms": 0.0, "fuse_n": 0, "fused": 0, "put_ms": 0.0, "put_n": 0, "upd_ms": 0.0, "upd_n": 0}
    def fuse_group(gi):
        t0_ = time.perf_counter()
        with store_lock[gi]:
            rc = L_.cms_kfstore_fuse_search_sets(tri_store[gi].h, *fuse_prep[gi]["args"])
        if rc < 0:
            raise RuntimeError("cms_kfstore_fuse_search_sets: %s" % L_.cms_last_error().decode())
        map_acc["fuse_ms"] += 1e3 * (time.perf_counter() - t0_); map_acc["fuse_n"] += 1
        map_acc["fused"] = int((fuse_prep[gi]["best_idx"] >= 0).sum())
    def write_back(gi, wi):
        """a window was read back (pool thread): its key frames' poses go to the store with the mapping thread's next turn, like LocalMapping applies a
        local BA's result before it takes the next key frame"""
        wb_queue[gi].append(wi)
    def apply_write_backs(gi):
        while len(wb_queue[gi]) >= upd_all[gi]["n"] and sorted(list(wb_queue[gi])[:upd_all[gi]["n"]]) == list(range(upd_all[gi]["n"])):
            for _ in range(upd_all[gi]["n"]):                      # every window of the previous step was read back: their 16 x 20 poses in one call
                wb_queue[gi].popleft()
            t0_ = time.perf_counter()
            rc = L_.cms_kfstore_update_poses(tri_store[gi].h, *upd_all[gi]["args"])
            if rc < 0:
                raise RuntimeError("cms_kfstore_update_poses: %s" % L_.cms_last_error().decode())
            map_acc["upd_ms"] += 1e3 * (time.perf_counter() - t0_); map_acc["upd_n"] += upd_all[gi]["n"]
        while wb_queue[gi]:
            wi = wb_queue[gi].popleft()
            t0_ = time.perf_counter()
            rc = L_.cms_kfstore_update_poses(tri_store[gi].h, *upd_prep[gi][wi]["args"])
            if rc < 0:
                raise RuntimeError("cms_kfstore_update_poses: %s" % L_.cms_last_error().decode())
            map_acc["upd_ms"] += 1e3 * (time.perf_counter() - t0_); map_acc["upd_n"] += 1

    schur_acc = {"ms": 0.0, "n": 0}
    worker_ms = {}
    # steps overlap by one: the mapping side of step s (CreateNewMapPoints + local BA of its 32 windows) is waited for at the end of step s + 1, so it
    # runs next to step s + 1's frame path the way LocalMapping runs next to Tracking.  CMS_BENCH_NO_PIPELINE=1 (and config.unpipelined): every step
    # waits for its own mapping side, round 3's loop
    pipeline_default = os.environ.get("CMS_BENCH_NO_PIPELINE", "") == ""
    step_trace = [] if os.environ.get("CMS_BENCH_STEP_TRACE", "") != "" else None      # developer knob: host time stamps of the step loop (stderr at exit)
    stagger = os.environ.get("CMS_BENCH_STAGGER", "") != ""        # developer knob: odd groups run their BA first and CreateNewMapPoints (of their NEXT step's key frames) after it
    tri_last = os.environ.get("CMS_BENCH_TRI_LAST", "") != ""      # developer knob: CreateNewMapPoints behind the group's BA instead of in front of it
    def ba_worker(grp, gi, keep):
        """optimise-only pass: one group of STANDING windows of one step; returns (elapsed ms, new map points, per-window stats)"""
        t_ba0 = time.perf_counter()
        res = tri_store[gi].create_new_map_points(tri_jobs[gi], copy=False)
        _, stats = api.ba_optimize_many(grp, (5, 10))   # the group's windows share every launch (kb_ba_* kernels, one window per blockIdx.z)
        if not keep:
            for ba in grp:                       # benchmark plumbing: put the initial estimate back for the next step (asynchronous; done
                ba.reset()                       # here, where the chip is quiet, rather than in front of the next step's first kernel)
        return 1e3 * (time.perf_counter() - t_ba0), sum(len(r[0]) for r in res), stats, None, None

    tri_pools = [ThreadPoolExecutor(max_workers=1) for _ in range(n_grp)]      # one thread per group: a store's calls stay in order
    def ba_worker_life(futs, gi, keep, tri_fut=None):
        """one window group of one step with the windows' whole life cycle: the group's windows were built by the pool while the previous
        step ran (futs); here they are optimised, then handed back to the pool to be read back and destroyed.  Returns (elapsed ms, new map
        points, per-window stats, futures of the read-backs, the windows' creation times)"""
        t_ba0 = time.perf_counter()
        made = []
        for f in futs:
            r_ = f.result()
            made.extend(r_ if isinstance(r_, list) else [r_])
        t_w = time.perf_counter()
        grp = [m[0] for m in made]
        grp[0].profile_kernel(3)          # HIP events around the Schur kernel of every round (the BA chain's largest kernel)
        last_here = tri_last or (stagger and gi % 2 == 1)
        t_p = time.perf_counter()
        if tri_fut is not None:
            res = tri_fut.result()        # (submitted when the step began; usually through long before this group's previous windows were)
        elif not last_here:
            res = tri_store[gi].create_new_map_points(tri_jobs[gi], copy=False)
        t_t = time.perf_counter()
        worker_ms["profile_arm"] = worker_ms.get("profile_arm", 0.0) + 1e3 * (t_p - t_w)
        worker_ms["create_new_map_points_library_call"] = worker_ms.get("create_new_map_points_library_call", 0.0) + getattr(tri_store[gi], "last_call_ms", 0.0)
        _, stats = api.ba_optimize_many(grp, (5, 10))
        t_o = time.perf_counter()
        if last_here and tri_fut is None:
            res = tri_store[gi].create_new_map_points(tri_jobs[gi], copy=False)
        ms, nl = grp[0].profile_get()
        schur_acc["ms"] += ms; schur_acc["n"] += nl
        outs = [wpool.submit(finish_window, ba, (gi, wi)) for wi, ba in enumerate(grp)] if per_window_finish else [wpool.submit(finish_group, grp, gi)]
        if step_trace is not None:
            step_trace.append(("worker %d" % gi, t_ba0, t_w, t_t, t_o, time.perf_counter()))
        for k_, v_ in (("wait_for_windows", t_w - t_ba0), ("create_new_map_points", t_t - t_w), ("optimize_many", t_o - t_t), ("hand_over", time.perf_counter() - t_o)):
            worker_ms[k_] = worker_ms.get(k_, 0.0) + 1e3 * v_
        worker_ms["n"] = worker_ms.get("n", 0) + 1
        return 1e3 * (time.perf_counter() - t_ba0), sum(len(r[0]) for r in res), stats, outs, [m[1] for m in made]

    part = os.environ.get("CMS_BENCH_PART", "")      # developer knob: "ba" / "frames" times one half of the step alone (not a bench line)
    part_env = part
    pool = ThreadPoolExecutor(max_workers=n_grp)       # one standing host thread per window group (LocalMapping-like)
    last = {"traj": None, "ba_stats": None, "tri_new": 0, "ba_out": None, "set": 0}
    acc = {"ba_ms": 0.0, "ba_n": 0, "create_ms": 0.0, "create_n": 0}
    life = {"on": True, "queue": collections.deque(), "reads": []}
    ahead = max(1, int(os.environ.get("CMS_BENCH_WINDOWS_AHEAD", "2")))      # sets of windows under construction in front of the running step
    # developer knob: the pool starts window w of a set w / 32 x spread_ms late, so that the windows' uploads and gather / reset kernels are spread
    # over the step instead of landing together on its first milliseconds, the frame path's.  Measured with 6 ms: extractor inside the step 0.38
    # against 0.33-0.35 of 8 TB/s on SURVEY's bytes, the step the same within noise -- and the Schur kernel of the Levenberg rounds, which the
    # spread-out uploads then overlap instead, 160-180 against 145-160 us per launch.  Off by default: the chain's kernels are the step's critical path.
    spread_ms = float(os.environ.get("CMS_BENCH_SPREAD_MS", "0"))
    def make_window_at(delay, p, gi):
        if delay > 0:
            time.sleep(delay)
        return make_window(p, gi)

    def submit_windows(j):
        """the pool starts building the n_ba windows of problem set j; returned per group"""
        if per_window_create:
            life["queue"].append((j, [[wpool.submit(make_window_at, 1e-3 * spread_ms * w / max(n_ba, 1), prob_sets[j][w], gi) for w in ids] for gi, ids in enumerate(group_ids)]))
        else:
            life["queue"].append((j, [[wpool.submit(make_group, j, gi, ids)] for gi, ids in enumerate(group_ids)]))

    # developer knob: hand the next set of windows to the pool only after the step's frame path has been waited for (the pool's uploads and
    # gather / reset kernels then stay off the extraction kernels -- and land on the Levenberg rounds instead: 14.8-15.4 against 14.4-14.7 ms)
    late_submit = os.environ.get("CMS_BENCH_LATE_SUBMIT", "") != ""
    def next_windows():
        """the oldest set of windows under construction (this step's), and one more set submitted in its place: `ahead` sets are always in the
        pool's hands, so a host hiccup of a step's length does not reach the critical path"""
        cur_set, cur = life["queue"].popleft()
        nxt_set = life["queue"][-1][0] ^ 1 if life["queue"] else cur_set ^ 1
        if late_submit:
            life["deferred"] = nxt_set          # ... submitted once this step's frame path is through (see step)
        else:
            submit_windows(nxt_set)
        return cur, cur_set

    # developer knob: queue the mapping side of a step behind the step's extraction (cms_stream_wait_extracted) instead of letting the two
    # overlap.  Measured: the extractor then runs at 35 % instead of 33 % of its byte roofline inside the step (43 % with no local BA in
    # the step at all: the FP64 chain also pulls the clocks down), and the step gets 4 % longer -- overlap stays the default
    serial = os.environ.get("CMS_BENCH_SERIAL_EXTRACT", "") != ""
    ba_first = os.environ.get("CMS_BENCH_BA_FIRST", "")     # developer knob: hand the mapping side to its threads BEFORE the frame path is enqueued (value = head start in us)
    def timed_tri(gi):
        t0_ = time.perf_counter()
        with store_lock[gi]:
            if life.get("mapping_full"):
                apply_write_backs(gi)   # poses of the windows read back since this thread's last turn (Optimizer.cpp:419-431)
            r = tri_store[gi].create_new_map_points(tri_jobs[gi], copy=True)      # (copies: the store's buffers serve the next call while this result waits)
        worker_ms["create_new_map_points_own_thread"] = worker_ms.get("create_new_map_points_own_thread", 0.0) + 1e3 * (time.perf_counter() - t0_)
        if life.get("mapping_full"):
            fuse_group(gi)          # SearchInNeighbors: both Fuse directions of the group's key frames, one call (the group's local BA waits for it too)
        return r
    def submit_tri(gi):
        return None if tri_inline else tri_pools[gi].submit(timed_tri, gi)

    def collect(ths, keep):
        """wait for a step's window groups (raises what a worker raised) and book their results"""
        res = [th.result() for th in ths]
        if step_trace is not None:
            step_trace.append(("workers done", time.perf_counter()))
        if res:
            acc["ba_ms"] += sum(r[0] for r in res) / len(res); acc["ba_n"] += 1
            last["tri_new"] = sum(r[1] for r in res)
            last["ba_stats"] = [r[2] for r in res]
            if res[0][3] is not None:
                # the read-backs of these windows finish under the next step; at most two steps' worth are ever outstanding
                for f in life["reads"]:
                    f.result()
                if step_trace is not None:
                    step_trace.append(("earlier read-backs waited for", time.perf_counter()))
                life["reads"] = [f for r in res for f in r[3]]
                if keep:
                    last["ba_out"] = [[o_ for f in r[3] for o_ in (f.result() if isinstance(f.result(), list) else [f.result()])] for r in res]
                for r in res:
                    acc["create_ms"] += sum(r[4]); acc["create_n"] += len(r[4])

    def collect_inflight():
        prev = life.pop("inflight", None)
        if prev is not None:
            collect(*prev)

    def step(i, streaming, keep=False):
        part = life.get("part", part_env)             # (the extract-only pass sets "frames" for its steps)
        if step_trace is not None:
            step_trace.append(("step %d begins" % i, time.perf_counter()))
        S = sets[i % 2]
        ths = []
        if ba_first and part != "frames" and life["on"]:
            cur, cur_set = next_windows()
            ths = [pool.submit(ba_worker_life, cur[gi], gi, keep, submit_tri(gi)) for gi in range(n_grp)]
            last["set"] = cur_set
            if int(ba_first) > 0:
                t_hs = time.perf_counter()
                while time.perf_counter() - t_hs < 1e-6 * int(ba_first):
                    pass
        if part != "ba":
            po.launch()                 # own stream, overlaps the frame path
            if not streaming:
                ctx.upload_device(S.d_frames.data_ptr(), B)        # inputs resident in HBM: staging <- device copy
            ctx.process(B, True)        # (streaming: waits on the device for the copy enqueued during the previous step)
        if step_trace is not None:
            step_trace.append(("frame path enqueued", time.perf_counter()))
        if streaming:
            ctx.upload_async(sets[(i + 1) % 2].pinned.array)   # next step's frames travel under this step's kernels
        if part != "frames":
            if serial and part != "ba":
                # extraction and the local-BA chain each fill the chip; side by side they only slow each other down.  The mapping side of the
                # step (CreateNewMapPoints + local BA of every window group) waits on the device for the extraction and overlaps the
                # tracking kernels (grids, projection searches, pose optimisation) instead
                for gi, grp in enumerate(groups):
                    ctx.stream_wait_extracted(tri_ctx[gi].stream)
                    ctx.stream_wait_extracted(grp[0].stream)
            if ths:
                pass
            elif life["on"]:
                cur, cur_set = next_windows()             # the coming steps' windows are built under this one
                ths = [pool.submit(ba_worker_life, cur[gi], gi, keep, submit_tri(gi)) for gi in range(n_grp)]
                last["set"] = cur_set
                if step_trace is not None:
                    step_trace.append(("windows + workers submitted", time.perf_counter()))
            else:
                ths = [pool.submit(ba_worker, grp, gi, keep) for gi, grp in enumerate(groups)]
        if part != "ba":
            S.enqueue_tracking(ext_stream)
            if life["on"] and life.get("mapping_full"):
                t_put = time.perf_counter()
                put_keyframes(i % 2)        # ProcessNewKeyFrame: this batch's key frames enter the stores (device to device, behind the tracking just enqueued)
                map_acc["put_ms"] += 1e3 * (time.perf_counter() - t_put); map_acc["put_n"] += 1
        if step_trace is not None:
            step_trace.append(("tracking enqueued", time.perf_counter()))
        frame_poses = None
        if part != "ba":
            # the step's frames are THROUGH when their key points, matches and optimised poses are on the host side of the boundary: wait for the frame
            # path and fetch the poses, every step (Tracking hands a pose back per frame).  (From the middle of round 4 to the middle of round 5 these
            # two calls sat under the developer trace's condition by an editing mistake: the frame path was only waited for at the end of the timed
            # region and the multi-GPU gather below had no poses to send.  All of the step's GPU work was inside the timed region either way --
            # barrier() synchronises the device -- but the per-step wait and the pose fetch were not.  Measured with the calls back in place: the same
            # frames/s within the run-to-run spread (the step is bound by the window groups' chain, the frame path's host thread has slack);
            # tests/test_bench_step_cpu.py guards the structure, DESIGN.md section 4 has the numbers.)
            ctx.sync()
            if step_trace is not None:
                step_trace.append(("ctx.sync returned", time.perf_counter()))
            _, frame_poses, _, _ = po.fetch()
        if life.get("deferred") is not None:
            submit_windows(life.pop("deferred"))
        if step_trace is not None:
            step_trace.append(("frame path waited for", time.perf_counter()))
        # The mapping side of step s is waited for at the end of step s + 1 (life["pipeline"]): like the reference's LocalMapping thread next to
        # Tracking (System.cpp:108-127), the local BA of one batch's key frames runs while the next batch is tracked.  The window groups' host
        # threads take step s + 1's windows as soon as step s's are through, so the Levenberg chain never waits for the frame path's host side.
        # Every step's work still happens inside the timed region: timed() collects the last step's mapping side before it stops the clock.
        if life["on"] and life.get("pipeline") and not keep:
            prev = life.pop("inflight", None)
            life["inflight"] = (ths, keep)
            if prev is not None:
                collect(*prev)
        else:
            collect_inflight()
            collect(ths, keep)
        if part != "ba" and (world > 1 or args.force_gather):   # trajectory assembly on rank 0 over RCCL (72 B / frame, latency only)
            recs = [cdist.make_records(my_streams[s], i * fps + np.arange(fps), frame_poses[s * fps:(s + 1) * fps]) for s in range(len(my_streams))]
            traj = cdist.gather_trajectory(np.concatenate(recs, 0), device=coll_dev, dst=0)
            if traj is not None:
                last["traj"] = traj

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        """everything the window pool still owes: read-backs of the last step, and the windows built for a step that will not run"""
        collect_inflight()
        for f in life["reads"]:
            f.result()
        life["reads"] = []
        while life["queue"]:
            for futs in life["queue"].popleft()[1]:
                for f in futs:
                    r_ = f.result()
                    for m_ in (r_ if isinstance(r_, list) else [r_]):
                        m_[0].close()

    step_times = [] if os.environ.get("CMS_BENCH_STEP_TIMES", "") != "" else None      # developer knob: host wall time of every timed step (stderr)
    def timed(streaming, lifecycle=True, steps=None, only="", pipeline=None, mapping_full=None):
        steps = args.steps if steps is None else steps
        part = only or part_env
        life["part"] = part
        life["mapping_full"] = (mapping_full_default if mapping_full is None else mapping_full) and lifecycle and not tri_inline
        for k_ in map_acc:
            map_acc[k_] = 0 if k_ == "fused" else 0.0
        life["on"] = lifecycle and part != "frames"
        life["pipeline"] = pipeline_default if pipeline is None else pipeline
        if streaming:
            ctx.upload_async(sets[0].pinned.array)
        if life["on"]:
            for a_ in range(ahead):
                submit_windows(a_ & 1)
        for i in range(args.warmup):
            step(i, streaming)
        collect_inflight()                     # (the warm-up's last mapping side ends before the clock starts)
        stage = {}
        if not life["on"]:
            for grp in groups:
                grp[0].profile_kernel(3)
        barrier()
        acc["ba_ms"], acc["ba_n"], acc["create_ms"], acc["create_n"] = 0.0, 0, 0.0, 0
        schur_acc["ms"], schur_acc["n"] = 0.0, 0
        worker_ms.clear()
        # The timed region holds, per step, the creation of one set of windows (the NEXT step's, by the pool), the optimisation of one set, and the
        # read-back + destruction of one set (the PREVIOUS step's finish under this one); the region ends only when the last step's read-backs are
        # through.  The windows the last step built for a step that never runs are destroyed after the clock stops (they were built inside it).
        # Python's cyclic garbage collector is held off for the timed steps (and run right before them): a generation-2 pass over this
        # process's ~10^6 objects took 45-70 ms in the middle of a 50-step run (one step of 84 ms among steps of 13-14: CMS_BENCH_STEP_TIMES=1);
        # reference counting still frees everything the steps allocate.  Bench plumbing, not product: the library is C++.
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        t_prev = t0
        for i in range(steps):
            step(args.warmup + i, streaming)
            if step_times is not None:
                t_now = time.perf_counter(); step_times.append(round(1e3 * (t_now - t_prev), 2)); t_prev = t_now
            for k, v in (ctx.profile_ms().items() if part != "ba" else ()):
                stage[k] = stage.get(k, 0.0) + v
        collect_inflight()                     # the last step's mapping side (pipelined steps)
        for f in life["reads"]:
            f.result()
        life["reads"] = []
        if life.get("mapping_full"):
            for gi_ in range(n_grp):           # the last read-backs' pose write-backs (no further mapping-thread turn would apply them)
                with store_lock[gi_]:
                    apply_write_backs(gi_)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        drain()
        if streaming:
            ctx.upload_wait()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        for k in stage:
            stage[k] /= max(steps, 1)
        schur = [(schur_acc["ms"], schur_acc["n"])] if life["on"] else [grp[0].profile_get() for grp in groups]
        life["part"] = part_env
        return dt, stage, acc["ba_ms"] / max(acc["ba_n"], 1), schur

    def thread_cpu():
        """CPU seconds (user + system) of every thread of this process by name: /proc/self/task/*/stat (developer knob CMS_BENCH_THREAD_CPU)"""
        out = {}
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                with open("/proc/self/task/%s/stat" % tid) as f:
                    st = f.read()
                name = st[st.index("(") + 1:st.rindex(")")]
                fld = st[st.rindex(")") + 2:].split()
                out[(tid, name)] = (int(fld[11]) + int(fld[12])) / tck
            except (OSError, ValueError):
                pass
        return out
    # ---- the same step loop WITHOUT the interpreter: cubemapslam_amd/host/batch_driver.cpp makes the very calls step() makes, in the same order and with the same
    # pipelining, from C++ threads (one frame thread; a mapping, a local-BA, a builder and a finisher thread per window group).  Everything it touches is what
    # this file prepared above.  It is the headline's driver when libcubemapslam_host.so has it (CMS_BENCH_PY_DRIVER=1: the Python loop); the Python loop's
    # figure is reported next to it (config.python_step_loop), and every other pass of this file still runs the Python loop.
    cpp = None
    if os.environ.get("CMS_BENCH_PY_DRIVER", "") == "" and not part and not tri_inline and not own_streams and not per_window_create and not per_window_finish:
        try:
            from cubemapslam_amd import build as _build
            HL = C_.CDLL(_build.HOST_LIB)
            HL.cbd_create.restype = C_.c_void_p; HL.cbd_last_error.restype = C_.c_char_p
            cpp = HL if hasattr(HL, "cbd_steps") else None
        except (OSError, AttributeError):
            cpp = None
    if cpp is not None:
        VP, CbdFrameSet, CbdGroup, STEP_DONE, CbdPlan, CbdStats = cbd_types()
        sz_ = (C_.c_int * 4)()
        cpp.cbd_sizes(sz_)
        if list(sz_) != [C_.sizeof(CbdFrameSet), C_.sizeof(CbdGroup), C_.sizeof(CbdPlan), C_.sizeof(CbdStats)]:
            raise RuntimeError("bench.py's descriptions of the batch driver's structures do not match libcubemapslam_host.so: %s" % list(sz_))
        ptr = lambda t: int(t.data_ptr())
        cpp_keep = []
        def make_plan(mapping_full):
            P_ = CbdPlan()
            P_.ctx = ctx.h.value; P_.po = po.h.value; P_.B = B; P_.device = local_rank; P_.ngroups = n_grp; P_.create_threads = create_threads
            P_.mapping_full = 1 if mapping_full else 0; P_.ahead = ahead; P_.n_pose_edges = int(po.off[-1])
            for j, S in enumerate(sets):
                Q = P_.sets[j]
                Q.d_frames = ptr(S.d_frames); Q.nq = S.nq; Q.d_mm_frame = ptr(S.d_mm_frame); Q.d_mm_pose = ptr(S.d_mm_pose); Q.d_mm_valid = ptr(S.d_mm_valid)
                Q.d_mm_xw = ptr(S.d_mm_xw); Q.d_mm_oct = ptr(S.d_mm_oct)
                for k_, t_ in enumerate(S.d_mm_q):
                    Q.d_mm_q[k_] = ptr(t_)
                Q.d_cnt = ptr(S.d_cnt); Q.d_off = ptr(S.d_off); Q.d_idx = ptr(S.d_idx); Q.cand_cap = S.cand_cap; Q.d_tot = ptr(S.d_tot)
                Q.d_mm_mpoff = ptr(S.d_mm_mpoff); Q.d_mm_desc = ptr(S.d_mm_desc); Q.d_mm_pd = ptr(S.d_mm_pd); Q.d_mm_match = ptr(S.d_mm_match); Q.d_mm_ang = ptr(S.d_mm_ang); Q.d_mm_n = ptr(S.d_mm_n)
                Q.n_mp = S.n_mp; Q.d_lm_frame = ptr(S.d_lm_frame); Q.d_lm_pose = ptr(S.d_lm_pose)
                for k_, t_ in enumerate(S.d_lm_in):
                    Q.d_lm_in[k_] = ptr(t_)
                Q.d_lm_vis = ptr(S.d_lm_vis)
                for k_, t_ in enumerate(S.d_lm_f):
                    Q.d_lm_f[k_] = ptr(t_)
                for k_, t_ in enumerate(S.d_lm_i):
                    Q.d_lm_i[k_] = ptr(t_)
                Q.d_lm_off = ptr(S.d_lm_off); Q.d_lm_idx = ptr(S.d_lm_idx); Q.lm_cap = S.lm_cap; Q.d_lm_tot = ptr(S.d_lm_tot); Q.d_lm_mpoff = ptr(S.d_lm_mpoff)
                Q.d_lm_desc = ptr(S.d_lm_desc); Q.d_lm_pd = ptr(S.d_lm_pd)
                Q.d_kpmp = ptr(S.d_kpmp); Q.d_kpmp0 = ptr(S.d_kpmp0); Q.kpmp_bytes = S.d_kpmp.numel() * S.d_kpmp.element_size()
                for gi in range(n_grp):
                    Q.put_items[gi] = C_.cast(put_items[j][gi], VP); Q.put_n[gi] = len(group_ids[gi])
            for gi in range(n_grp):
                G = P_.groups[gi]
                st_ = tri_store[gi]
                st_.create_new_map_points(tri_jobs[gi], copy=False)          # (sizes the job arrays and output buffers the wrapper keeps: st_._cnmp)
                key_, arrs_, _ = st_._cnmp
                cur_, off_, neigh_, n_new_, on_, o1_, o2_, ox_ = arrs_
                G.store = st_.h.value; G.ba_stream = int(group_stream[gi])
                G.njobs = len(tri_jobs[gi]); G.cur_slot = cur_.ctypes.data; G.neigh_off = off_.ctypes.data; G.neigh_slot = neigh_.ctypes.data; G.cap = key_[0]
                G.n_new = n_new_.ctypes.data; G.o_neigh = on_.ctypes.data; G.o_idx1 = o1_.ctypes.data; G.o_idx2 = o2_.ctypes.data; G.o_x3d = ox_.ctypes.data
                fa = fuse_prep[gi]["args"]
                val = lambda a_: a_.value if hasattr(a_, "value") else a_
                G.nsets = fa[0]; G.set_off = val(fa[1]); G.pos = val(fa[2]); G.normal = val(fa[3]); G.min_d = val(fa[4]); G.max_d = val(fa[5]); G.desc = val(fa[6])
                G.nfjobs = fa[7]; G.job_slot = val(fa[8]); G.job_set = val(fa[9]); G.skip = val(fa[10]); G.th = 3.0; G.best_idx = val(fa[12]); G.best_dist = val(fa[13])
                ua = upd_all[gi]["args"]
                G.n_upd = ua[0]; G.upd_slots = val(ua[1]); G.upd_R = val(ua[2]); G.upd_t = val(ua[3]); G.upd_Ow = val(ua[4])
                G.nwin = len(group_ids[gi])
                for j in range(2):
                    wa = api.ba_window_array([prob_sets[j][w] for w in group_ids[gi]])
                    cpp_keep.append(wa)
                    G.windows[j] = C_.cast(wa[0], VP)
            return P_
        def step_done_py(user, step_i, poses_ptr, nframes):
            # trajectory assembly on rank 0 over RCCL (72 B / frame, latency only), like step() does it
            frame_poses = np.ctypeslib.as_array(poses_ptr, shape=(nframes, 7)).copy()
            recs = [cdist.make_records(my_streams[s_], step_i * fps + np.arange(fps), frame_poses[s_ * fps:(s_ + 1) * fps]) for s_ in range(len(my_streams))]
            traj = cdist.gather_trajectory(np.concatenate(recs, 0), device=coll_dev, dst=0)
            if traj is not None:
                last["traj"] = traj
        step_done_c = STEP_DONE(step_done_py)
        def timed_cpp(steps=None, mapping_full=True):
            steps = args.steps if steps is None else steps
            P_ = make_plan(mapping_full)
            if world > 1 or args.force_gather:
                P_.step_done = step_done_c
            h_ = cpp.cbd_create(C_.byref(P_), int(po.off[-1]))
            if not h_:
                raise RuntimeError("cbd_create: %s" % cpp.cbd_last_error().decode())
            h_ = VP(h_)
            def ck(rc, what):
                if rc != 0:
                    raise RuntimeError("%s: %s" % (what, cpp.cbd_last_error().decode()))
            try:
                ck(cpp.cbd_begin(h_), "cbd_begin")
                ck(cpp.cbd_steps(h_, 0, args.warmup), "cbd_steps")
                ck(cpp.cbd_collect_inflight(h_), "cbd_collect_inflight")      # (the warm-up's last mapping side ends before the clock starts)
                barrier()
                cpp.cbd_stats_get(h_, None, 1)
                gc.collect(); gc.disable()
                t0 = time.perf_counter()
                ck(cpp.cbd_steps(h_, args.warmup, steps), "cbd_steps")
                ck(cpp.cbd_collect_inflight(h_), "cbd_collect_inflight")      # the last step's mapping side (pipelined steps)
                ck(cpp.cbd_finish(h_), "cbd_finish")                          # the last read-backs and their pose write-backs
                barrier()
                dt_ = time.perf_counter() - t0
                gc.enable()
                ck(cpp.cbd_drain(h_), "cbd_drain")
                if cpu0 is not None:
                    life["cpp_thread_cpu"] = thread_cpu()      # (the driver's threads end with the handle: their CPU seconds are read while they still exist)
                st_ = CbdStats()
                cpp.cbd_stats_get(h_, C_.byref(st_), 0)
            finally:
                gc.enable()
                cpp.cbd_destroy(h_)
            if world > 1:
                tt = torch.tensor([dt_], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_ = float(tt.item())
            return dt_, st_
    cpu0 = thread_cpu() if os.environ.get("CMS_BENCH_THREAD_CPU", "") != "" else None
    ctx.profile(True)
    python_loop = None
    cpu_t0 = time.process_time()
    if cpp is not None:
        dt, cst = timed_cpp()
        names7 = ("remap", "pyramid", "fast", "octree", "cull", "describe", "total")
        stage_ms = {k_: cst.stage_ms[i_] / max(cst.steps, 1) for i_, k_ in enumerate(names7)}
        ba_ms_per_step = cst.ba_ms_sum / max(cst.ba_jobs, 1)
        schur_prof = [(cst.schur_ms, cst.schur_launches)]
        life["mapping_full"] = True; life["on"] = True
        map_acc.update(put_ms=cst.put_ms_sum, put_n=cst.put_calls, fuse_ms=cst.fuse_ms_sum, fuse_n=cst.fuse_calls, upd_ms=cst.upd_ms_sum, upd_n=max(cst.upd_calls, 1) * upd_all[0]["n"],
                       fused=int(sum((fq["best_idx"] >= 0).sum() for fq in fuse_prep[-1:])))
        acc["create_ms"] = cst.create_ms_sum; acc["create_n"] = cst.create_windows
        worker_ms.clear()
        worker_ms.update(n=max(cst.ba_jobs, 1), wait_for_windows=cst.wait_windows_ms, create_new_map_points=cst.wait_tri_ms, optimize_many=cst.optimize_ms,
                         create_new_map_points_library_call=cst.tri_ms_sum)
        last["tri_new"] = int(cst.new_map_points_last_step)
    else:
        dt, stage_ms, ba_ms_per_step, schur_prof = timed(False)
    host["host_cores_used"] = round((time.process_time() - cpu_t0) / max(dt, 1e-9), 2)      # CPU seconds of this process per second of the timed pass (warm-up included in both)
    host["step_driver"] = ("c++ (cubemapslam_amd/host/batch_driver.cpp: the step's C-ABI calls from C++ threads, no interpreter in the loop)" if cpp is not None
                           else "python (bench.py's own loop; ~25 threads)")
    host_all = [host]
    if world > 1:
        host_all = [None] * world
        dist.all_gather_object(host_all, host)
    if cpu0 is not None:
        cpu1 = life.pop("cpp_thread_cpu", None) or thread_cpu()
        use = sorted(((cpu1[k] - cpu0.get(k, 0.0), k) for k in cpu1), reverse=True)
        tot = sum(u for u, _ in use)
        print("window threads: cms_ba_create %.2f ms CPU per window, read + destroy %.2f ms CPU per window (%d windows since start)" % (
            1e3 * cpu_acc["create"] / max(cpu_acc["n"], 1), 1e3 * cpu_acc["finish"] / max(cpu_acc["n"], 1), cpu_acc["n"]), file=sys.stderr)
        print("thread CPU over the timed pass: %.2f s in %.2f s of wall = %.1f cores; top: %s" % (
            tot, dt, tot / dt, ", ".join("%s/%s %.0f%%" % (k[1], k[0], 100 * u / dt) for u, k in use[:48])), file=sys.stderr)
    if step_times is not None:
        print("step times (ms):", step_times, file=sys.stderr); step_times.clear()
    if step_trace:
        # the last three steps: every stamp in ms after the step's beginning
        begins = [k for k, e in enumerate(step_trace) if e[0].startswith("step ")]
        for b0, b1 in zip(begins[-3:], begins[-2:] + [len(step_trace)]):
            tb = step_trace[b0][1]
            print(step_trace[b0][0] + ": " + "; ".join("%s %s" % (e[0], " ".join("%.2f" % (1e3 * (t - tb)) for t in e[1:])) for e in sorted(step_trace[b0 + 1:b1], key=lambda e: e[-1])), file=sys.stderr)
        step_trace.clear()
    # what the rest of LocalMapping's sequence took inside the headline pass (host wall time of the calls; their device work overlaps the step)
    mapping_side = {"in_timed_region": bool(life.get("mapping_full")),
                    "calls_per_key_frame": ["cms_kfstore_put_from_frame (ProcessNewKeyFrame: frame -> store, device to device)", "cms_kfstore_create_new_map_points",
                                            "cms_kfstore_fuse_search_sets (SearchInNeighbors: the key frame's map points into 20 neighbours + the neighbours' into the key frame; every set of map points uploaded once)",
                                            "cms_ba_create / cms_ba_optimize_many / cms_ba_read (LocalBundleAdjustment)", "cms_kfstore_update_poses (20 key-frame poses written back)"],
                    "put_from_frame_ms_per_step": round(map_acc["put_ms"] / max(map_acc["put_n"], 1), 3), "key_frames_per_step": n_ba,
                    "fuse_search_ms_per_group_call": round(map_acc["fuse_ms"] / max(map_acc["fuse_n"], 1), 3),
                    "fuse_jobs_per_step": int(sum(fq["njobs"] for fq in fuse_prep)), "fuse_projections_per_step": int(sum(fq["n_mp"] for fq in fuse_prep)),
                    "fuse_map_points_uploaded_per_step": int(sum(fq["n_set_points"] for fq in fuse_prep)),
                    "fused_matches_last_group_call": int(map_acc["fused"]),
                    "update_poses_ms_per_window": round(map_acc["upd_ms"] / max(map_acc["upd_n"], 1), 4),
                    "note": "LocalMapping::Run's sequence per key frame (LocalMapping.cpp:52-117, 388-466) inside the timed region since round 5.  The synthetic streams are not one "
                            "consistent scene: the inserted key frame lands in a slot of its own, the write-back writes the scene key frames' own poses -- every call's cost is in "
                            "the step, CreateNewMapPoints' geometry stays consistent.  `minimal` = the same steps with CreateNewMapPoints + local BA only (round 4's step)"}
    create_ms_in_step = acc["create_ms"] / max(acc["create_n"], 1)
    worker_break = {k_: round(v_ / max(worker_ms.get("n", 1), 1), 3) for k_, v_ in worker_ms.items() if k_ != "n"}      # per window group and step
    if cpp is not None and rank == 0 and world == 1 and os.environ.get("CMS_BENCH_NO_PY_LOOP", "") == "":
        n_py = min(args.steps, 10)
        c0_ = time.process_time()
        dt_p, _, ba_ms_p, _ = timed(False, steps=n_py)
        python_loop = {"value": round(total_frames_per_step * n_py / dt_p, 2), "ms_per_step": round(1e3 * dt_p / n_py, 3), "ba_ms_per_step": round(ba_ms_p, 3), "steps": n_py,
                       "host_cores_used": round((time.process_time() - c0_) / max(dt_p, 1e-9), 2),
                       "note": "the same step through bench.py's own Python loop (the driver of rounds 1-5): same calls, same pipelining, ~25 interpreter threads"}
    if part:
        if rank == 0:
            print(json.dumps({"developer_part": part, "ms_per_step": round(1e3 * dt / args.steps, 3), "config": {"ba_ms_per_step": round(ba_ms_per_step, 3), "ba_window_setup": {"ms_per_window_inside_the_step": round(create_ms_in_step, 2)}, "ba_worker_ms": worker_break, "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()}}}))
        pool.shutdown()
        wpool.shutdown()
        for tp_ in tri_pools:
            tp_.shutdown()
        return
    streamed = None
    if not args.no_streaming_pass:
        dt_s, _, ba_ms_s, _ = timed(True)
        streamed = {"value": round(total_frames_per_step * args.steps / dt_s, 2), "ms_per_step": round(1e3 * dt_s / args.steps, 3), "ba_ms_per_step": round(ba_ms_s, 3),
                    "host_bytes_per_step": int(B * camd["Ih"] * g.fisheye_stride),
                    "note": "same steps with every batch copied from pinned host memory (cms_frames_upload_async, overlapped with the previous step)"}

    # ---- round 2's figure for comparison: the same steps with STANDING windows that are only reset between steps (no creation, no read-back,
    # no destruction inside the timed region) -- what the kernels alone allow
    optimise_only = None
    if args.optimise_only_steps > 0:
        dt_o, _, ba_ms_o, _ = timed(False, lifecycle=False, steps=args.optimise_only_steps)
        optimise_only = {"value": round(total_frames_per_step * args.optimise_only_steps / dt_o, 2), "ms_per_step": round(1e3 * dt_o / args.optimise_only_steps, 3),
                         "ba_ms_per_step": round(ba_ms_o, 3), "steps": args.optimise_only_steps,
                         "note": "windows created once before the timed steps and reset between them (BENCH_r02's headline): kernels only, NOT like for like with the CPU baseline"}

    # ---- the frame path alone (no mapping side in the step): the extractor's roofline without the local BA next to it, timed by this run
    extract_only = None
    if args.extract_only_steps > 0:
        dt_f, stage_f, _, _ = timed(False, lifecycle=False, steps=args.extract_only_steps, only="frames")
        extract_only = {"ms_per_step": round(1e3 * dt_f / args.extract_only_steps, 3), "steps": args.extract_only_steps,
                        "stage_ms_per_step": {k: round(v, 4) for k, v in stage_f.items()},
                        "note": "the same batches through remap + extraction + grids + searches + pose optimisation with no local BA / CreateNewMapPoints in the step"}
    # ---- round 3's loop for comparison: every step waits for its own mapping side before the next one starts
    unpipelined = None
    if args.unpipelined_steps > 0 and n_ba > 0 and pipeline_default:
        dt_u, _, ba_ms_u, _ = timed(False, steps=args.unpipelined_steps, pipeline=False)
        unpipelined = {"value": round(total_frames_per_step * args.unpipelined_steps / dt_u, 2), "ms_per_step": round(1e3 * dt_u / args.unpipelined_steps, 3),
                       "ba_ms_per_step": round(ba_ms_u, 3), "steps": args.unpipelined_steps,
                       "note": "the same steps, each waiting for its own CreateNewMapPoints + local BA before the next step's frames are enqueued (BENCH_r03's loop)"}
    # ---- round 4's mapping side for comparison: CreateNewMapPoints + local BA only (no key-frame insertion, no Fuse, no pose write-back)
    if args.unpipelined_steps > 0 and n_ba > 0 and mapping_side["in_timed_region"]:
        dt_n, _, ba_ms_n, _ = timed(False, steps=args.unpipelined_steps, mapping_full=False)
        mapping_side["minimal"] = {"value": round(total_frames_per_step * args.unpipelined_steps / dt_n, 2), "ms_per_step": round(1e3 * dt_n / args.unpipelined_steps, 3),
                                   "ba_ms_per_step": round(ba_ms_n, 3), "steps": args.unpipelined_steps}
    # ---- the mapping side alone (no frame path in the step): what the Levenberg rounds' kernels take when nothing else shares the chip
    mapping_only = None
    if args.mapping_only_steps > 0 and n_ba > 0:
        dt_m, _, ba_ms_m, schur_m = timed(False, steps=args.mapping_only_steps, only="ba")
        sm_ms, sm_n = sum(m for m, _ in schur_m), sum(n for _, n in schur_m)
        mapping_only = {"ms_per_step": round(1e3 * dt_m / args.mapping_only_steps, 3), "ba_ms_per_step": round(ba_ms_m, 3), "steps": args.mapping_only_steps,
                        "schur_ms_per_launch": round(sm_ms / max(sm_n, 1), 4), "schur_launches": int(sm_n),
                        "note": "the same windows (created, optimised, read back, destroyed) and CreateNewMapPoints with no frame path in the step"}
    # ---- windows with RANDOM views (every point seen by an arbitrary subset of key frames: no signature runs, round 2's windows, the edge-major body
    # on every point) next to the headline's tracked views: same steps, same life cycle
    random_views = None
    if args.random_views_steps > 0 and args.ba_views != "random":
        with ThreadPoolExecutor(max_workers=8) as tp:
            rp = list(tp.map(lambda w: synth.ba_problem(K=20, P=22150, obs_per_point=4, F=F, seed=42 + 1000 * rank + w, views="random"), range(2 * n_ba)))
        for p_ in rp:
            p_["poses"] = np.ascontiguousarray(p_["poses"], np.float64); p_["points"] = np.ascontiguousarray(p_["points"], np.float64)
            p_["e_obs"] = np.ascontiguousarray(p_["e_obs"], np.float64)
        keep_sets = list(prob_sets)
        prob_sets[0], prob_sets[1] = rp[:n_ba], rp[n_ba:]
        dt_r, _, ba_ms_r, _ = timed(False, steps=args.random_views_steps)
        prob_sets[0], prob_sets[1] = keep_sets
        random_views = {"value": round(total_frames_per_step * args.random_views_steps / dt_r, 2), "ms_per_step": round(1e3 * dt_r / args.random_views_steps, 3),
                        "ba_ms_per_step": round(ba_ms_r, 3), "steps": args.random_views_steps, "edges_per_window": int(np.mean([len(p_["e_pose"]) for p_ in rp])),
                        "note": "synth.ba_problem(views='random'): no point shares its set of observing key frames with enough others, every point goes through the edge-major Schur body"}

    # ---- determinism as a product mode (cms_ba_set_deterministic: fixed summation order, bit-identical runs like the reference's single-threaded g2o):
    # the same steps with every window created under the mode -- since round 6 the fused chain with its LDS additions in a fixed order and slices instead
    # of the global copy (kb_ba_lin_schur_runs_det), planned like any other window; CMS_BA_DET_POINTS=1: rounds 3-5's pair-owner kernel on the host's full plan
    deterministic = None
    if args.deterministic_steps > 0 and n_ba > 0:
        api.ba_set_deterministic(True)
        try:
            warm_keep = args.warmup
            args.warmup = max(args.warmup, 3)          # (these windows are larger: their slabs enter the pool during the warm-up)
            dt_d, _, ba_ms_d, _ = timed(False, steps=args.deterministic_steps)
            dt_dc = None
            if cpp is not None and world == 1 and os.environ.get("CMS_BA_DET_POINTS", "") == "":      # ... and through the headline's C++ step driver (the mode is the process's: its windows are deterministic too)
                dt_dc, _ = timed_cpp(steps=args.deterministic_steps)
            args.warmup = warm_keep
        finally:
            api.ba_set_deterministic(False)
        deterministic = {"value": round(total_frames_per_step * args.deterministic_steps / dt_d, 2), "ms_per_step": round(1e3 * dt_d / args.deterministic_steps, 3),
                         "ba_ms_per_step": round(ba_ms_d, 3), "steps": args.deterministic_steps,
                         "of_python_step_loop": (round(total_frames_per_step * args.deterministic_steps / dt_d / python_loop["value"], 3) if python_loop else None),      # (this pass runs the Python loop too)
                         "step_driver": ({"value": round(total_frames_per_step * args.deterministic_steps / dt_dc, 2), "ms_per_step": round(1e3 * dt_dc / args.deterministic_steps, 3),
                                          "of_value": round(total_frames_per_step * args.deterministic_steps / dt_dc / (total_frames_per_step * args.steps / dt), 3),
                                          "note": "the same deterministic steps through the C++ step driver, like `value`"} if dt_dc else None),
                         "note": ("cms_ba_set_deterministic(1): the fused chain with the Schur kernel's LDS additions in a fixed order (kb_ba_lin_schur_runs_det: keys from the plan's "
                                  "estimated chunk costs), workgroup slices added by the solve kernel in slice order instead of global FP64 atomics, 16 workgroups per window whatever the "
                                  "group, kb_ba_first_pass adding the key frames' diagonal sums in chunk and slice order; planned like default windows -- bit-identical from run to run"
                                  if os.environ.get("CMS_BA_DET_POINTS", "") == "" else
                                  "CMS_BA_DET_POINTS: windows run kb_ba_schur_points (pair-owner kernel) and are planned by the host planner with every work list (rounds 3-5)")}

    # ---- outside the timed region: one more step with the windows' whole life cycle whose results are kept; iteration counts and outlier
    # counts of ALL its windows, and the estimates of a sample, against the CPU oracle
    ba_check = None
    cpu_ba_ms = []
    if args.verify_windows > 0:
        life["on"] = True
        for a_ in range(ahead):
            submit_windows(a_ & 1)
        step(args.warmup + args.steps, False, keep=True)          # (all ranks: it holds the gather)
        drain()
    if rank == 0 and args.verify_windows > 0:
        import orc
        vprobs = prob_sets[last["set"]]
        stats_of, out_of = {}, {}
        for gi, ids in enumerate(group_ids):
            for wi, w in enumerate(ids):
                stats_of[w] = last["ba_stats"][gi][wi]; out_of[w] = last["ba_out"][gi][wi]
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as tp:     # the oracle on every window (its own threads: the check, not a timing)
            wants = list(tp.map(orc.ba_run, vprobs))
        for w in range(n_ba):
            st, ws = stats_of[w], wants[w]["stats"]
            if list(st.iterations_done) != list(ws.iterations_done) or st.n_outliers_mid != ws.n_outliers_mid or st.n_outliers_final != ws.n_outliers_final or \
                    not np.array_equal(out_of[w][2], wants[w]["outliers"]):
                raise SystemExit("bench.py: local-BA window %d differs from the CPU oracle: iterations %s vs %s, outliers %d vs %d" %
                                 (w, list(st.iterations_done), list(ws.iterations_done), st.n_outliers_final, ws.n_outliers_final))
        sample = sorted(set(np.linspace(0, n_ba - 1, min(args.verify_windows, n_ba)).astype(int).tolist()))
        # estimates, per block, against the oracle's (1e-4 of the block's update, floor 1 % of the median update).  The reference rounds the
        # camera-frame point to float inside the double optimisation (g2o_cubemap_vertices_edges.cpp:225-233): about one window in twenty gets a
        # rounding flip that cascades (the oracle does it to itself under a 1e-12 m perturbation: tests/test_oracle_ba.py) -- such a window's
        # key frames must still be within 1e-4, of its points at most 2 % may be beyond and none beyond 1e-2; it is counted, not hidden
        worst, cascaded = 0.0, []
        for w in sample:
            want = wants[w]
            poses, pts, flags = out_of[w]
            rel = []
            for got, ref, ref0 in ((pts, want["points"], vprobs[w]["points"]), (poses[:, :3], want["poses"][:, :3], vprobs[w]["poses"][:, :3])):
                nrm = np.linalg.norm(ref - ref0, axis=1)
                err = np.linalg.norm(got - ref, axis=1)
                floor = 0.01 * np.median(nrm[nrm > 0]) if np.any(nrm > 0) else 0.0
                rel.append(err / np.maximum(nrm, max(floor, 1e-300)))
            if rel[1].max() > 1e-4:
                raise SystemExit("bench.py: local-BA window %d: key-frame update differs from the CPU oracle by %.3g relative" % (w, rel[1].max()))
            if rel[0].max() > 1e-4:
                if (rel[0] > 1e-4).mean() > 0.02 or rel[0].max() > 1e-2:
                    raise SystemExit("bench.py: local-BA window %d: point updates differ from the CPU oracle (%.3g relative, %d points beyond 1e-4)" %
                                     (w, rel[0].max(), int((rel[0] > 1e-4).sum())))
                cascaded.append({"window": w, "points_beyond_1e-4": int((rel[0] > 1e-4).sum()), "worst_point": float("%.3g" % rel[0].max()),
                                 "worst_key_frame": float("%.3g" % rel[1].max())})
            else:
                worst = max(worst, float(rel[0].max()), float(rel[1].max()))
        # about one window in twenty cascades; three or more in a sample of eight (0.6 % by chance) means something else is wrong
        if len(cascaded) > max(2, len(sample) // 4):
            raise SystemExit("bench.py: %d of %d sampled local-BA windows show a float-rounding cascade against the CPU oracle (expected about 1 in 20): %s" %
                             (len(cascaded), len(sample), cascaded))
        its = [tuple(s.iterations_done) for gs in last["ba_stats"] for s in gs]
        ba_check = {"windows_with_iterations_and_outlier_flags_equal_to_the_oracle": n_ba, "windows_with_estimates_checked": sample,
                    "worst_relative_update_error": float("%.3g" % worst), "windows_with_a_float_rounding_cascade": cascaded,
                    "iterations_done_min_max": [list(min(its)), list(max(its))], "distinct_iteration_counts": len(set(its))}

    # ---- rooflines (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
    sumP = sum(g.level_w[l] * g.level_h[l] for l in range(g.nlevels))
    P = [g.level_w[l] * g.level_h[l] for l in range(g.nlevels)]
    nkp = float(np.mean([len(k) for k in kps]))
    alg = {
        "remap": camd["Iw"] * camd["Ih"] + 5 * F * F * (1 + 4),                     # source + dest + one packed u32 LUT entry / px
        "pyramid": sum(P[l - 1] + P[l] for l in range(1, g.nlevels)),               # read level l-1, write level l
        "fast": sumP,                                                                # every pyramid pixel read once
        "describe": nkp * (43 * 43 + 32 + 24),                                       # raw patch + descriptor + key-point record
    }
    peak = 8000.0
    # measured HBM traffic / instruction mix of the kernels: committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    # separate runs, tools/run_profiles.sh), valid for the batch size and geometry they were taken at
    def pmc(name, kname):
        for rnd in ("r05", "r04", "r03", "r02", "r01"):
            pj = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))
            if os.path.exists(pj):
                J = json.load(open(pj))
                ok = kname.startswith("kb_") or J.get("frames_per_dispatch", 64) == B      # the local-BA passes are per 8 windows, whatever the frame batch
                if ok and J.get("face", 550) == F and kname in J.get("kernels", {}):
                    return J["kernels"][kname]
        return None
    def traffic_of(kname, scale=1.0):
        kk = pmc("pmc_hbm_traffic", kname)
        if kk and "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
            # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section) -> doubled; unit KB
            return int((2.0 * kk["FETCH_SIZE"]["per_dispatch"] + kk["WRITE_SIZE"]["per_dispatch"]) * 1024 * scale)
        return None
    fast_ms = stage_ms.get("fast", 0.0)
    fast_gbs = alg["fast"] * B / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else None
    mixk = pmc("pmc_instruction_mix", "k_fast_cells")
    roof_fast = {"kernel": "k_fast_cells", "bound": "hbm", "achieved": None if fast_gbs is None else round(fast_gbs, 1), "peak": peak, "unit": "GB/s",
                 "frac": None if fast_gbs is None else round(fast_gbs / peak, 4), "traffic": traffic_of("k_fast_cells"),
                 "ms_per_launch": round(fast_ms, 4), "ms_per_step": round(fast_ms, 4),
                 "valu_issue_bound_ms": None if not mixk else round(mixk["valu_issue_bound_us"] / 1e3, 4),
                 "algorithmic_bytes_per_launch": int(alg["fast"] * B),
                 "all_stages_GBps": {k: round(alg[k] * B / (stage_ms[k] * 1e-3) / 1e9, 1) for k in alg if stage_ms.get(k, 0) > 0}}
    # The Schur kernel of the local-BA chain (kb_ba_lin_schur_runs by default: linearisation + Schur complement of one Levenberg trial of one
    # window group per launch).  It moves little memory (the 6x3 blocks are rebuilt from the estimate, never read: PMC traffic is a fraction of
    # the E 144 + P 96 bytes SURVEY.md 8d counts), so HBM is not its ceiling.  Two ceilings apply and both are reported:
    #   fp64 vector issue  algorithmic flops (per observation: residual, Huber weight, Jacobians, its shares of Hll / bl / Hpp / bp, W = B L^-T:
    #                      320; per point: the 3x3 LDL^T: 30; per tuple of observations of one point, diagonal ones included: W_a D^-1 W_b^T and
    #                      its share of the right-hand side: 234) against 78.6 TFLOP/s (gfx950: the FP64 matrix rate equals the vector rate)
    #   LDS atomics        ds_add_f64 lane-operations on the workgroup's copy of the reduced system against the 7.8 per clock and CU the LDS
    #                      takes on conflict-free addresses (tools/probe/lds_atomics.hip, profiles/r03_probe_lds_atomics.txt)
    # `bound` names the ceiling the kernel sits closer to.
    sch_ms = sum(p[0] for p in schur_prof); sch_n = sum(p[1] for p in schur_prof)
    roof_ba = None
    if sch_n > 0 and rank == 0:
        avg_ms = sch_ms / sch_n
        wpl = n_ba / n_grp                     # windows per launch
        fl, atom, byt, run_stats = [], [], [], []
        fixed0 = probs[0]["fixed"]
        slot_free = fixed0 == 0
        for pw in probs[:8]:
            Pn, En = len(pw["points"]), len(pw["e_pose"])
            plan = api.ba_plan(pw["fixed"], Pn, pw["e_pose"], pw["e_point"])
            kf = np.bincount(pw["e_point"][slot_free[pw["e_pose"]]], minlength=Pn).astype(np.int64)      # free observations per point
            tuples = int((kf * (kf + 1) // 2).sum())
            fl.append(En * 320.0 + Pn * 30.0 + tuples * 234.0)
            byt.append(En * 144.0 + Pn * 96.0)
            prank = np.empty(Pn, np.int64); prank[plan["pinv"]] = np.arange(Pn)
            left = prank >= plan["rm_points"]                                                           # points the edge-major body takes
            a_se = int((kf[left] * 33 + (kf[left] * (kf[left] - 1) // 2) * 36).sum())
            R_rm_grp = max(1, int(round(256 / wpl * 0.5 * plan["n_rm"] / max(0.5 * plan["n_rm"] + plan["n_chunks"] - plan["n_rm"], 1)))) if plan["n_rm"] else 0
            a_rm = (plan["n_runs"] + 4 * R_rm_grp) * 64 * (36 + 33) if plan["n_rm"] else 0              # one set of additions per run and per pair range
            atom.append(a_se + a_rm)
            run_stats.append((plan["n_runs"], plan["rm_points"] / Pn, plan["n_rm"], plan["n_chunks"]))
        flops = float(np.mean(fl)) * wpl; atoms = float(np.mean(atom)) * wpl
        tflops = flops / (avg_ms * 1e-3) / 1e12
        lane_ops = atoms / (avg_ms * 1e-3 * 2.4e9 * 256)
        f_fp64, f_lds = tflops / 78.6, lane_ops / 7.8
        schur_name = ("kb_ba_schur_points" if os.environ.get("CMS_BA_DETERMINISTIC") or os.environ.get("CMS_BA_HOST_LM") else
                      "kb_ba_schur_edges" if os.environ.get("CMS_BA_NO_FUSED_LIN") else
                      "kb_ba_lin_schur_edges" if os.environ.get("CMS_BA_NO_RUNS") or os.environ.get("CMS_BA_RUNS_AS_EDGES") or args.ba_views == "random" else "kb_ba_lin_schur_runs")
        traffic = traffic_of(schur_name, wpl / 8.0)       # PMC pass: 8 windows per dispatch (tools/prof_ba_many.py); one launch here carries n_ba / n_grp windows
        fp64_bound = f_fp64 >= f_lds
        # (bound "mfma": the dense FP64 matrix peak of gfx950 -- 78.6 TFLOP/s, the same figure as its FP64 vector peak; the kernel's flops are split
        # between the two pipes: products on v_mfma_f64_16x16x4_f64, residuals / Jacobians / 3x3 factorisations on the vector ALU)
        roof_ba = {"kernel": schur_name, "bound": "mfma" if fp64_bound else "lds-atomic",
                   "achieved": round(tflops, 3) if fp64_bound else round(lane_ops, 3), "peak": 78.6 if fp64_bound else 7.8,
                   "unit": "TFLOP/s" if fp64_bound else "ds_add_f64 lane-ops/clk/CU", "frac": round(max(f_fp64, f_lds), 4), "traffic": traffic,
                   "fp64": {"algorithmic_flops_per_launch": int(flops), "TFLOPs": round(tflops, 3), "peak_TFLOPs": 78.6, "frac": round(f_fp64, 4),
                            "pipes": "FP64 vector ALU + v_mfma_f64_16x16x4_f64 (one 78.6 TFLOP/s peak on gfx950: the matrix rate equals the vector rate)"},
                   "lds_atomic": {"lane_ops_per_launch": int(atoms), "per_clk_per_cu": round(lane_ops, 3), "peak_per_clk_per_cu": 7.8, "frac": round(f_lds, 4),
                                  "clock_GHz_assumed": 2.4},
                   "hbm_GBps_measured": None if traffic is None else round(traffic / (avg_ms * 1e-3) / 1e9, 1),
                   "hbm_equivalent": {"survey_bytes_per_launch": int(float(np.mean(byt)) * wpl), "GBps": round(float(np.mean(byt)) * wpl / (avg_ms * 1e-3) / 1e9, 1),
                                      "note": "E 144 + P 96 bytes per window (SURVEY.md 8d) the kernel never moves: not a ceiling, kept for comparison with BENCH_r02"},
                   "ms_per_launch": round(avg_ms, 4), "launches_per_step": round(sch_n / args.steps, 1), "ms_per_step": round(sch_ms / args.steps, 4),
                   "windows_per_launch": wpl,
                   "signature_runs": {"runs_per_window": round(float(np.mean([r[0] for r in run_stats])), 1), "points_in_runs": round(float(np.mean([r[1] for r in run_stats])), 3),
                                      "run_chunks_of_all_chunks": "%d / %d" % (int(np.mean([r[2] for r in run_stats])), int(np.mean([r[3] for r in run_stats])))},
                   "traffic_note": "PMC counters (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE) of an 8-window launch of tools/prof_ba_many.py, "
                                   "scaled to this run's %d windows per launch" % wpl,
                   "note": "HIP events on each window group's stream around the kernel of every round, inside the timed region; the groups' launches overlap each "
                           "other and the frame path, so ms_per_step is summed kernel time"}
    if roof_ba and mapping_only is not None and mapping_only["schur_launches"] > 0:
        # the same kernel, the same windows, with no frame path next to it (config.mapping_only): what the launch takes when only the mapping side's
        # own queues share the chip -- `frac` above stays the driver's figure, measured inside the full step
        ms_a = mapping_only["schur_ms_per_launch"]
        roof_ba["without_the_frame_path"] = {"ms_per_launch": ms_a, "TFLOPs": round(flops / (ms_a * 1e-3) / 1e12, 3), "frac": round(flops / (ms_a * 1e-3) / 1e12 / 78.6, 4),
                                             "launches": mapping_only["schur_launches"]}
    # `roofline` = the kernel with the most time per step; the other one is reported next to it
    if roof_ba and roof_ba["ms_per_step"] > roof_fast["ms_per_step"]:
        roof, roof_other = roof_ba, roof_fast
    else:
        roof, roof_other = roof_fast, roof_ba
    # whole ORBextractor::operator() against SURVEY.md 8(d)'s compulsory traffic at the reference's stage granularity
    # (B_pyr + B_fast + B_blur + B_kp; the full-level blur pass is counted although the product never runs it)
    b_extract = (2 * P[0] + sum(P[l - 1] + P[l] for l in range(1, g.nlevels))) + sumP + 2 * sumP + nkp * (709 + 961 + 32 + 28)
    def extractor_of(stage):
        ext_ms = sum(stage.get(k, 0.0) for k in ("pyramid", "fast", "octree", "cull", "describe"))
        if ext_ms <= 0:
            return None
        ext_gbs = b_extract * B / (ext_ms * 1e-3) / 1e9
        return {"survey_bytes_per_frame": int(b_extract), "us_per_frame": round(1e3 * ext_ms / B, 2), "GBps": round(ext_gbs, 1), "frac_of_8TBps": round(ext_gbs / peak, 4)}
    extractor = extractor_of(stage_ms)
    if extract_only is not None:
        extract_only["extractor_vs_survey_bytes"] = extractor_of(extract_only["stage_ms_per_step"])

    # ---- CPU baseline: the oracle, single thread, on a bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        import orc
        S0 = sets[0]
        frames0 = S0.pinned.array[:, :, :camd["Iw"]]
        ocam = orc.make_camera(camd)
        m1, m2 = orc.build_lut(ocam)
        o = orc.Orb(nfeatures=nfeat)
        n = min(args.cpu_frames, B)
        t1 = time.perf_counter()
        for b in range(n):
            cube = orc.fisheye_to_cubemap(ocam, m1, m2, np.ascontiguousarray(frames0[b]))
            o.extract(ocam, cube, mask)
        t_ext = time.perf_counter() - t1
        # TrackWithMotionModel's SearchByProjection, then TrackLocalMap's SearchLocalPoints on what is left, per frame like the GPU leg
        t1 = time.perf_counter()
        kp_after = []
        for b in range(n):
            kb, db = S0.fetched[b]
            m = S0.mm[b]
            kpm = np.full(len(kb), -1, np.int32)
            orc.search_by_projection_frames(ocam, m["pose12"][:9], m["pose12"][9:], kb["x"], kb["y"], kb["octave"], kb["angle"], db, m["scale_factors"],
                                            m["valid"], m["Xw"], m["octave"], m["angle"], m["desc"], kpm, th=15.0, check_ori=True)
            kp_after.append(kpm)
        t_match = (time.perf_counter() - t1) / n
        t1 = time.perf_counter()
        for b in range(n):
            kb, db = S0.fetched[b]
            l = S0.lm[b]
            fr = orc.is_in_frustum(ocam, l["pose15"], l["pos"], l["normal"], l["min_dist"], l["max_dist"])
            orc.search_local_points(ocam, kb["x"], kb["y"], kb["octave"], db, l["scale_factors"], fr, l["desc"], kp_after[b])
        t_local = (time.perf_counter() - t1) / n
        t1 = time.perf_counter()
        T0 = tri_sets[0]
        oks = [orc.make_keyframe(ocam, k) for k in T0["kfs"]]
        for _ in range(2):
            orc.create_new_map_points(ocam, oks[0][0], [k for k, _ in oks[1:]], T0["scale_factors"], T0["level_sigma2"], T0["kfs"][0]["mp"].copy())
        t_tri = (time.perf_counter() - t1) / 2
        # SearchInNeighbors' Fuse searches for one key frame (both directions: 21 jobs), the oracle's scan
        t1 = time.perf_counter()
        inv_s2 = (np.float32(1.0) / (T0["scale_factors"] * T0["scale_factors"])).astype(np.float32)
        for slot_, skip_, pos_, nrm_, mind_, maxd_, desc_ in fuse_jobs_of(T0, 0):
            orc.fuse_search(ocam, oks[slot_][0], skip_, pos_, nrm_, mind_, maxd_, desc_, 3.0, T0["scale_factors"], inv_s2)
        t_fuse = time.perf_counter() - t1
        while len(cpu_ba_ms) < 3:            # three windows, one after the other, one thread
            t1 = time.perf_counter()
            orc.ba_run(probs[len(cpu_ba_ms) % n_ba])
            cpu_ba_ms.append(1e3 * (time.perf_counter() - t1))
        t_ba = 1e-3 * float(np.mean(cpu_ba_ms))
        t1 = time.perf_counter()
        for b in range(min(n, 8)):
            orc.pose_optimize(pose_probs[b])
        t_pose = (time.perf_counter() - t1) / min(n, 8)
        per_frame = t_ext / n + t_match + t_local + t_pose + (t_ba + t_tri + t_fuse) / args.ba_every
        cpu = {"value": round(1.0 / per_frame, 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames remap+extract (%.1f ms/frame), frame-to-frame SearchByProjection (%.2f ms/frame), local-map search "
                         "(%.2f ms/frame), pose-only optimisation (%.2f ms/frame), CreateNewMapPoints (%.1f ms per key frame), SearchInNeighbors' Fuse searches (%.1f ms per key frame), %d local-BA windows (%.1f ms each, 1 per %d frames); "
                         "oracle/liborc.so, single thread (two_threads_like_the_reference: the same timings with local BA on a second core next to tracking, System.cpp:108-127)" %
                         (n, 1e3 * t_ext / n, 1e3 * t_match, 1e3 * t_local, 1e3 * t_pose, 1e3 * t_tri, 1e3 * t_fuse, len(cpu_ba_ms), 1e3 * t_ba, args.ba_every),
               "two_threads_like_the_reference": round(1.0 / max(t_ext / n + t_match + t_local + t_pose, (t_ba + t_tri + t_fuse) / args.ba_every), 3),
               "host_cores_available": os.cpu_count()}

    # ---- one LocalBundleAdjustment call as the reference issues it (cms_ba_run: graph set-up + optimisation + read-back + destroy), chip quiet
    ba_call = None
    if rank == 0 and world == 1 and n_ba > 0 and args.closed_loop_frames > 0:      # (--closed-loop-frames 0, the profiling command: only the step's own launches in the trace)
        ts = []
        for _ in range(4):
            t1 = time.perf_counter()
            api.ba_run(probs[0], device=local_rank)
            ts.append(1e3 * (time.perf_counter() - t1))
        ba_call = {"ms": round(float(np.median(ts[1:])), 2), "edges": int(len(probs[0]["e_pose"])),
                   "note": "cms_ba_run on window 0's problem: set-up (host work list, uploads) + both optimisation stages + read-back + destroy; median of 3"}

    # ---- single stream, closed loop (configs[2] the way the reference runs it: one frame after the other, matches feed the pose feed the
    # next projection; cubemapslam_amd/harness.py, parity in tests/test_gpu_harness.py) next to the batch figure above
    closed = None
    if rank == 0 and world == 1 and args.closed_loop_frames >= 8 and not front:
        from cubemapslam_amd import harness
        fr_cl, gt_cl = harness.render_sequence(camd, args.closed_loop_frames)
        be = harness.GpuBackend(camd, mask, device=local_rank)
        harness.run_sequence(camd, be, fr_cl[:6], gt_cl[:6])                     # warm-up
        trk, secs = harness.run_sequence(camd, be, fr_cl, gt_cl)
        be.close()
        tr = [s_ for s_, r in zip(secs, trk.log) if r.get("stage") == "track" and "ba_iterations" not in r]
        kfr = [s_ for s_, r in zip(secs, trk.log) if "ba_iterations" in r]
        closed = {"frames": len(secs), "state": trk.state, "frames_per_s": round(len(secs) / sum(secs), 1),
                  "median_ms_tracked_frame": round(1e3 * float(np.median(tr)), 3) if tr else None,
                  "median_ms_key_frame_with_local_ba": round(1e3 * float(np.median(kfr)), 3) if kfr else None,
                  "mean_inliers": round(float(np.mean([r["n_inliers"] for r in trk.log if "n_inliers" in r])), 1) if trk.state == "ok" else None,
                  "note": "ray-cast box-room stream through the harness, one frame at a time through the host-buffer C-ABI entries; Python glue included"}
        # the same stream through the Python-free driver (cubemapslam_amd/host/closed_loop_driver.cpp: image files in, the reference's summary out)
        try:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                harness.export_sequence(td, camd, fr_cl, gt_cl, mask)
                rc_d, recs_d, out_d = harness.run_driver(td, device=local_rank, warmup=len(fr_cl))      # warm-up: a throw-away tracker over the same images (pools, first launches)
            trd = [r["ms"] for r in recs_d if r.get("stage") == "track" and "ba_iterations" not in r and "n_inliers" in r]
            kfd = [r["ms"] for r in recs_d if "ba_iterations" in r]
            closed["cpp_driver"] = {"exit_code": rc_d, "frames": len(recs_d), "frames_per_s": round(1e3 * len(recs_d) / max(sum(r["ms"] for r in recs_d), 1e-9), 1),
                                    "median_ms_tracked_frame": round(float(np.median(trd)), 3) if trd else None,
                                    "median_ms_key_frame_with_local_ba": round(float(np.median(kfd)), 3) if kfd else None,
                                    "mean_inliers": round(float(np.mean([r["n_inliers"] for r in recs_d if "n_inliers" in r])), 1) if trd else None,
                                    "max_ms_frame": round(float(max(r["ms"] for r in recs_d)), 3) if recs_d else None,
                                    "note": "cubemap_closed_loop: plain C++ over the C-ABI, one frame after the other, no Python in the loop; measured after a "
                                            "throw-away tracker has been through the same images once (device pools and first launches are one-time costs)"}
        except Exception as ex:      # the driver is a report line, not the metric
            closed["cpp_driver"] = {"error": str(ex)[:200]}

    # ---- the same step with the whole process confined to 2 and to 4 host cores: measured by child processes BEFORE this process touched the GPU (main()'s beginning)
    if early_confined:
        host.update(early_confined)
    if rank == 0 and args.save_trajectory and last["traj"] is not None:
        # rank 0's assembled trajectory of the last step in the reference's TUM format (System.cpp:238-268)
        tr = last["traj"]
        q = tr[:, 5:9]
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Tcw = np.zeros((len(tr), 4, 4), np.float32)
        Tcw[:, 0, 0] = 1 - 2 * (y * y + z * z); Tcw[:, 0, 1] = 2 * (x * y - z * w); Tcw[:, 0, 2] = 2 * (x * z + y * w)
        Tcw[:, 1, 0] = 2 * (x * y + z * w); Tcw[:, 1, 1] = 1 - 2 * (x * x + z * z); Tcw[:, 1, 2] = 2 * (y * z - x * w)
        Tcw[:, 2, 0] = 2 * (x * z - y * w); Tcw[:, 2, 1] = 2 * (y * z + x * w); Tcw[:, 2, 2] = 1 - 2 * (x * x + y * y)
        Tcw[:, :3, 3] = tr[:, 2:5]; Tcw[:, 3, 3] = 1
        cdist.write_trajectory_tum(args.save_trajectory, tr[:, 1], Tcw)
    if rank == 0:
        S0 = sets[0]
        name = "front_cam (parkinglot_front) x %d streams" % args.streams if front else "Lafida cam0"
        out = {
            "metric": "frames/sec (extract+match+localBA) on %s" % ("8 front_cam streams" if front else "Lafida cam0"),
            "value": round(total_frames_per_step * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8 (extract/match), f64 (BA)", "data": "synthetic",
            **({"developer_smoke_test": "CMS_BENCH_SHARED_GPU_GLOO: all ranks on ONE GPU, collectives over gloo -- control flow only, NOT a measurement"} if shared_gpu_debug else {}),
            "config": {"workload": "%s synthetic streams, %dx%d fisheye, face=%d (%dx%d cross), nFeatures %d; per step and GPU %d frames (%d streams x %d consecutive frames; two "
                                   "batches alternate): remap+ORB extract, frame grids, frame-to-frame SearchByProjection (projection + GetFeaturesInArea windows + greedy Hamming match + "
                                   "rotation histogram: %d map points, %d candidate pairs), local-map search (isInFrustum + SearchByProjection, %d map points, %d candidate pairs), "
                                   "pose-only optimisation (%d edges/frame), %d key frames x (CreateNewMapPoints against 20 neighbours + local BA window K=20, E~%d, every window a "
                                   "different problem)"
                                   % (name, camd["Iw"], camd["Ih"], F, 3 * F, 3 * F, nfeat, B, len(my_streams), fps, S0.nq, S0.n_pairs, S0.n_mp, S0.lm_pairs, args.pose_edges, n_ba,
                                      int(np.mean([b.E for b in bas]))),
                       "frames_per_step_per_gpu": B, "frames_per_step_total": total_frames_per_step, "keypoints_per_frame": round(nkp, 1), "ba_every_frames": args.ba_every,
                       "inputs": "resident in HBM (two batches, device-to-device copy into the staging buffer inside the step)",
                       "queues": {"frame_path_priority": fprio or "normal", "mapping_side_priority": os.environ.get("CMS_BENCH_MAP_PRIORITY", "") or "normal"},
                       "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                       "fast_kernel_GBps": None if fast_gbs is None else round(fast_gbs, 1),
                       "extractor_vs_survey_bytes": extractor, "extract_only": extract_only,
                       "host": host_all if world > 1 else host,
                       "ba_windows_per_step": n_ba, "ba_groups": n_grp,
                       "ba_window_setup": {"in_timed_region": True, "ms_per_window_one_after_the_other": round(ba_setup_ms, 2),
                                           "ms_per_window_inside_the_step": round(create_ms_in_step, 2), "window_threads": n_wthreads,
                                           "note": "every step creates its %d windows from the problems' host arrays (cms_ba_create: host work lists, one pinned upload), optimises "
                                                   "them, reads poses / points / outlier flags back (cms_ba_read) and destroys them; a pool of host threads builds step s + 1's "
                                                   "windows and finishes step s - 1's while step s runs, all inside the timed region" % n_ba},
                       "ba_worker_ms": worker_break, "mapping_side": mapping_side, "deterministic": deterministic, "optimise_only": optimise_only, "unpipelined": unpipelined, "mapping_only": mapping_only,
                       "step_pipelining": ("the mapping side of step s (CreateNewMapPoints + local BA, windows created / read back / destroyed) is waited for at the end of step s + 1: "
                                           "it overlaps the next batch's frame path like LocalMapping overlaps Tracking; all of it inside the timed region") if pipeline_default else "off (CMS_BENCH_NO_PIPELINE)", "ba_views": args.ba_views, "ba_views_random": random_views,
                       "ba_ms_per_step": round(ba_ms_per_step, 3), "new_map_points_per_step": last["tri_new"],
                       "ba_check": ba_check, "one_local_ba_call": ba_call, "with_input_streaming": streamed, "single_stream_closed_loop": closed, "python_step_loop": python_loop},
            "roofline": roof, "roofline_other": roof_other, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    pool.shutdown()
    wpool.shutdown()
    for tp_ in tri_pools:
        tp_.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
