"""N > 1 path on CPU: two gloo processes shard 8 streams, run the (host-side) per-stream bookkeeping and gather the
trajectory on rank 0 -- the only exchange step of the multi-GPU design."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cubemapslam_amd import dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = cdist.streams_of_rank(n_streams, world, rank)
    recs = []
    for s in mine:
        n = frames + s            # ragged: streams have different lengths
        ts = np.arange(n) / 30.0
        poses = np.zeros((n, 7)); poses[:, 0] = s; poses[:, 1] = np.arange(n); poses[:, 6] = 1.0
        recs.append(cdist.make_records(s, ts, poses))
    local = np.concatenate(recs, 0) if recs else np.zeros((0, cdist.RECORD))
    traj = cdist.gather_trajectory(local)
    dist.barrier()
    if rank == 0:
        q.put(traj)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_streams", [(2, 8), (2, 3)])
def test_stream_sharding_and_trajectory_gather(world, n_streams):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    traj = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sum(5 + s for s in range(n_streams))
    assert traj.shape == (want, 9)
    # sorted by (stream, ts), every stream complete and untouched
    for s in range(n_streams):
        rows = traj[traj[:, 0] == s]
        assert len(rows) == 5 + s
        assert np.allclose(rows[:, 1], np.arange(5 + s) / 30.0) and np.all(rows[:, 2] == s) and np.all(rows[:, 8] == 1.0)
    assert np.all(np.diff(traj[:, 0]) >= 0)


def test_partition_covers_all_streams():
    from cubemapslam_amd import dist as cdist
    for world in (1, 2, 4, 8):
        got = sorted(s for r in range(world) for s in cdist.streams_of_rank(8, world, r))
        assert got == list(range(8))
        assert max(len(cdist.streams_of_rank(8, world, r)) for r in range(world)) == 8 // world
    rec = cdist.make_records(3, [0.0, 1.0], np.ones((2, 7)))
    assert rec.shape == (2, 9) and np.all(rec[:, 0] == 3)
    one = cdist.gather_trajectory(rec)           # world size 1: local sort only
    assert one.shape == (2, 9)


def _json_records(text):
    """the launcher self-test records in a process group's shared stdout (several objects may share a line)"""
    import json
    dec, recs, i = json.JSONDecoder(), [], 0
    while True:
        i = text.find('{"launcher_selftest"', i)
        if i < 0:
            return recs
        obj, end = dec.raw_decode(text, i)
        recs.append(obj)
        i = end


def test_bench_launcher_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (torch.distributed.run on 127.0.0.1), each with its own
    RANK / LOCAL_RANK, all in one world of N; a world that is not --gpus is refused (the driver calls bench.py this way)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = _json_records(out.stdout)
    assert len(recs) == 2, out.stdout
    assert sorted(r["rank"] for r in recs) == [0, 1] and sorted(r["local_rank"] for r in recs) == [0, 1]
    assert all(r["world"] == 2 and r["gpus_arg"] == 2 and r["ranks_seen"] == [0, 1] and r["spawned_by_bench"] for r in recs)
    assert len({r["pid"] for r in recs}) == 2
    # under a launcher that started the wrong number of ranks the bench refuses to run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)
    # N = 1: no launcher involved
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=120)
    rec = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert one.returncode == 0 and rec["world"] == 1 and not rec["spawned_by_bench"]


def test_ranks_of_a_node_get_disjoint_host_thread_budgets():
    """Every rank builds, reads back and destroys its local-BA windows on host threads; the ranks of one node must not size their pools for the
    whole machine (10 host cores per GPU).  bench.py's host_budget() splits the usable cores (scheduler affinity, bounded by the
    cgroup's quota) into disjoint slices, one per local rank, pins the rank to its slice and sizes the window pool from it."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "CMS_BA_RELAXED_WAIT")}
    ncores = len(os.sched_getaffinity(0))
    if ncores < 4:
        pytest.skip("needs at least four cores")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = sorted(_json_records(out.stdout), key=lambda r: r["local_rank"])
    assert len(recs) == 2
    hosts = [r["host"] for r in recs]
    assert all(h["local_world_size"] == 2 and h["pinned"] for h in hosts)
    assert sum(h["thread_budget"] for h in hosts) <= ncores
    (a0, a1), (b0, b1) = hosts[0]["core_slice"], hosts[1]["core_slice"]
    assert a1 < b0 or b1 < a0, hosts                                   # the slices do not overlap
    # a pool thread per two cores; a rank with <= 4 cores switches to sleeping / blocking host waits by itself (bench.py main()), the others spin
    assert all(4 <= h["window_threads"] <= max(4, h["thread_budget"]) for h in hosts)
    assert all(h["host_waits"] == ("sleep" if h["thread_budget"] <= 4 else "spin") for h in hosts), hosts
    # a single rank keeps the whole affinity mask and is not pinned
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=120)
    h = _json_records(one.stdout)[0]["host"]
    assert h["local_world_size"] == 1 and not h["pinned"] and h["thread_budget"] >= min(ncores, h["cpu_quota_cores"] or ncores)
    # --window-threads overrides the pool size, not the budget
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest", "--window-threads", "5"], env=env, capture_output=True, text=True, timeout=120)
    assert _json_records(two.stdout)[0]["host"]["window_threads"] == 5


def test_a_rank_short_of_cores_plans_its_windows_on_the_device():
    """bench.py's policy for cms_ba_window.flags: eight ranks under a 16-core quota get two cores each and let the plan kernel make their windows' work lists;
    a rank with three or more cores keeps the host's plan; the environment overrides either way."""
    import bench
    assert bench.plans_on_device(2, env={}) and bench.plans_on_device(1, env={})
    assert not bench.plans_on_device(3, env={}) and not bench.plans_on_device(16, env={})
    assert bench.plans_on_device(16, env={"CMS_BENCH_PLAN_ON_DEVICE": "1"}) and not bench.plans_on_device(2, env={"CMS_BENCH_PLAN_ON_DEVICE": "0"})
    os.environ["LOCAL_WORLD_SIZE"] = "8"
    try:
        hb = bench.host_budget(8, 3, pin=False)
    finally:
        del os.environ["LOCAL_WORLD_SIZE"]
    assert bench.plans_on_device(hb["thread_budget"], env={}) == (hb["thread_budget"] <= 2)
