"""Multi-GPU plumbing of the hot path (torch.distributed only: RCCL on MI355X via backend "nccl", gloo on CPU for tests).

The path shards at stream granularity: a camera stream is an independent SLAM instance, so rank r
owns streams {s : s % world == r} and no data-path collective is needed.  The single exchange step is the assembly of the
trajectory on rank 0: every frame contributes one record [stream, ts, tx, ty, tz, qx, qy, qz, qw] (the TUM order of
System::SaveKeyFrameTrajectoryTUM, System.cpp:261-262, prefixed by the stream id).
"""
import numpy as np

RECORD = 9  # stream, ts, t(3), q(4)


def streams_of_rank(n_streams, world, rank):
    """stream s -> rank s mod world (8 streams on 1/2/4/8 GPUs -> 8/4/2/1 streams per GPU)."""
    return [s for s in range(n_streams) if s % world == rank]


def make_records(stream, ts, poses7):
    """poses7: (n, 7) = tx,ty,tz,qx,qy,qz,qw per frame; returns (n, RECORD) float64."""
    ts = np.asarray(ts, np.float64)
    rec = np.zeros((len(ts), RECORD), np.float64)
    rec[:, 0] = stream
    rec[:, 1] = ts
    rec[:, 2:] = np.asarray(poses7, np.float64).reshape(len(ts), 7)
    return rec


def gather_trajectory(local_records, device=None, dst=0):
    """Gather the (n_local, RECORD) records of every rank on `dst`; returns the (N, RECORD) trajectory sorted by
    (stream, ts) on dst and None elsewhere.  Ranks may hold different numbers of records (padded to the maximum)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        rec = np.asarray(local_records, np.float64).reshape(-1, RECORD)
        return rec[np.lexsort((rec[:, 1], rec[:, 0]))]
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    rec = torch.as_tensor(np.asarray(local_records, np.float64).reshape(-1, RECORD), device=device)
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((nmax, RECORD), dtype=torch.float64, device=device)
    pad[:rec.shape[0]] = rec
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    parts = [o[:int(c.item())].cpu().numpy() for o, c in zip(out, counts)]
    traj = np.concatenate(parts, 0) if parts else np.zeros((0, RECORD))
    return traj[np.lexsort((traj[:, 1], traj[:, 0]))]


def write_trajectory_tum(path, ts, Tcw):
    """System::SaveKeyFrameTrajectoryTUM (System.cpp:238-268) for rank 0's assembled trajectory: one line per key frame,
    `ts tx ty tz qx qy qz qw` = time stamp (6 decimals), camera centre -R^T t and quaternion of R^T (7 decimals), all through
    float like the reference.  Tcw: (n, 4, 4) float32 world->camera poses."""
    Tcw = np.asarray(Tcw, np.float32).reshape(-1, 4, 4)
    with open(path, "w") as f:
        for t_s, T in zip(np.asarray(ts, np.float64), Tcw):
            Rt = T[:3, :3].T.astype(np.float64)
            c = np.zeros(3, np.float32)
            for r in range(3):
                acc = np.float32(0)
                for k in range(3):
                    acc = np.float32(acc + np.float32(T[k, r] * T[k, 3]))
                c[r] = -acc
            q = _quat_from_matrix(Rt).astype(np.float32)
            f.write("%.6f %.7f %.7f %.7f %.7f %.7f %.7f %.7f\n" % (t_s, c[0], c[1], c[2], q[0], q[1], q[2], q[3]))


def _quat_from_matrix(m):
    """Eigen::Quaterniond(Matrix3d) (x, y, z, w), the conversion Converter::toQuaternion uses."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t; q[1] = (m[0, 2] - m[2, 0]) * t; q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t; q[j] = (m[j, i] + m[i, j]) * t; q[k] = (m[k, i] + m[i, k]) * t
    return q
