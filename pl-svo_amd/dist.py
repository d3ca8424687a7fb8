"""Multi-GPU sharding of independent streams: one process per GPU, static block assignment,
no data-path collective; the only exchange is an all-gather of the per-stream result poses.
Backend-agnostic (torch.distributed): "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

`timed_sharded_steps` is the control flow bench.py times -- warm-up, barrier, K steps of {local hot path, pose copy,
all-gather}, barrier, MAX over ranks -- kept here so that the CPU suite can run it with two gloo ranks and a recorded
result table in place of the kernels (tests/test_dist_cpu.py)."""
import time

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def stream_block(rank, world, streams_per_rank):
    """global stream indices owned by `rank` (weak scaling: every rank owns streams_per_rank streams)"""
    if not (0 <= rank < world) or streams_per_rank <= 0:
        raise ValueError("bad rank/world/streams_per_rank")
    first = rank * streams_per_rank
    return range(first, first + streams_per_rank)


def stream_seed(global_stream_index, base_seed=1234):
    """seed = 1234 + stream index, independent of how streams are sharded"""
    return base_seed + int(global_stream_index)


def rank_seeds(rank, world, streams_per_rank, base_seed=1234):
    """seeds of the streams `rank` owns: base_seed + rank * streams_per_rank + i"""
    return [stream_seed(g, base_seed) for g in stream_block(rank, world, streams_per_rank)]


def records_to_tensor(records, device="cpu"):
    """numpy structured array of plsvo_pose_record (pl-svo_amd/abi.py::POSE_RECORD_DTYPE, 96 B each) -> uint8 tensor [n, 96]: the shape
    the pose all-gather moves (pose + n_tracked, num_obs_pt, num_obs_ls, status per stream)"""
    import numpy as np
    from . import abi
    r = np.ascontiguousarray(records, dtype=abi.POSE_RECORD_DTYPE)
    return torch.from_numpy(r.view(np.uint8).reshape(r.shape[0], abi.POSE_RECORD_BYTES).copy()).to(device)


def tensor_to_records(t):
    """uint8 tensor [n, 96] (host or device) -> numpy structured array of plsvo_pose_record"""
    import numpy as np
    from . import abi
    a = np.ascontiguousarray(t.detach().cpu().numpy())
    return a.reshape(-1).view(abi.POSE_RECORD_DTYPE).copy()


def lost_streams(records, min_tracked=0, min_obs=0):
    """global indices (rank-major) of the streams a gathered record table marks as failed: the alignment's solver stopped on NaN or
    reported a device error, the pose optimiser had nothing to optimise, or the counts fall below the caller's thresholds (what
    src/frame_handler_mono.cpp:272-274, :327-335 test: img_align_n_tracked, sfba_n_edges_final = num_obs_pt + num_obs_ls)"""
    import numpy as np
    from . import abi
    r = records
    st = r["status"]
    bad = (st & (abi.REC_ALIGN_STOP | abi.REC_ALIGN_ERROR | abi.REC_POSEOPT_EMPTY)) != 0
    bad |= ((st & abi.REC_ALIGN) != 0) & (r["n_tracked"] < min_tracked)
    bad |= ((st & abi.REC_POSEOPT) != 0) & ((r["num_obs_pt"] + r["num_obs_ls"]) < min_obs)
    return [int(i) for i in np.nonzero(bad)[0]]


def gather_poses(local_poses, out=None):
    """All-gather [n_local, K] pose records -> [world * n_local, K], rank-major (K = 96 bytes of plsvo_pose_record as uint8; any 2-D
    tensor works).  Works un-initialised (single process) by returning the local tensor."""
    if world_size() == 1:
        if out is not None:
            out.copy_(local_poses)
            return out
        return local_poses
    world = world_size()
    if out is None:
        out = torch.empty((world * local_poses.shape[0], local_poses.shape[1]), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous())
    return out


def timed_sharded_steps(step_local, copy_local_poses, local_poses, steps, warmup, device_sync=lambda: None, before_timed=lambda: None,
                        gather=None, force_gather=False):
    """The timed region of one benchmark run over a sharded batch of independent streams.

    step_local():            enqueue one pass of the hot path over this rank's streams
    copy_local_poses(t):     enqueue the copy of this rank's result poses into tensor t ([n_local, 7]), on the SAME stream
                             as step_local's kernels, so that the collective that follows is ordered after them
    local_poses:             the [n_local, 7] tensor (device of the backend: HBM for nccl, host for gloo)
    device_sync():           wait for everything enqueued (torch.cuda.synchronize on a GPU, nothing on the CPU)
    before_timed():          called once between the warm-up and the timed steps (bench.py switches its kernel timers on here)

    gather(local, out):      the all-gather of the pose records.  On GPUs this is the C ABI's plsvo_gather_poses (ncclAllGather on the
                             library's stream, the ONE gather path of the product); None = torch.distributed's all-gather, which is
                             what the gloo ranks of the CPU suite use in its place
    force_gather:            run the copy + gather with a single rank as well (bench.py --dist-selftest: the N>1 code on a 1-GPU box)

    One step = step_local, and with more than one rank: copy_local_poses + the all-gather of the poses (the only collective).
    Returns (elapsed seconds of the K timed steps, MAX over ranks; the gathered [world * n_local, 7] tensor or None)."""
    world = world_size()
    exchange = world > 1 or force_gather
    gathered = None
    if exchange:
        gathered = torch.empty((world * local_poses.shape[0], local_poses.shape[1]), dtype=local_poses.dtype, device=local_poses.device)
    if gather is None:
        gather = lambda local, out: gather_poses(local, out=out)

    def step():
        step_local()
        if exchange:
            copy_local_poses(local_poses)
            gather(local_poses, gathered)

    for _ in range(warmup):
        step()
    device_sync()
    before_timed()
    if world > 1:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=local_poses.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if exchange:
        # the gathered table must hold this rank's poses in this rank's block (guards the copy -> all-gather ordering)
        r, n = rank(), local_poses.shape[0]
        if not torch.equal(gathered[r * n:(r + 1) * n], local_poses):
            raise RuntimeError("all-gather returned stale poses for this rank's block: the collective was not ordered after the pose copy")
    return elapsed, gathered
