"""Multi-GPU sharding of independent streams: one process per GPU, static block assignment,
no data-path collective; the only exchange is an all-gather of the per-stream result poses.
Backend-agnostic (torch.distributed): "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def stream_block(rank, world, streams_per_rank):
    """global stream indices owned by `rank` (weak scaling: every rank owns streams_per_rank streams)"""
    if not (0 <= rank < world) or streams_per_rank <= 0:
        raise ValueError("bad rank/world/streams_per_rank")
    first = rank * streams_per_rank
    return range(first, first + streams_per_rank)


def stream_seed(global_stream_index, base_seed=1234):
    """seed = 1234 + stream index, independent of how streams are sharded"""
    return base_seed + int(global_stream_index)


def gather_poses(local_poses, out=None):
    """All-gather [n_local, 7] pose records -> [world * n_local, 7], rank-major.  Works un-initialised
    (single process) by returning the local tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if out is not None:
            out.copy_(local_poses)
            return out
        return local_poses
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local_poses.shape[0], local_poses.shape[1]), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous())
    return out
