"""Multi-GPU partitioning: independent images go one shard per GPU, weights replicated, no communication during
sampling, ONE all-gather of the uint8 outputs at the end.  One process per GPU under
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Images [lo, hi) of rank `rank`: contiguous, sizes differ by at most one, lower ranks take the remainder."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_inputs(inputs, rank, world):
    """Slice every per-image tensor (dim 0 = image) of a make_inputs()-style dict; lists are sliced elementwise."""
    total = inputs["x_T"].shape[0]
    lo, hi = shard_bounds(total, rank, world)
    out = {}
    for k, v in inputs.items():
        out[k] = [t[lo:hi] for t in v] if isinstance(v, (list, tuple)) else v[lo:hi]
    return out


def all_gather_images(local, total, group=None):
    """Gather per-rank (b_local, ...) uint8 tensors into (total, ...) on every rank, in image order."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)      # a one-rank group still goes through the collective (same code path as N ranks)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
