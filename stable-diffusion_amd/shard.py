"""Multi-GPU partitioning: independent images go one shard per GPU, weights replicated, no communication during
sampling, ONE all-gather of the uint8 outputs at the end.  One process per GPU under
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Second, optional mode for single-image LATENCY ("CFG halves on two GPUs"): ranks (2p, 2p+1) form a
pair, rank 2p evaluates the unconditional and rank 2p+1 the conditional half of every UNet step, and the pair exchanges
its (b, 4, h, w) fp32 noise predictions once per step (37 KB at 768x768: one small all-gather over one xGMI link).
Both ranks then apply the same update, so the latents stay identical without a broadcast.  Images are sharded over
PAIRS the way they are sharded over ranks in the default mode."""
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


def cfg_pair_layout(rank, world):
    """(pair index, number of pairs, role) of `rank` in CFG-split mode; role 0 = unconditional, 1 = conditional."""
    if world % 2:
        raise ValueError("CFG-split mode pairs ranks: world size %d is odd" % world)
    return rank // 2, world // 2, rank % 2


def new_cfg_pair_group(backend=None):
    """Create the process groups of all pairs (collective: every rank of the default group calls this once, in the same
    order) and return this rank's own."""
    rank, world = dist.get_rank(), dist.get_world_size()
    pair, pairs, _ = cfg_pair_layout(rank, world)
    mine = None
    for p in range(pairs):
        g = dist.new_group(ranks=[2 * p, 2 * p + 1], backend=backend)
        if p == pair:
            mine = g
    return mine


def exchange_cfg_halves(eps, role, group=None):
    """eps (2b, ...) contiguous: this rank has written its half [role*b, (role+1)*b); fetch the partner's half into the
    other one.  RCCL: one in-place all-gather on the device.  gloo (tests): staged through the host."""
    if dist.get_world_size(group) != 2:
        raise ValueError("a CFG pair has exactly two ranks")
    b = eps.shape[0] // 2
    mine = eps[role * b:(role + 1) * b]
    if eps.is_cuda and dist.get_backend(group) == "gloo":
        out = torch.empty(eps.shape, dtype=eps.dtype)
        dist.all_gather_into_tensor(out, mine.cpu(), group=group)
        eps[(1 - role) * b:(2 - role) * b].copy_(out[(1 - role) * b:(2 - role) * b])
        return eps
    dist.all_gather_into_tensor(eps, mine, group=group)
    return eps
