"""CPU, world_size 2 over gloo: image sharding + the single all-gather reproduce the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stable_diffusion_amd.shard import (shard_bounds, shard_inputs, all_gather_images, cfg_pair_layout, new_cfg_pair_group,
                                        exchange_cfg_halves)
from stable_diffusion_amd.synth import make_inputs


def _fake_pipeline(inp):
    """Stand-in for sample(): a per-image deterministic function of every per-image input."""
    acc = inp["x_T"].sum(dim=(1, 2, 3)) + inp["c_concat"].mean(dim=(1, 2, 3)) + inp["cond"].std(dim=(1, 2)) - inp["uncond"].amax(dim=(1, 2))
    for n in inp["noises"]:
        acc = acc + n.abs().sum(dim=(1, 2, 3))
    img = (acc[:, None, None, None] * torch.arange(1, 13).reshape(1, 2, 2, 3)).remainder(255)
    return img.to(torch.uint8)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inp = make_inputs(total, 8, 3, seed=7)
    local = _fake_pipeline(shard_inputs(inp, rank, world))
    full = all_gather_images(local, total)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bounds_cover_and_balance():
    for total in (1, 2, 5, 16, 17):
        for world in (1, 2, 4, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_equals_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    for total in (4, 5):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
        [p.start() for p in procs]
        res = dict(q.get(timeout=120) for _ in range(2))
        [p.join(timeout=60) for p in procs]
        ref = _fake_pipeline(make_inputs(total, 8, 3, seed=7)).numpy()
        assert res[0].shape == ref.shape
        np.testing.assert_array_equal(res[0], ref)
        np.testing.assert_array_equal(res[1], ref)


def _fake_half(x, ctx):
    """Stand-in for one CFG half of a UNet step: depends on the latent and on that half's prompt only."""
    return torch.sin(x * 3.0) + ctx.mean(dim=(1, 2))[:, None, None, None]


def _cfg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = new_cfg_pair_group()
    pair, pairs, role = cfg_pair_layout(rank, world)
    assert dist.get_rank(group) == role and dist.get_world_size(group) == 2
    inp = shard_inputs(make_inputs(3, 8, 3, seed=11), pair, pairs)      # images are sharded over PAIRS
    x, b = inp["x_T"].clone(), inp["x_T"].shape[0]
    for n in inp["noises"]:
        eps = torch.full((2 * b,) + tuple(x.shape[1:]), float("nan"))
        eps[role * b:(role + 1) * b] = _fake_half(x, (inp["uncond"], inp["cond"])[role])
        exchange_cfg_halves(eps, role, group)
        x = x - 0.1 * (eps[:b] + 7.5 * (eps[b:] - eps[:b])) + 0.01 * n     # both ranks of a pair: the same update
    q.put((rank, x.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_layout():
    assert [cfg_pair_layout(r, 4) for r in range(4)] == [(0, 2, 0), (0, 2, 1), (1, 2, 0), (1, 2, 1)]
    try:
        cfg_pair_layout(0, 3)
        assert False
    except ValueError:
        pass


def test_cfg_halves_on_rank_pairs_equal_single_process():
    """World 4 = two pairs: each pair samples its image shard with the unconditional half on its even and the
    conditional half on its odd rank, exchanging the halves every step; equal to one process doing both halves."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_worker, args=(r, 4, port, q)) for r in range(4)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(4))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    np.testing.assert_array_equal(res[0], res[1])
    np.testing.assert_array_equal(res[2], res[3])
    for pair in range(2):       # per shard: the CPU stand-in's vectorised sin() is not batch-invariant to the last ulp
        inp = shard_inputs(make_inputs(3, 8, 3, seed=11), pair, 2)
        x = inp["x_T"].clone()
        for n in inp["noises"]:
            eu, ec = _fake_half(x, inp["uncond"]), _fake_half(x, inp["cond"])
            x = x - 0.1 * (eu + 7.5 * (ec - eu)) + 0.01 * n
        np.testing.assert_array_equal(res[2 * pair], x.numpy())
