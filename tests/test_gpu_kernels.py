"""GPU: every C-ABI entry point against the fp32 oracle / torch fp32 definition on seeded inputs.
bf16 storage + fp32 accumulation: tolerances are stated per test as (max-abs / max|ref|, relative L2)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sd_oracle as O
from conftest import rel_err
from stable_diffusion_amd import engine as E, _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rs(seed):
    return np.random.RandomState(seed)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


H16 = E.act_dtype() if torch.cuda.is_available() else torch.float16


def _bf(x):
    """fp32 -> h16 -> fp32 rounding (what the kernels see as input)."""
    return x.to(H16).float()


def _act(x_nchw):
    return E.from_nchw(x_nchw.to(DEV))


TOL = 8.0 if H16 == torch.bfloat16 else 1.0      # bf16 build (-DSDMI_BF16) has 3 fewer mantissa bits


def _check(got, ref, max_tol, l2_tol, what):
    """fp16 storage, fp32 accumulate: one output rounding is ~2e-4 rel-L2; gates are ~3x the measured error."""
    max_tol, l2_tol = max_tol * TOL, l2_tol * TOL
    m, l2 = rel_err(got.float().cpu().numpy(), ref.numpy())
    print("%s: max-rel %.3e  rel-L2 %.3e" % (what, m, l2))
    assert m <= max_tol and l2 <= l2_tol, (what, m, l2)


def test_layout_roundtrip():
    x = _bf(_t(_rs(0).randn(3, 40, 7, 9)))
    a = _act(x)
    assert torch.equal(E.to_nchw(a).cpu(), x)      # bf16-representable values survive exactly


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (200, 320, 640), (77, 64, 1024), (1, 8, 8), (300, 72, 136), (4608, 640, 640)])
def test_gemm_bias_residual(m, n, k):
    rs = _rs(m + n + k)
    a, w, b, r = _bf(_t(rs.randn(m, k))), _bf(_t(rs.randn(n, k) / np.sqrt(k))), _t(rs.randn(n)), _bf(_t(rs.randn(m, n)))
    ref = a @ w.t() + b + r
    A = E.Act(a.to(H16).to(DEV).reshape(1, m, k), 1, 1, m, k)
    R = E.Act(r.to(H16).to(DEV).reshape(1, m, n), 1, 1, m, n)
    wp = w.to(H16).to(DEV).contiguous()
    out = E.gemm(A, wp, n, bias=b.to(DEV), residual=R)
    _check(out.buf.reshape(m, n), ref, 1.5e-3, 6e-4, "gemm %dx%dx%d" % (m, n, k))


def test_gemm_pack_and_geglu():
    rs = _rs(5)
    m, c = 150, 64
    a, w, b = _bf(_t(rs.randn(m, c))), _t(rs.randn(8 * c, c) / 8), _t(rs.randn(8 * c))
    wp = E.packed_linear((torch.nn.Parameter(w.to(DEV)),), geglu=True)
    h = a @ _bf(w).t() + b
    x, g = h.chunk(2, dim=-1)
    ref = x * F.gelu(g)
    A = E.Act(a.to(H16).to(DEV).reshape(1, m, c), 1, 1, m, c)
    out = E.gemm(A, wp, 8 * c, bias=b.to(DEV), geglu=True)
    assert out.c == 4 * c
    _check(out.buf.reshape(m, 4 * c), ref, 1.5e-3, 6e-4, "geglu")


@pytest.mark.parametrize("case", ["s1", "s2", "s2enc", "up", "cout3_nchw", "odd"])
def test_conv3x3(case):
    rs = _rs(hash(case) % 1000)
    n, cin, cout, h, w = 2, 64, 128, 12, 10
    kw = dict(stride=1, pad=(1, 1, 1, 1), upsample=False)
    if case == "s2":
        kw.update(stride=2)
    elif case == "s2enc":
        kw.update(stride=2, pad=(0, 1, 0, 1))
    elif case == "up":
        kw.update(upsample=True)
    elif case == "cout3_nchw":
        cout = 3
    elif case == "odd":
        n, cin, cout, h, w = 3, 72, 40, 7, 5
    x = _bf(_t(rs.randn(n, cin, h, w)))
    wt = _t(rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin))
    b = _t(rs.randn(cout))
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if kw["upsample"] else x
    ref = F.conv2d(F.pad(xi, kw["pad"]), _bf(wt), b, stride=kw["stride"])
    temb = _t(rs.randn(n, cout))
    res = _bf(_t(rs.randn(*ref.shape)))
    W, B = torch.nn.Parameter(wt.to(DEV)), b.to(DEV)
    if case == "cout3_nchw":
        out = E.conv3x3(_act(x), W, B, out_nchw=True, **kw)
        _check(out, ref, 1.5e-3, 6e-4, "conv " + case)
        return
    ref = ref + temb[:, :, None, None] + res
    out = E.conv3x3(_act(x), W, B, rowbias=temb.to(DEV), ldrb=cout, residual=_act(res), **kw)
    _check(E.to_nchw(out), ref, 1.5e-3, 6e-4, "conv " + case)


@pytest.mark.parametrize("cin,cout", [(9, 320), (4, 512), (3, 128)])
def test_conv3x3_in(cin, cout):
    rs = _rs(cin)
    x, wt, b = _t(rs.randn(2, cin, 9, 11)), _t(rs.randn(cout, cin, 3, 3) / 6), _t(rs.randn(cout))
    ref = F.conv2d(x, wt, b, padding=1)
    a = E.new_act(2, 9, 11, cout, DEV)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    L.call("sdmi_conv3x3_in", xd.data_ptr(), 2, cin, 9, 11, wd.data_ptr(), bd.data_ptr(), cout, a.ptr, a.ld, E._stream())
    _check(E.to_nchw(a), ref, 1.5e-3, 6e-4, "conv_in %d->%d" % (cin, cout))


@pytest.mark.parametrize("c,hw,silu,eps", [(320, (12, 12), True, 1e-5), (64, (5, 7), False, 1e-6), (1920, (6, 6), True, 1e-5), (128, (40, 40), True, 1e-6), (2560, (3, 3), True, 1e-5),
                                            (320, (128, 128), True, 1e-5), (640, (48, 48), True, 1e-5), (320, (96, 96), False, 1e-6)])
def test_groupnorm(c, hw, silu, eps):
    """Small (sample, group) slices run the fused single-launch kernel, (320, 128x128) the stats + apply pair."""
    rs = _rs(c)
    x = _bf(_t(rs.randn(2, c, *hw) * 2 + 0.5))
    g, b = _t(1 + 0.1 * rs.randn(c)), _t(0.1 * rs.randn(c))
    ref = F.group_norm(x, 32, g, b, eps)
    ref = F.silu(ref) if silu else ref
    y = E.groupnorm(_act(x), g.to(DEV), b.to(DEV), eps, silu)
    _check(E.to_nchw(y), ref, 1.5e-3, 6e-4, "groupnorm c=%d" % c)


@pytest.mark.parametrize("c,rows", [(320, 37), (640, 130), (1280, 9), (2560, 5)])
def test_layernorm(c, rows):
    rs = _rs(c + 1)
    x, g, b = _bf(_t(rs.randn(1, rows, c) * 3 + 1)), _t(1 + 0.1 * rs.randn(c)), _t(0.1 * rs.randn(c))
    ref = F.layer_norm(x, (c,), g, b, 1e-5)
    a = E.Act(x.to(H16).to(DEV), 1, 1, rows, c)
    y = E.layernorm(a, g.to(DEV), b.to(DEV))
    _check(y.buf, ref, 1.5e-3, 6e-4, "layernorm c=%d" % c)


@pytest.mark.parametrize("name,heads", [("cross", 4), ("self", 3)])
def test_attention_kat(golden, name, heads):
    """fp64-definition known-answer vectors (tests/golden/g5_attention.npz); (B*H, N, 64) -> token-major views."""
    g = golden("g5_attention")
    q, k, v, o = (_t(g[name + "_" + s]) for s in "qkvo")
    tok = lambda t: t.permute(1, 0, 2).reshape(1, t.shape[1], heads * 64)          # (1, N, H*64)
    ref = O.attention(_bf(q), _bf(k), _bf(v))
    mk = lambda t: E.Act(tok(t).to(H16).to(DEV).contiguous(), 1, 1, t.shape[1], heads * 64)
    out = E.attention(mk(q), mk(k), mk(v), heads)
    _check(out.buf, tok(ref), 1.5e-3, 6e-4, "attention " + name)
    _check(out.buf, tok(o), 2.5e-3, 1.2e-3, "attention vs fp64 KAT " + name)


def test_attention_rescale_branch_and_batch():
    """A spiked key late in the sequence forces the online-softmax running max to jump (rare-branch test), batch 2,
    strided q/k/v views of one fused buffer, nq/nkv not multiples of the tile sizes."""
    rs = _rs(3)
    b, heads, nq, nk = 2, 2, 150, 333
    qkv = _t(rs.randn(b, nk, 3 * heads * 64))
    qkv[:, 300, heads * 64: 2 * heads * 64] *= 12.0          # key row 300 dominates
    qkv = _bf(qkv)
    buf = E.Act(qkv.to(H16).to(DEV).contiguous(), b, 1, nk, 3 * heads * 64)
    q, k, v = buf.cols(0, heads * 64), buf.cols(heads * 64, heads * 64), buf.cols(2 * heads * 64, heads * 64)
    qa = E.Act(q.buf, b, 1, nk, heads * 64, q.off)
    out = E.attention(qa, k, v, heads)
    sp = lambda t: t.reshape(b, nk, heads, 64).permute(0, 2, 1, 3).reshape(b * heads, nk, 64)
    qq, kk, vv = (sp(qkv[..., i * heads * 64:(i + 1) * heads * 64]) for i in range(3))
    ref = O.attention(qq.double(), kk.double(), vv.double()).float().reshape(b, heads, nk, 64).permute(0, 2, 1, 3).reshape(b, nk, heads * 64)
    _check(out.buf, ref, 1.5e-3, 6e-4, "attention spiked")
    assert torch.isfinite(out.buf.float()).all()


def test_small_linear_and_timestep_embedding(golden):
    rs = _rs(4)
    x, w, b = _t(rs.randn(4, 1280)), _t(rs.randn(650, 1280) / 36), _t(rs.randn(650))
    ref = F.silu(x) @ _bf(w).t() + b
    out = E.linear_small_m(x.to(DEV), w.to(H16).to(DEV), b.to(DEV), silu_in=True)
    _check(out, ref, 1e-4, 1e-4, "linear_small_m")
    g = golden("g0_timestep_embedding")
    emb = E.timestep_embedding(torch.from_numpy(g["t"]).to(DEV), 320)
    # the reference's fp32 exp() of the frequency table is host dependent in the last ulp, worth <= 6e-5 in cos(t*f)
    assert np.abs(emb.cpu().numpy() - g["emb"]).max() <= 1e-4


def test_cfg_ddim_bit_exact():
    """fp32 elementwise chain in the reference's operation order: bit-identical to the oracle's IEEE-fp32 numpy
    restatement (== the torch restatement in the build container), within 2 ulp-ish of torch on any host."""
    rs = _rs(8)
    b = 3
    x, eu, ec, nz = (_t(rs.randn(b, 4, 16, 16)) for _ in range(4))
    sch = O.ddim_schedule(50)
    for index in (0, 17, 49):
        a_t, a_p, sg, s1 = sch["alphas"][index], sch["alphas_prev"][index], sch["sigmas"][index], sch["sqrt_one_minus_alphas"][index]
        ref, ref0 = (torch.from_numpy(v) for v in O.cfg_ddim_update_np(x, eu, ec, 9.0, a_t, a_p, sg, s1, nz))
        tref, _ = O.cfg_ddim_update(x, eu, ec, 9.0, a_t, a_p, sg, s1, nz)
        assert (tref - ref).abs().max() <= 1e-6 * ref.abs().max()
        f = lambda v: float(torch.full((1,), v).item())
        p0 = torch.empty((b, 4, 16, 16), device=DEV)
        xc = torch.zeros((2 * b, 9, 16, 16), device=DEV)
        out = E.cfg_ddim_step(torch.cat([eu, ec]).to(DEV), x.to(DEV), nz.to(DEV), 9.0, f(s1), f(a_t), f(a_p), f(sg), pred_x0=p0, xc_out=xc)
        assert torch.equal(out.cpu(), ref), index
        assert torch.equal(p0.cpu(), ref0), index
        assert torch.equal(xc[:b, :4].cpu(), ref) and torch.equal(xc[b:, :4].cpu(), ref) and float(xc[:, 4:].abs().max()) == 0.0


def test_conv_rowbias_tile_rows_past_m():
    """M = 2*144 = 288 rows on 64-row tiles: the last tile's second wave starts past M.  Its epilogue prefetch must not
    index a per-sample bias row that does not exist (it did once: an out-of-range read that only faulted when the
    embedding buffer ended an allocation)."""
    rs = _rs(91)
    n, cin, cout, h = 2, 64, 256, 12
    x = _bf(_t(rs.randn(n, cin, h, h)))
    wt, b = _t(rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)), _t(rs.randn(cout))
    temb, res = _t(rs.randn(n, cout)), _bf(_t(rs.randn(n, cout, h, h)))
    ref = F.conv2d(x, _bf(wt), b, padding=1) + temb[:, :, None, None] + res
    out = E.to_nchw(E.conv3x3(_act(x), torch.nn.Parameter(wt.to(DEV)), b.to(DEV), rowbias=temb.to(DEV), ldrb=cout, residual=_act(res)))
    _check(out, ref, 1.5e-3, 6e-4, "conv rowbias, rows past M")


@pytest.mark.parametrize("tile,splits", [(1, 1), (2, 1), (1, 3), (2, 2)])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 128, 16, 16), (1, 320, 192, 13, 9), (3, 72, 64, 7, 20), (2, 128, 320, 24, 24)])
def test_conv_shared_window_kernel(n, cin, cout, h, w, tile, splits):
    """The shared-window kernel (128-row tiles, stride 1): parity against torch, and the SAME bits as the generic im2col
    kernel on the same tile / split (same K order, same arithmetic).  Shapes: rows that wrap image rows inside a tile,
    M not a multiple of 128, images that end inside a tile, Cin with a partial 64-channel chunk, 5 chunks."""
    rs = _rs(n * cin + cout + h)
    x = _bf(_t(rs.randn(n, cin, h, w)))
    wt, b = _t(rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)), _t(rs.randn(cout))
    temb, res = _t(rs.randn(n, cout)), _bf(_t(rs.randn(n, cout, h, w)))
    ref = F.conv2d(x, _bf(wt), b, padding=1) + temb[:, :, None, None] + res
    W, B, TE = torch.nn.Parameter(wt.to(DEV)), b.to(DEV), temb.to(DEV)
    lib = L.load()
    outs = {}
    try:
        for win in (1, 0):
            lib.sdmi_tune_override(tile, splits)
            lib.sdmi_tune_conv_window(win)
            o = E.conv3x3(_act(x), W, B, rowbias=TE if h * w >= 128 else None, ldrb=cout, residual=_act(res))
            outs[win] = (E.to_nchw(o), lib.sdmi_last_kernel().decode())
    finally:
        lib.sdmi_tune_override(-1, -1)
        lib.sdmi_tune_conv_window(-1)
    assert outs[1][1].startswith("conv_win") and outs[0][1].startswith("igemm_kernel"), (outs[1][1], outs[0][1])
    if h * w < 128:
        ref = ref - temb[:, :, None, None]
    _check(outs[1][0], ref, 1.5e-3, 6e-4, "shared-window conv")
    assert torch.equal(outs[1][0], outs[0][0]), "shared-window and generic kernels differ"


def test_splitk_conv_and_gemm_batch_invariant():
    """Under-filled grids run split-K (fp32 partials + a fixed-order reduce): parity holds and one image gives the
    same bits alone or inside a batch (the split factor depends on the per-sample shape only)."""
    rs = _rs(77)
    n, cin, cout, h = 2, 512, 256, 6
    assert L.load().sdmi_conv3x3_ws_bytes(n, cin, cout, h, h) > 0
    x = _bf(_t(rs.randn(n, cin, h, h)))
    wt, b = _t(rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)), _t(rs.randn(cout))
    temb, res = _t(rs.randn(n, cout)), _bf(_t(rs.randn(n, cout, h, h)))
    ref = F.conv2d(x, _bf(wt), b, padding=1) + temb[:, :, None, None] + res
    W, B, TE = torch.nn.Parameter(wt.to(DEV)), b.to(DEV), temb.to(DEV)
    out = E.to_nchw(E.conv3x3(_act(x), W, B, rowbias=TE, ldrb=cout, residual=_act(res)))
    _check(out, ref, 1.5e-3, 6e-4, "split-K conv")
    one = E.to_nchw(E.conv3x3(_act(x[1:]), W, B, rowbias=TE[1:].contiguous(), ldrb=cout, residual=_act(res[1:])))
    assert torch.equal(one, out[1:])
    m, nn, k = 288, 1280, 5120
    assert L.load().sdmi_gemm_ws_bytes(m, nn, k, 1, 144, 0) > 0
    a, w2 = _bf(_t(rs.randn(m, k))), _bf(_t(rs.randn(nn, k) / np.sqrt(k)))
    A = E.Act(a.to(H16).to(DEV).reshape(2, 144, k), 2, 12, 12, k)
    o = E.gemm(A, w2.to(H16).to(DEV).contiguous(), nn)
    _check(o.buf.reshape(m, nn), a @ w2.t(), 1.5e-3, 6e-4, "split-K gemm")


def test_measurement_probes():
    """bench.py's measured-peak probes run and give figures of the right order (MI355X: ~2 PFLOP/s, ~4-5 TB/s copy)."""
    import ctypes
    from stable_diffusion_amd import _lib
    st = torch.cuda.current_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    scratch = torch.zeros(16, device=DEV)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.call("sdmi_probe_mfma", scratch.data_ptr(), 512, 100, sp)
    e0.record(st); _lib.call("sdmi_probe_mfma", scratch.data_ptr(), 512, 5000, sp); e1.record(st)
    torch.cuda.synchronize()
    tf = 512 * 4 * 5000 * 8 * (2.0 * 32 * 32 * 16) / (e0.elapsed_time(e1) * 1e-3) / 1e12
    n = 1 << 28
    src = torch.arange(n // 8, dtype=torch.int64, device=DEV)
    dst = torch.zeros_like(src)
    _lib.call("sdmi_probe_copy", src.data_ptr(), dst.data_ptr(), n, sp)
    e0.record(st); _lib.call("sdmi_probe_copy", src.data_ptr(), dst.data_ptr(), n, sp); e1.record(st)
    torch.cuda.synchronize()
    gbs = 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print("probe: %.0f TFLOP/s MFMA issue, %.0f GB/s copy" % (tf, gbs))
    assert torch.equal(src, dst)
    assert scratch.abs().max().item() == 0.0
    assert 500.0 < tf < 6000.0 and 500.0 < gbs < 20000.0
