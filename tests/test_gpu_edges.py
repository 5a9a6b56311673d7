"""GPU: degenerate and ragged shapes of every hot-path entry point (one pixel, one token, one row, sizes one off the tile
sizes, saturating values) against the torch fp32 / float64 definition on inputs rounded to the storage type."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from stable_diffusion_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda"
H16 = E.act_dtype() if torch.cuda.is_available() else torch.float16
TOL = 8.0 if H16 == torch.bfloat16 else 1.0


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _bf(x):
    return x.to(H16).float()


def _check(got, ref, max_tol, l2_tol, what):
    m, l2 = rel_err(got.float().cpu().numpy(), ref.float().numpy())
    print("%s: max-rel %.3e  rel-L2 %.3e" % (what, m, l2))
    assert m <= max_tol * TOL and l2 <= l2_tol * TOL, (what, m, l2)


@pytest.mark.parametrize("n,cin,cout,h,w,stride", [(1, 64, 64, 1, 1, 1), (1, 8, 8, 1, 5, 1), (2, 72, 24, 3, 1, 1), (1, 64, 128, 2, 2, 2),
                                                   (1, 320, 320, 1, 129, 1), (1, 64, 64, 129, 1, 1), (5, 16, 8, 5, 7, 2)])
def test_conv3x3_degenerate_images(n, cin, cout, h, w, stride):
    """1-pixel-wide / 1-pixel-high images (every tap but one column/row is padding), rows one past a 128-row tile."""
    rs = np.random.RandomState(n * 1000 + h * 10 + w)
    x = _bf(_t(rs.randn(n, cin, h, w)))
    wt = _t(rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin))
    b = _t(rs.randn(cout))
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1)), _bf(wt), b, stride=stride)
    res = _bf(_t(rs.randn(*ref.shape)))
    out = E.conv3x3(E.from_nchw(x.to(DEV)), torch.nn.Parameter(wt.to(DEV)), b.to(DEV), residual=E.from_nchw(res.to(DEV)),
                    stride=stride, pad=(1, 1, 1, 1))
    _check(E.to_nchw(out), ref + res, 1.5e-3, 6e-4, "conv %dx%d s%d" % (h, w, stride))


@pytest.mark.parametrize("b,heads,nq,nkv", [(1, 1, 1, 1), (1, 2, 33, 65), (2, 1, 129, 64), (1, 3, 128, 63), (1, 1, 5, 77), (3, 2, 31, 1)])
def test_attention_ragged(b, heads, nq, nkv):
    rs = np.random.RandomState(nq * 100 + nkv)
    c = heads * 64
    q, k, v = (_bf(_t(rs.randn(b, n_, c))) for n_ in (nq, nkv, nkv))
    mk = lambda t, n_: E.Act(t.to(H16).to(DEV).contiguous(), b, 1, n_, c)
    out = E.attention(mk(q, nq), mk(k, nkv), mk(v, nkv), heads).buf.float().cpu()
    sp = lambda t, n_: t.reshape(b, n_, heads, 64).permute(0, 2, 1, 3).double()
    ref = torch.softmax(sp(q, nq) @ sp(k, nkv).transpose(-1, -2) / 8.0, dim=-1) @ sp(v, nkv)
    ref = ref.permute(0, 2, 1, 3).reshape(b, nq, c).float()
    _check(out, ref, 2e-3, 8e-4, "attention %dx%d" % (nq, nkv))


def test_attention_large_logits_stay_finite():
    """Scores of +-200 before the softmax (exp would overflow fp16 and, unscaled, fp32): the running-max form must not."""
    rs = np.random.RandomState(4)
    q = _bf(_t(rs.randn(1, 70, 64) * 14.0))
    k = _bf(_t(rs.randn(1, 90, 64) * 14.0))
    v = _bf(_t(rs.randn(1, 90, 64)))
    mk = lambda t: E.Act(t.to(H16).to(DEV).contiguous(), 1, 1, t.shape[1], 64)
    out = E.attention(mk(q), mk(k), mk(v), 1).buf.float().cpu()
    ref = (torch.softmax(q.double() @ k.double().transpose(-1, -2) / 8.0, dim=-1) @ v.double()).float()
    assert torch.isfinite(out).all()
    _check(out, ref, 4e-3, 2e-3, "attention large logits")


@pytest.mark.parametrize("n,c,h,w,silu", [(2, 32, 1, 1, True), (3, 64, 1, 2, False), (2, 320, 1, 1, True), (2, 128, 3, 5, True)])
def test_groupnorm_tiny_slices(n, c, h, w, silu):
    """One pixel: a group's statistics come from its C/32 channels only (variance 0 when C = 32: output = beta)."""
    rs = np.random.RandomState(c + h)
    x = _bf(_t(rs.randn(n, c, h, w) * 2 + 0.5))
    g, b = _t(1 + 0.1 * rs.randn(c)), _t(0.1 * rs.randn(c))
    ref = F.group_norm(x, 32, g, b, 1e-5)
    ref = F.silu(ref) if silu else ref
    y = E.groupnorm(E.from_nchw(x.to(DEV)), g.to(DEV), b.to(DEV), 1e-5, silu)
    _check(E.to_nchw(y), ref, 2e-3, 8e-4, "groupnorm %dx%dx%d" % (c, h, w))


@pytest.mark.parametrize("rows,c", [(1, 8), (1, 2560), (3, 328), (65, 64)])
def test_layernorm_single_rows(rows, c):
    rs = np.random.RandomState(rows + c)
    x, g, b = _bf(_t(rs.randn(1, rows, c) * 3 + 1)), _t(1 + 0.1 * rs.randn(c)), _t(0.1 * rs.randn(c))
    ref = F.layer_norm(x, (c,), g, b, 1e-5)
    y = E.layernorm(E.Act(x.to(H16).to(DEV), 1, 1, rows, c), g.to(DEV), b.to(DEV))
    _check(y.buf, ref, 2e-3, 8e-4, "layernorm %dx%d" % (rows, c))


@pytest.mark.parametrize("m,n,k", [(1, 8, 8), (2, 8, 2048), (129, 136, 72), (65, 8, 64), (1, 1280, 1280)])
def test_gemm_ragged(m, n, k):
    rs = np.random.RandomState(m + n + k)
    a, w, b = _bf(_t(rs.randn(m, k))), _bf(_t(rs.randn(n, k) / np.sqrt(k))), _t(rs.randn(n))
    out = E.gemm(E.Act(a.to(H16).to(DEV).reshape(1, m, k), 1, 1, m, k), w.to(H16).to(DEV).contiguous(), n, bias=b.to(DEV))
    _check(out.buf.reshape(m, n), a @ w.t() + b, 1.5e-3, 6e-4, "gemm %dx%dx%d" % (m, n, k))


def test_store_saturates_instead_of_overflowing():
    """A GEMM whose fp32 result exceeds the storage range must store the largest finite value, not inf (fp16 build)."""
    if H16 != torch.float16:
        pytest.skip("bf16 build: the range is fp32's")
    a = torch.full((1, 4, 64), 200.0, dtype=H16, device=DEV)
    w = torch.full((8, 64), 200.0, dtype=H16, device=DEV)
    out = E.gemm(E.Act(a, 1, 1, 4, 64), w, 8).buf.float()
    assert torch.isfinite(out).all() and (out == 65504.0).all()
