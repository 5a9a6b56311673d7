"""GPU: sdmi_groupnorm_static (the entry point that allocates nothing: scratch in a static device array) against
sdmi_groupnorm with caller-allocated buffers, bit for bit, on two calls in a row, the second with a smaller input
(the static array keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from stable_diffusion_amd import engine as E, _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(n, h, w, c, seed, silu):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.randn(n, h * w, c).astype(np.float32) * 1.7 + 0.3).to(E.act_dtype()).to(DEV)
    gamma = torch.from_numpy(rs.randn(c).astype(np.float32)).to(DEV)
    beta = torch.from_numpy(rs.randn(c).astype(np.float32)).to(DEV)
    return E.Act(x, n, h, w, c), gamma, beta, silu


def _run_both(a, gamma, beta, silu, eps=1e-5):
    st = torch.cuda.current_stream().cuda_stream
    flags = L.F_SILU if silu else 0
    ref = E.groupnorm(a, gamma, beta, eps, silu)                      # sdmi_groupnorm, torch-allocated scratch
    y = torch.full_like(a.buf, 7.0)
    L.call("sdmi_groupnorm_static", a.ptr, a.ld, y.data_ptr(), y.shape[-1], a.n, a.hw, a.c, 32, gamma.data_ptr(),
           beta.data_ptr(), eps, flags, st)
    torch.cuda.synchronize()
    return ref.buf, y


def test_groupnorm_static_matches_groupnorm_twice():
    lib = L.load()
    assert lib.sdmi_groupnorm_static_max_bytes() == 1 << 20
    # both shapes take the stats + apply path (a (sample, group) slice too large for the fused kernel), so the static
    # array is really used: first the larger input, then a smaller one over the stale contents
    big = _case(2, 96, 96, 320, 1, True)
    small = _case(1, 48, 48, 640, 2, False)
    for case in (big, small):
        ref, got = _run_both(*case)
        assert torch.equal(ref.view(torch.int16), got.view(torch.int16))
    # and the order reversed within the same process (small first leaves fewer words behind)
    for case in (small, big):
        ref, got = _run_both(*case)
        assert torch.equal(ref.view(torch.int16), got.view(torch.int16))


def test_groupnorm_static_small_fused_shape():
    # a slice that fits the fused single-launch kernel needs no scratch at all: same bits too
    ref, got = _run_both(*_case(2, 12, 12, 1280, 3, True))
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16))


def test_groupnorm_static_rejects_oversize():
    a, gamma, beta, _ = _case(1, 8, 8, 64, 4, False)
    st = torch.cuda.current_stream().cuda_stream
    y = torch.empty_like(a.buf)
    # claims 64 samples of 512x512x2560: far beyond 1 MiB of scratch; must fail before touching memory
    with pytest.raises(L.SdmiError, match="static buffer holds"):
        L.call("sdmi_groupnorm_static", a.ptr, 2560, y.data_ptr(), 2560, 64, 512 * 512, 2560, 32, gamma.data_ptr(),
               beta.data_ptr(), 1e-5, 0, st)


def test_groupnorm_static_under_graph_capture():
    """For the callers it exists for: after sdmi_groupnorm_static_init() the call is capturable (the symbol lookup is not
    a stream operation and has already happened), and a replay of the captured graph writes the bits of the eager call."""
    lib = L.load()
    assert lib.sdmi_groupnorm_static_init() == 0
    a, gamma, beta, silu = _case(2, 48, 48, 320, 5, True)
    ref = E.groupnorm(a, gamma, beta, 1e-5, silu).buf
    y = torch.full_like(a.buf, 3.0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            L.call("sdmi_groupnorm_static", a.ptr, a.ld, y.data_ptr(), y.shape[-1], a.n, a.hw, a.c, 32, gamma.data_ptr(),
                   beta.data_ptr(), 1e-5, L.F_SILU, torch.cuda.current_stream().cuda_stream)
    y.fill_(3.0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int16), y.view(torch.int16))
