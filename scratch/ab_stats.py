"""A/B of the statistics-emitting / LayerNorm-folding epilogues against the plain ones on the UNet's level-0 shapes:
20 back-to-back launches in a captured graph, median of 20 replays."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd; stable_diffusion_amd.install_dropin()
from stable_diffusion_amd import engine as E
torch.set_grad_enabled(False)
dev = "cuda"
H = E.act_dtype()

def timeit(fn, reps=20, inner=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(ts))

def conv_case(hh, cin, cout):
    a = E.Act(torch.randn(1, hh * hh, cin, device=dev).to(H), 1, hh, hh, cin)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5))
    b = torch.nn.Parameter(torch.randn(cout, device=dev))
    res = {}
    for mode in (False, True):
        E.USE_COLSTATS = mode
        res[mode] = timeit(lambda: E.conv3x3(a, w, b))
    print("conv %dx%d %d->%d: plain %.2f us, with colstats %.2f us" % (hh, hh, cin, cout, res[False], res[True]))

def gemm_case(m, n, k, geglu=False):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w0 = (torch.randn(k, k, device=dev) / k ** 0.5).to(H)
    t = E.gemm(a, w0, k, rowstats=True)
    ln = torch.nn.LayerNorm(k).to(dev)
    W = torch.nn.Parameter(torch.randn(n, k, device=dev) / k ** 0.5)
    B = torch.nn.Parameter(torch.randn(n, device=dev))
    res = {}
    for mode in (False, True):
        E.USE_LNFOLD = mode
        res[mode] = timeit(lambda: E.gemm_ln(t, ln, (W,), bias=B, geglu=geglu))
    plain = timeit(lambda: E.gemm(t, E.packed_linear((W,), geglu=geglu), n, bias=B, geglu=geglu))
    rs = timeit(lambda: E.gemm(t, E.packed_linear((W,), geglu=geglu), n, bias=B, geglu=geglu, rowstats=True)) if not geglu else float("nan")
    print("gemm %dx%dx%d%s: LN kernel + gemm %.2f us, folded %.2f us; plain gemm alone %.2f us, + rowstats out %.2f us" %
          (m, n, k, " geglu" if geglu else "", res[False], res[True], plain, rs))

conv_case(96, 320, 320)
conv_case(96, 640, 320)
gemm_case(9216, 2560, 320, True)
gemm_case(9216, 960, 320)
gemm_case(9216, 320, 320)
gemm_case(2304, 5120, 640, True)
gemm_case(2304, 1920, 640)
gemm_case(576, 10240, 1280, True)
gemm_case(576, 3840, 1280)
