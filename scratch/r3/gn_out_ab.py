"""A/B of the split-K reduce fused with the following GroupNorm (sdmi_epilogue_ext.gn_out) against reduce kernel + gn_fused,
per shape, in HIP-graph replay (20 conv -> GroupNorm pairs per graph).  python scratch/r3/gn_out_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from stable_diffusion_amd import engine as E, _lib as L

DEV = "cuda"
torch.set_grad_enabled(False)


def graph_time(fn, reps=20, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        ts = []
        for _ in range(rounds):
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(ts))


def one(n, hh, cin, cout, keep):
    H = E.act_dtype()
    a = E.Act(torch.randn(n, hh * hh, cin, device=DEV).to(H), n, hh, hh, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) / (9 * cin) ** 0.5)
    B = torch.nn.Parameter(torch.randn(cout, device=DEV))
    gw, gb = torch.nn.Parameter(torch.ones(cout, device=DEV)), torch.nn.Parameter(torch.zeros(cout, device=DEV))
    r = {}
    for mode in (False, True, False, True):
        E.GN_OUT_FUSE = mode
        def f():
            o = E.conv3x3(a, W, B, gn_next=(gw, gb, 1e-5, True), keep_raw=keep)
            E.groupnorm(o, gw, gb, 1e-5, silu=True)
        r.setdefault(mode, []).append(graph_time(f))
    E.GN_OUT_FUSE = False
    t_conv = graph_time(lambda: E.conv3x3(a, W, B))
    print("n=%d %dx%d %d->%d keep_raw=%d: conv+reduce+gn_fused %s us | conv + reduce_gn %s us | conv+reduce alone %.1f us  [%s]" %
          (n, hh, hh, cin, cout, keep, ["%.1f" % v for v in r[False]], ["%.1f" % v for v in r[True]], t_conv, L.load().sdmi_last_kernel().decode()), flush=True)


if __name__ == "__main__":
    for keep in (0, 1):
        one(1, 24, 1280, 1280, keep)
        one(1, 12, 1280, 1280, keep)
        one(1, 24, 2560, 1280, keep)
        one(1, 12, 2560, 1280, keep)
        one(2, 24, 1280, 1280, keep)
