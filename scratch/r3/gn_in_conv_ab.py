"""A/B of north-star item (b): GroupNorm-apply + SiLU inside the conv (sdmi_epilogue_ext.gn_in) against sdmi_groupnorm_apply +
conv, on the VAE's 128-channel level, and the whole decode.  python scratch/r3/gn_in_conv_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from stable_diffusion_amd import engine as E, _lib as L

DEV = "cuda"


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def one(n, hh, cin, cout):
    rs = np.random.RandomState(1)
    H = E.act_dtype()
    x = torch.randn(n, hh * hh, cin, device=DEV).to(H)
    a0 = E.Act(x, n, hh, hh, cin)
    w0 = (torch.randn(cin, cin, device=DEV) / cin ** 0.5).to(H).contiguous()
    a = E.gemm(a0, w0, cin, colstats=True)
    gn = torch.nn.GroupNorm(32, cin).to(DEV)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) / (9 * cin) ** 0.5)
    B = torch.nn.Parameter(torch.randn(cout, device=DEV))
    out = E.new_act(n, hh, hh, cout, DEV)
    res = {}
    for mode in (False, True, False, True):
        E.GN_IN_CONV = mode
        def f():
            h = E.groupnorm(a, gn.weight, gn.bias, gn.eps, silu=True, conv_cout=cout)
            E.conv3x3(h, W, B, out=out)
        res.setdefault(mode, []).append(timed(f))
    E.GN_IN_CONV = False
    h = E.groupnorm(a, gn.weight, gn.bias, gn.eps, silu=True)
    t_conv = timed(lambda: E.conv3x3(h, W, B, out=out))
    print("n=%d %dx%d %d->%d: GN(finalize+apply)+conv %s us | finalize + fused conv %s us | plain conv alone %.1f us" %
          (n, hh, hh, cin, cout, ["%.1f" % v for v in res[False]], ["%.1f" % v for v in res[True]], t_conv), flush=True)


def decode(b):
    import stable_diffusion_amd
    stable_diffusion_amd.install_dropin()
    from AutoencoderKL import AutoencoderKL
    from stable_diffusion_amd.synth import fill_state_dict
    vae = AutoencoderKL()
    fill_state_dict(vae.state_dict(), seed=5, fast=True)
    vae = vae.to(DEV).eval()
    z = torch.randn(b, 4, 96, 96, device=DEV)
    r = {}
    for mode in (False, True, False, True):
        E.GN_IN_CONV = mode
        r.setdefault(mode, []).append(timed(lambda: vae.decode_image_u8(z), iters=5, warm=2) / 1e3)
    E.GN_IN_CONV = False
    print("decode 768x768 b=%d: unfused %s ms | fused %s ms" % (b, ["%.2f" % v for v in r[False]], ["%.2f" % v for v in r[True]]), flush=True)


if __name__ == "__main__":
    one(1, 768, 128, 128)
    one(1, 768, 256, 128)
    one(4, 768, 128, 128)
    one(1, 384, 128, 128)
    decode(1)
    decode(8)
