"""Time sdmi_attention_d512 at the VAE mid-block shapes (graph of 5 launches, median of 10) + error against fp64 on a slice."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
def timeit(fn, inner=5, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(ts))
for b, n in [(1, 9216), (4, 9216), (1, 16384), (1, 4096)]:
    qkv = torch.randn(b, n, 1536, device=dev).to(H)
    A = E.Act(qkv, b, 1, n, 1536)
    o = E.new_act(b, 1, n, 512, dev)
    tus = timeit(lambda: E.attention_d512(A.cols(0, 512), A.cols(512, 512), A.cols(1024, 512), out=o, log2q=True))
    q64 = qkv[0, :64, :512].double(); k64 = qkv[0, :, 512:1024].double(); v64 = qkv[0, :, 1024:].double()
    ref = torch.softmax(q64 @ k64.t() * np.log(2.0), dim=-1) @ v64
    err = float((o.buf[0, :64].double() - ref).norm() / ref.norm())
    print("b%d n%d: %8.1f us  %6.0f TF algorithmic   rel-L2 (64 rows, fp64) %.2e" % (b, n, tus, 4.0 * b * n * n * 512 / tus / 1e6, err), flush=True)
