"""How long is ONE CFG half's dependent chain on an otherwise idle chip?  Graph replay of a batch-1 UNet forward at 768x768,
of the two-stream step, and of one batch-2 pass."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stable_diffusion_amd import _lib, engine as E
from stable_diffusion_amd.sampler import Sampler
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev); unet = unet.to(dev)
L = 96
def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep
def timeit(g):
    ts = []
    for _ in range(6):
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    return float(np.median(ts))
for b in (1, 2):
    x = torch.randn(b, 9, L, L, device=dev)
    cc = unet.precompute_context(torch.randn(b, 77, 1024, device=dev))
    te = unet.precompute_time_embedding(torch.full((b,), 501, dtype=torch.long, device=dev))
    g, keep = graph_of(lambda: unet(x, timesteps=te, context=cc))
    print("one pass, batch %d, one stream: %.3f ms" % (b, timeit(g)), flush=True)
s = Sampler(unet, None)
x2 = torch.randn(1, 9, L, L, device=dev).repeat(2, 1, 1, 1)
te2 = unet.precompute_time_embedding(torch.full((2,), 501, dtype=torch.long, device=dev))
cc2 = s.make_context(torch.randn(1, 77, 1024, device=dev), torch.randn(1, 77, 1024, device=dev))
s._unet_step(x2, te2, cc2, True, same_halves=True); torch.cuda.synchronize()
print("two streams (shared prefix): %.3f ms" % timeit(s._graph), flush=True)
