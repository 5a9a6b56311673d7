"""UNet step latency (graph replay, 768^2 CFG batch 2) for the library named by SDMI_LIB_PATH; 3 rounds of 20 replays."""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import _lib
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet = unet.to(dev)
x = torch.randn(2, 9, 96, 96, device=dev); t = torch.full((2,), 501, dtype=torch.long, device=dev)
ctx = unet.precompute_context(torch.randn(2, 77, 1024, device=dev)); te = unet.precompute_time_embedding(t)
for _ in range(2): unet(x, timesteps=te, context=ctx)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = unet(x, timesteps=te, context=ctx)
res = []
for r in range(4):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 20 * 1e3)
print("unet step ms:", " ".join("%.3f" % v for v in res), flush=True)
