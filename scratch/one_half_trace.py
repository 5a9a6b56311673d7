"""Workload for rocprofv3 --kernel-trace: graph replay of ONE batch-1 UNet pass at 768x768 (after a spin kernel marker)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stable_diffusion_amd import _lib, engine as E
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev); unet = unet.to(dev)
L = 96
x = torch.randn(1, 9, L, L, device=dev)
cc = unet.precompute_context(torch.randn(1, 77, 1024, device=dev))
te = unet.precompute_time_embedding(torch.full((1,), 501, dtype=torch.long, device=dev))
unet(x, timesteps=te, context=cc); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = unet(x, timesteps=te, context=cc)
for _ in range(3): g.replay()
torch.cuda.synchronize()
torch.cuda._sleep(int(2e6)); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
