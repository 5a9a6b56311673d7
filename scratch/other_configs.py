"""BASELINE.json configs 4 and 5 timed on one GPU (not bench lines: reported in DESIGN.md section 5)."""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import _lib
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet, vae = unet.to(dev), vae.to(dev)

def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

# config 4: 1024^2, b=4 -> UNet CFG batch 8 at 128x128 latents, 16384-token self-attention
x = torch.randn(8, 9, 128, 128, device=dev); t = torch.full((8,), 501, dtype=torch.long, device=dev)
ctx = unet.precompute_context(torch.randn(8, 77, 1024, device=dev)); te = unet.precompute_time_embedding(t)
unet(x, timesteps=te, context=ctx); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = unet(x, timesteps=te, context=ctx)
ms = timed(g.replay, 5)
gf = 4 * 9350.9
print("config4 UNet step 1024^2 b=4 (CFG batch 8): %.2f ms  -> %.0f TFLOP/s (%.1f%% of 2.5 PF)" % (ms, gf / ms, gf / ms / 25.0))
del g, y, x
# config 5: decode only 768^2 b=32
z = torch.randn(32, 4, 96, 96, device=dev)
ms = timed(lambda: vae.decode(z), 3)
gf = 32 * 5754.3
print("config5 decode 768^2 b=32: %.1f ms  (%.2f ms/image) -> %.0f TFLOP/s (%.1f%% of 2.5 PF)" % (ms, ms / 32, gf / ms, gf / ms / 25.0))
z1 = z[:1].contiguous()
ms1 = timed(lambda: vae.decode(z1), 5)
print("decode 768^2 b=1: %.2f ms" % ms1)
