"""Sustained run: N images with the graph sampler, every image must be bit-identical (same inputs), time per image."""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import _lib
from stable_diffusion_amd.sampler import Sampler
from stable_diffusion_amd.synth import make_inputs
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet, vae = unet.to(dev), vae.to(dev)
inp = make_inputs(1, 96, 50, seed=100)
g = lambda t: t.to(dev)
args = (50, g(inp["cond"]), g(inp["uncond"]), g(inp["c_concat"]), g(inp["x_T"]), [g(n) for n in inp["noises"]], 9.0)
s = Sampler(unet, vae)
x0, img0 = s.sample(*args, use_graph=True)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts = []
for i in range(N):
    t0 = time.perf_counter()
    x, img = s.sample(*args, use_graph=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    assert torch.equal(x, x0) and torch.equal(img, img0), "image %d differs" % i
print("%d images bit-identical; ms per image: first %.1f, min %.1f, max %.1f, last %.1f" % (N, ts[0], min(ts), max(ts), ts[-1]), flush=True)
