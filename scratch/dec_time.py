import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import _lib
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
vae = vae.to(dev)
z = torch.randn(1, 4, 96, 96, device=dev)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("decode (fp32 NCHW out): %.3f ms" % timed(lambda: vae.decode(z)))
print("decode_image_u8       : %.3f ms" % timed(lambda: vae.decode_image_u8(z)))
