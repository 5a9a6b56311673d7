import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
for rows, c in [(18432, 320), (4608, 640), (1152, 1280), (288, 1280)]:
    x = E.Act(torch.randn(1, rows, c, device=dev).to(H), 1, 1, rows, c)
    g = torch.randn(c, device=dev); b = torch.randn(c, device=dev)
    fn = lambda: E.layernorm(x, g, b, 1e-5)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): fn()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 200 * 1e6
    print("layernorm %5d x %4d: %6.2f us  %5.0f GB/s" % (rows, c, t, rows * c * 4 / t / 1e3), flush=True)
