"""GroupNorm stats+apply in graph chains at the UNet's shapes."""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"
def timed(g, reps, iters=10):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / reps * 1e6
tot = 0
for (hw, c, cnt) in [(96, 320, 13), (96, 640, 2), (96, 960, 1), (48, 320, 1), (48, 640, 11), (48, 1280, 1), (48, 1920, 1), (48, 960, 1), (24, 640, 1), (24, 1280, 12), (24, 2560, 2), (24, 1920, 1), (12, 1280, 12), (12, 2560, 3)]:
    x = E.new_act(2, hw, hw, c, dev); x.buf.normal_()
    w = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    fn = lambda: E.groupnorm(x, w, b, 1e-5, True)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    t = timed(g, 20)
    tot += t * cnt
    print("GN %3dx%-3d C=%4d: %6.1f us  (x%d per step)" % (hw, hw, c, t, cnt), flush=True)
print("GN per UNet step: %.0f us" % tot)
