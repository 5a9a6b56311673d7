"""Split-K conv / GEMM chains with the separate reduce kernel against the in-launch (ticketed) reduction."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load(); REPS = 20
def measure(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / REPS * 1e6
for (hw, cin, cout) in [(24, 1280, 1280), (24, 2560, 1280), (12, 1280, 1280), (48, 640, 640), (48, 1280, 640)]:
    x = E.Act(torch.randn(1, hw * hw, cin, device=dev).to(H), 1, hw, hw, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev); out = E.new_act(1, hw, hw, cout, dev); res = E.new_act(1, hw, hw, cout, dev)
    fn = lambda: E.conv3x3(x, W, b, residual=res, out=out)
    r = []
    for fl in (0, 1024, 0, 1024):
        E.WS_FLAGS = fl
        r.append(measure(fn)); k = lib.sdmi_last_kernel().decode()
    print("conv %dx%d %d->%d: reduce kernel %.1f / %.1f us, in launch %.1f / %.1f us   [%s]" % (hw, hw, cin, cout, r[0], r[2], r[1], r[3], k[:60]), flush=True)
