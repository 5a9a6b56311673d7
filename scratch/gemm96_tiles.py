"""Plain GEMMs of the 96x96 / 48x48 transformer blocks (one CFG half and the batch-2 pass): library's choice against forced variants."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
REPS = 20
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
for m, n, k in [(9216, 320, 320), (18432, 320, 320), (9216, 960, 320), (18432, 960, 320), (9216, 320, 1280), (18432, 320, 1280), (2304, 640, 640), (4608, 640, 640), (4608, 1920, 640), (4608, 640, 2560)]:
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(1, 1, m, n, dev); res = E.new_act(1, 1, m, n, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, residual=res, out=out)
    r = []
    for tile in (-1, 2, 0, 1, 3, 7):
        lib.sdmi_tune_override(tile, -1)
        t = measure(fn); r.append("t%d %5.1f" % (tile, t))
    lib.sdmi_tune_override(-1, -1)
    print("gemm %5d x %4d x %4d   %s   (us; best case HBM %.1f, MFMA %.1f)" % (m, n, k, "  ".join(r), 2.0 * (m * k + n * k + 2 * m * n) / 8e6, 2.0 * m * n * k / 2.5e9), flush=True)
