"""GEGLU GEMMs of one CFG half (batch 1) at the UNet's three transformer levels: library's choice against forced tile variants."""
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
for m, n, k in [(9216, 2560, 320), (2304, 5120, 640), (576, 10240, 1280)]:
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(1, 1, m, n // 2, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, geglu=True, out=out)
    res = []
    for tile in (-1, 0, 1, 2, 3):
        lib.sdmi_tune_override(tile, -1)
        try:
            t = measure(fn); res.append("tile %2d %6.1f us (%4.0f TF) %s" % (tile, t, 2.0 * m * n * k / t / 1e6, lib.sdmi_last_kernel().decode()[:44]))
        except Exception as e:
            res.append("tile %d failed: %s" % (tile, str(e)[:60]))
    lib.sdmi_tune_override(-1, -1)
    print("geglu %d x %d x %d\n  " % (m, n, k) + "\n  ".join(res), flush=True)
