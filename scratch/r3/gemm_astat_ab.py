"""A-stationary GEMM kernel against igemm_kernel on the K = 320 / 384 GEMM shapes, same process, interleaved, graph chains of 20."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
REPS = 20
def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS): fn()
    return g
def timed(g, iters=5):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / REPS * 1e6
shapes = [(9216, 960, 320, False), (9216, 2560, 320, True), (9216, 320, 320, False), (18432, 960, 320, False), (18432, 2560, 320, True),
          (16384, 960, 320, False), (16384, 2560, 320, True), (65536, 960, 320, False), (65536, 2560, 320, True)]
print("%-28s %12s %12s %8s %10s" % ("m x n x k", "igemm us", "astat us", "B/A", "TFLOP/s B"))
for m, n, k, geglu in shapes:
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(1, 1, m, n // 2 if geglu else n, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, geglu=geglu, out=out)
    gs, names = {}, {}
    for v in (-1, 1):
        lib.sdmi_set_option(b"gemm_astat", max(v, 0))
        gs[v] = graph_of(fn); names[v] = lib.sdmi_last_kernel().decode()
    lib.sdmi_set_option(b"gemm_astat", 0)
    best = {-1: 1e9, 1: 1e9}
    for rnd in range(4):
        for v in (-1, 1): best[v] = min(best[v], timed(gs[v]))
    print("%-28s %12.1f %12.1f %8.3f %10.0f  %s -> %s" % ("%d x %d x %d%s" % (m, n, k, " geglu" if geglu else ""), best[-1], best[1], best[1] / best[-1],
          2.0 * m * n * k / best[1] * 1e-6, names[-1][:44], names[1]), flush=True)
