"""Tile-variant sweep of the UNet's plain / GEGLU GEMM shapes at 768x768 as ONE CFG half runs them (batch 1); graph chains of 20."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
lib = _lib.load()
REPS = 20
def timed(g, iters=5):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / REPS * 1e6
def measure(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS): fn()
    return timed(g)
def gemm(m, n, k, geglu=False, nb=1):
    a = E.Act(torch.randn(nb, m // nb, k, device=dev).to(H), nb, 1, m // nb, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(nb, 1, m // nb, n // 2 if geglu else n, dev)
    res = None if geglu else E.new_act(nb, 1, m // nb, n, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, residual=res, geglu=geglu, out=out)
    lib.sdmi_tune_override(-1, -1)
    auto = measure(fn); kern = lib.sdmi_last_kernel().decode()
    rows = {}
    tiles = (1, 2, 3, 4, 5, 6, 7) if not geglu else (1, 3, 7)
    for tile in tiles:
        for s in ((1,) if k < 2048 or geglu else (1, 2)):
            lib.sdmi_tune_override(tile, s)
            try:
                rows[(tile, s)] = measure(fn)
            except Exception as e:
                rows[(tile, s)] = float("nan")
    lib.sdmi_tune_override(-1, -1)
    top = sorted(rows.items(), key=lambda kv: kv[1])
    fl = 2.0 * m * n * k
    print("gemm %5d x %5d x %4d%s nb=%d auto %6.1f us (%4.0f TF) [%s] | %s" % (m, n, k, " geglu" if geglu else "      ", nb, auto, fl / auto / 1e6, kern[13:30],
          "  ".join("t%d/s%d %.1f" % (kk[0], kk[1], v) for kk, v in top)), flush=True)
shapes = [(9216, 320, 320), (9216, 960, 320), (9216, 2560, 320, True), (9216, 320, 1280), (9216, 320, 640), (9216, 320, 960),
          (2304, 640, 640), (2304, 1920, 640), (2304, 5120, 640, True), (2304, 640, 2560), (2304, 640, 1280), (2304, 640, 1920), (2304, 640, 960), (2304, 640, 320),
          (576, 1280, 1280), (576, 3840, 1280), (576, 10240, 1280, True), (576, 1280, 5120), (576, 1280, 2560), (576, 1280, 1920), (576, 1280, 640),
          (144, 1280, 1280), (144, 3840, 1280), (144, 10240, 1280, True), (144, 1280, 5120), (144, 1280, 2560)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for sh in shapes:
    gemm(sh[0] * nb, sh[1], sh[2], len(sh) > 3, nb)
