"""(tile variant, split-K factor) sweep on the UNet's conv / GEMM shapes at 768x768, CFG batch 2; graph chains of 20 launches."""
import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
AUTO_ONLY = os.environ.get("SWEEP_AUTO_ONLY") == "1"     # only the library's own choice, with a checksum of the output: A/B of two builds
def sweep(label, fn, nk, flops):
    lib.sdmi_tune_override(-1, -1)
    auto = measure(fn); kern = lib.sdmi_last_kernel().decode()
    if AUTO_ONLY:
        o = fn()
        h = float(o.buf.float().abs().sum().item()) if hasattr(o, "buf") else 0.0
        print("%-34s auto %6.1f us (%4.0f TF) [%s] sum|out| %.6e" % (label, auto, flops / auto / 1e6, kern[:40], h), flush=True)
        return
    best = (auto, "auto")
    rows = {}
    for tile in (1, 2, 3, 4):
        for s in (1, 2, 3, 4, 6, 8, 12):
            if s > 1 and nk // s < 8: continue
            lib.sdmi_tune_override(tile, s)
            try:
                t = measure(fn)
            except Exception as e:
                t = float("nan")
            rows[(tile, s)] = t
            if t < best[0]: best = (t, "tile %d s %d" % (tile, s))
    lib.sdmi_tune_override(-1, -1)
    top = sorted(rows.items(), key=lambda kv: kv[1])[:4]
    print("%-34s auto %6.1f us (%4.0f TF) [%s]  best %6.1f us %-12s | %s" % (label, auto, flops / auto / 1e6, kern[13:34], best[0], best[1],
          "  ".join("t%d/s%d %.1f" % (k[0], k[1], v) for k, v in top)), flush=True)
def conv(hw, cin, cout, stride=1, ups=False, n=2):
    hin = hw * stride if not ups else hw // 2
    x = E.Act(torch.randn(n, hin * hin, cin, device=dev).to(H), n, hin, hin, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
    kw = dict(stride=stride, upsample=ups)
    if stride == 2: kw["pad"] = (1, 1, 1, 1)
    fn = lambda: E.conv3x3(x, W, b, residual=res, out=out, **kw)
    sweep("conv %dx%d %d->%d s%d%s" % (hw, hw, cin, cout, stride, " ups" if ups else ""), fn, 9 * ((cin + 63) // 64), 2.0 * n * hw * hw * cout * 9 * cin)
def gemm(m, n, k, geglu=False):
    a = E.Act(torch.randn(2, m // 2, k, device=dev).to(H), 2, 1, m // 2, k)
    if geglu:
        wf = torch.randn(n, k, device=dev) / k ** 0.5
        w = E.packed_linear([wf], True) if False else (wf.to(H))
    else:
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(2, 1, m // 2, n // 2 if geglu else n, dev)
    res = None if geglu else E.new_act(2, 1, m // 2, n, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, residual=res, geglu=geglu, out=out)
    sweep("gemm %d x %d x %d%s" % (m, n, k, " geglu" if geglu else ""), fn, (k + 63) // 64, 2.0 * m * n * k)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which in ("all", "conv"):
    for a in [(96, 320, 320), (96, 640, 320), (96, 960, 320), (96, 640, 640, 1, True), (48, 320, 320, 2), (48, 320, 640), (48, 640, 640), (48, 1280, 640), (48, 1920, 640), (48, 960, 640),
              (48, 1280, 1280, 1, True), (24, 640, 640, 2), (24, 640, 1280), (24, 1280, 1280), (24, 2560, 1280), (24, 1920, 1280), (24, 1280, 1280, 1, True),
              (12, 1280, 1280, 2), (12, 1280, 1280), (12, 2560, 1280)]:
        conv(*a, **({} if len(a) > 5 else {'n': NB}))
if which in ("all", "gemm"):
    for a in [(18432, 320, 320), (18432, 960, 320), (18432, 2560, 320, True), (18432, 320, 1280), (18432, 320, 640), (18432, 320, 960),
              (4608, 640, 640), (4608, 1920, 640), (4608, 5120, 640, True), (4608, 640, 2560), (4608, 640, 320), (4608, 640, 1280), (4608, 640, 1920), (4608, 640, 960),
              (1152, 1280, 1280), (1152, 3840, 1280), (1152, 10240, 1280, True), (1152, 1280, 5120), (1152, 1280, 640), (1152, 1280, 2560), (1152, 1280, 1920),
              (288, 1280, 1280), (288, 3840, 1280), (288, 10240, 1280, True), (288, 1280, 5120), (288, 1280, 2560)]:
        gemm(*a)
