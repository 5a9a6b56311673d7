"""Persistent-block GEMM (igemm_persist_kernel) against the one-tile-per-block kernel: same bits, time per launch.
Shapes: the plain / GEGLU / LayerNorm-folded GEMMs of the 96x96 and 48x48 transformer blocks (one CFG half and batch 2)."""
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
torch.manual_seed(0)
cases = [("plain", m, n, k) for m, n, k in [(9216, 320, 320), (18432, 320, 320), (9216, 960, 320), (18432, 960, 320), (9216, 320, 1280),
                                            (18432, 320, 1280), (2304, 640, 640), (4608, 640, 640), (4608, 1920, 640), (4608, 640, 2560), (1000, 200, 328)]]
cases += [("geglu", m, n, k) for m, n, k in [(9216, 2560, 320), (18432, 2560, 320), (2304, 5120, 640), (4608, 5120, 640), (576, 10240, 1280)]]
cases += [("ln", m, n, k) for m, n, k in [(9216, 960, 320), (18432, 960, 320), (4608, 1920, 640)]]
cases += [("lngeglu", m, n, k) for m, n, k in [(9216, 2560, 320), (4608, 5120, 640)]]
for kind, m, n, k in cases:
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.randn(n, device=dev)
    geglu = "geglu" in kind
    nc = n // 2 if geglu else n
    out = E.new_act(1, 1, m, nc, dev)
    kw = {}
    if kind == "plain":
        kw = dict(residual=E.Act(torch.randn(1, m, nc, device=dev).to(H), 1, 1, m, nc), rowstats=True)
    if kind.startswith("ln"):
        # producer GEMM leaves the row statistics the folded LayerNorm needs
        src = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
        w0 = (torch.randn(k, k, device=dev) / k ** 0.5).to(H)
        lib.sdmi_set_option(b"gemm_persist", 0)
        a = E.gemm(src, w0, k, bias=torch.zeros(k, device=dev), rowstats=True)
        if a.rstats is None:
            print("%s %d x %d x %d: no rowstats from the producer, skipped" % (kind, m, n, k)); continue
        kw = dict(ln=(torch.randn(n, device=dev), 1e-5))
    fn = lambda: E.gemm(a, w, n, bias=b, geglu=geglu, out=out, **kw)
    r = []; outs = []; names = []
    for opt in (0, 2, 1):
        lib.sdmi_set_option(b"gemm_persist", opt)
        out.buf.zero_()
        res = fn(); torch.cuda.synchronize()
        outs.append(out.buf.clone()); names.append(lib.sdmi_last_kernel().decode()[:48])
        if res.rstats is not None: outs[-1] = (outs[-1], res.rstats[0].clone())
        r.append("opt%d %6.1f" % (opt, measure(fn)))
    def same(x, y):
        return all(torch.equal(p, q) for p, q in zip(x, y)) if isinstance(x, tuple) else torch.equal(x, y)
    print("%-7s %5d x %5d x %4d  %s  us  bits %s %s | %s | auto: %s" % (kind, m, n, k, "  ".join(r), same(outs[0], outs[1]), same(outs[0], outs[2]), names[1], names[2]), flush=True)
lib.sdmi_set_option(b"gemm_persist", 0)
