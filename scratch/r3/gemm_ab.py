"""Interleaved same-process A/B of the GEMM loaders (gemm_buf_off option) on the UNet's GEMM shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
OPT = sys.argv[1].encode() if len(sys.argv) > 1 else b"gemm_buf_off"
shapes = [(9216, 320, 320, "res"), (9216, 960, 320, ""), (9216, 320, 1280, "res"), (18432, 960, 320, ""), (18432, 320, 320, "res"),
          (9216, 2560, 320, "geglu"), (2304, 640, 640, "res"), (2304, 1920, 640, ""), (2304, 5120, 640, "geglu"), (2304, 640, 2560, "res"),
          (576, 1280, 1280, "res"), (576, 3840, 1280, ""), (576, 10240, 1280, "geglu"), (576, 1280, 5120, "res"), (144, 1280, 1280, "res")]
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("option", OPT.decode())
print("%-28s %10s %10s %8s  %s" % ("m:n:k:kind", "opt=1 us", "opt=0 us", "0/1", "TFLOP/s opt=0"))
for m, n, k, kind in shapes:
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H).contiguous()
    b = torch.randn(n, device=dev)
    nout = n // 2 if kind == "geglu" else n
    res = E.Act(torch.randn(1, m, nout, device=dev).to(H), 1, 1, m, nout) if kind == "res" else None
    out = E.new_act(1, 1, m, nout, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, residual=res, geglu=(kind == "geglu"), out=out)
    best = {0: 1e9, 1: 1e9}
    for rnd in range(4):
        for v in (1, 0):
            lib.sdmi_set_option(OPT, v)
            best[v] = min(best[v], timeit(fn, 40))
    lib.sdmi_set_option(OPT, 0)
    fn(); kn = lib.sdmi_last_kernel().decode()
    print("%-28s %10.1f %10.1f %8.3f  %8.0f  %s" % ("%d:%d:%d:%s" % (m, n, k, kind), best[1], best[0], best[0] / best[1], 2.0 * m * n * k / best[0] * 1e-6, kn))
