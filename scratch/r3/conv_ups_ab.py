"""Interleaved same-process A/B of the nearest-2x upsample convs: shared-window variant (conv_win2_kernel<..., UPS>) against
the generic gather kernel (option conv_win_ups_off), with a bit check.  argv: optional list of hw_in:c:n"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [
    (48, 640, 1), (24, 1280, 1), (12, 1280, 1), (48, 640, 2), (96, 512, 1), (192, 512, 1), (384, 256, 1), (384, 256, 4)]
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hw, c, n in shapes:
    x = E.Act(torch.randn(n, hw * hw, c, device=dev).to(H), n, hw, hw, c)
    W = torch.nn.Parameter(torch.randn(c, c, 3, 3, device=dev) / (9 * c) ** 0.5)
    b = torch.randn(c, device=dev)
    out = E.new_act(n, 2 * hw, 2 * hw, c, dev)
    fn = lambda: E.conv3x3(x, W, b, upsample=True, out=out)
    reps = 30 if hw <= 48 else 6
    best = {1: 1e9, 0: 1e9}; outs = {}; kn = {}
    for rnd in range(4):
        for v in (1, 0):
            lib.sdmi_set_option(b"conv_win_ups_off", v)
            best[v] = min(best[v], timeit(fn, reps))
            if rnd == 0:
                torch.cuda.synchronize(); outs[v] = out.buf.clone(); kn[v] = lib.sdmi_last_kernel().decode()
    lib.sdmi_set_option(b"conv_win_ups_off", 0)
    fl = 2.0 * n * 4 * hw * hw * c * 9 * c
    print("%-16s generic %8.1f us  window %8.1f us  ratio %.3f  %6.0f TFLOP/s  %s  %s -> %s" % ("%d:%d:%d" % (hw, c, n), best[1], best[0], best[0] / best[1],
          fl / best[0] * 1e-6, "bits equal" if torch.equal(outs[1], outs[0]) else "BITS DIFFER", kn[1], kn[0]), flush=True)
