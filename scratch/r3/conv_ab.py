"""Interleaved same-process A/B of the window conv kernels under an option (env OPT, default conv_win_v1; env VALS = "A,B",
default "1,0": A is the baseline) on the UNet's conv shapes, graph-free event timing of back-to-back launches, with a bit
check of the two outputs.  argv: optional list of hw:cin:cout:n"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
OPT = os.environ.get("OPT", "conv_win_v1").encode()
VA, VB = (int(v) for v in os.environ.get("VALS", "1,0").split(","))
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [
    (96, 320, 320, 1), (96, 320, 320, 2), (96, 640, 320, 1), (96, 960, 320, 1), (96, 640, 320, 2),
    (48, 640, 640, 1), (48, 640, 640, 2), (48, 1280, 640, 1), (48, 1920, 640, 1), (48, 320, 640, 1),
    (24, 1280, 1280, 1), (24, 2560, 1280, 1), (12, 1280, 1280, 1), (12, 2560, 1280, 1),
    (768, 128, 128, 1), (384, 256, 256, 1), (192, 512, 512, 1), (96, 512, 512, 1)]
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-28s %10s %10s %8s  %s" % ("hw:cin:cout:n", "%s=%d us" % (OPT.decode(), VA), "=%d us" % VB, "B/A", "TFLOP/s B"))
for hw, cin, cout, n in shapes:
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
    fn = lambda: E.conv3x3(x, W, b, residual=res, out=out)
    reps = 30 if hw <= 96 else 8
    best = {VA: 1e9, VB: 1e9}; outs = {}; kn = {}
    for rnd in range(4):
        for v in (VA, VB):
            lib.sdmi_set_option(OPT, v)
            best[v] = min(best[v], timeit(fn, reps))
            if rnd == 0:
                torch.cuda.synchronize(); outs[v] = out.buf.clone(); kn[v] = lib.sdmi_last_kernel().decode()
    lib.sdmi_set_option(OPT, 0)
    fl = 2.0 * n * hw * hw * cout * 9 * cin
    print("%-28s %10.1f %10.1f %8.3f  %8.0f  %s  %s -> %s" % ("%d:%d:%d:%d" % (hw, cin, cout, n), best[VA], best[VB], best[VB] / best[VA], fl / best[VB] * 1e-6,
          "bits equal" if torch.equal(outs[VA], outs[VB]) else "BITS DIFFER", kn[VA], kn[VB]), flush=True)
