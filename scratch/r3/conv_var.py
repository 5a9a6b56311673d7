"""Interleaved same-process timing of conv_win2_kernel experiment variants (conv_win_var option), with a bit check against
variant 0.  argv: comma-separated variant list, then optional hw:cin:cout:n shapes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
OPT = os.environ.get("OPT", "conv_win_var").encode()
variants = [int(v) for v in sys.argv[1].split(",")]
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [
    (96, 320, 320, 1), (96, 320, 320, 2), (96, 640, 320, 2), (48, 640, 640, 1), (48, 640, 640, 2), (48, 1280, 640, 2),
    (24, 1280, 1280, 2), (12, 1280, 1280, 2), (768, 128, 128, 1), (384, 256, 256, 1), (192, 512, 512, 1)]
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-22s " % "hw:cin:cout:n" + " ".join("%9s" % ("var%d us" % v) for v in variants) + "   TFLOP/s best   kernel")
for hw, cin, cout, n in shapes:
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
    fn = lambda: E.conv3x3(x, W, b, residual=res, out=out)
    reps = 30 if hw <= 96 else 8
    best = {v: 1e9 for v in variants}; ref = None; same = True
    for rnd in range(4):
        for v in variants:
            lib.sdmi_set_option(OPT, v)
            best[v] = min(best[v], timeit(fn, reps))
            if rnd == 0:
                torch.cuda.synchronize()
                if ref is None: ref = out.buf.clone()
                else: same = same and torch.equal(ref, out.buf)
    lib.sdmi_set_option(OPT, 0)
    fn(); kn = lib.sdmi_last_kernel().decode()
    fl = 2.0 * n * hw * hw * cout * 9 * cin
    print("%-22s " % ("%d:%d:%d:%d" % (hw, cin, cout, n)) + " ".join("%9.1f" % best[v] for v in variants) +
          "   %8.0f  %s  %s" % (fl / min(best.values()) * 1e-6, "bits equal" if same else "BITS DIFFER", kn[:40]), flush=True)
