"""Interleaved same-process A/B of the long self-attention kernel under an option (default attn_pipe_off: 1 = round-2 lockstep
block, 0 = software-pipelined block), with a bit check.  argv: optional b:h:nq:nkv shapes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
OPT = os.environ.get("OPT", "attn_pipe_off").encode()
VA, VB = (int(v) for v in os.environ.get("VALS", "1,0").split(","))   # A = the baseline value, B = the candidate
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1, 5, 9216, 9216), (2, 5, 9216, 9216), (1, 10, 2304, 2304), (2, 10, 2304, 2304), (4, 5, 16384, 16384)]
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-24s %10s %10s %8s %12s" % ("b:h:nq:nkv", "%s=%d us" % (OPT.decode(), VA), "=%d us" % VB, "B/A", "TFLOP/s B"))
for b, h, nq, nkv in shapes:
    q = E.Act((torch.randn(b, nq, h * 64, device=dev) * E.Q_LOG2_SCALE).to(H), b, 1, nq, h * 64)
    k = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
    v = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
    o = E.new_act(b, 1, nq, h * 64, dev)
    fn = lambda: E.attention(q, k, v, h, out=o, log2q=True)
    best = {VA: 1e9, VB: 1e9}; outs = {}; names = {}
    for rnd in range(4):
        for val in (VA, VB):
            lib.sdmi_set_option(OPT, val)
            best[val] = min(best[val], timeit(fn, 10))
            if rnd == 0:
                torch.cuda.synchronize(); outs[val] = o.buf.clone(); names[val] = lib.sdmi_last_kernel().decode()
    lib.sdmi_set_option(OPT, 0)
    same = torch.equal(outs[VA], outs[VB])
    rel = float((outs[VA].double() - outs[VB].double()).norm() / outs[VA].double().norm())
    print("%-24s %10.1f %10.1f %8.3f %12.0f  %s  %s -> %s" % ("%d:%d:%d:%d" % (b, h, nq, nkv), best[VA], best[VB], best[VB] / best[VA],
          4.0 * b * h * nq * nkv * 64 / best[VB] * 1e-6, "bits equal" if same else "rel-L2 %.2e" % rel, names[VA], names[VB]), flush=True)
