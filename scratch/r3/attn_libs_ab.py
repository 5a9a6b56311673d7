"""attention timing of two library builds in ONE process is not possible (one dlopen per name space): run this once per
library (SDMI_LIB_PATH) back to back on the same box and compare."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype(); lib = _lib.load()
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for b, h, nq, nkv in [(1, 5, 9216, 9216), (2, 5, 9216, 9216), (2, 10, 2304, 2304), (4, 5, 16384, 16384)]:
    q = E.Act((torch.randn(b, nq, h * 64, device=dev) * E.Q_LOG2_SCALE).to(H), b, 1, nq, h * 64)
    k = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
    v = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
    o = E.new_act(b, 1, nq, h * 64, dev)
    fn = lambda: E.attention(q, k, v, h, out=o, log2q=True)
    best = min(timeit(fn, 10) for _ in range(5))
    print("%-20s %8.1f us %6.0f TFLOP/s  %s" % ("%d:%d:%d:%d" % (b, h, nq, nkv), best, 4.0 * b * h * nq * nkv * 64 / best * 1e-6, lib.sdmi_last_kernel().decode()), flush=True)
