"""Per-block phase timestamps of one conv / gemm (SDMI_DEBUG_ABLATE=16)."""
import os, sys, ctypes, numpy as np, torch
os.environ["SDMI_DEBUG_ABLATE"] = "16"
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
lib = _lib.load()
lib.sdmi_debug_dump.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
def dump(nblocks, label):
    torch.cuda.synchronize()
    buf = np.zeros((nblocks, 8), dtype=np.uint64)
    rc = lib.sdmi_debug_dump(buf.ctypes.data, buf.nbytes)
    t = buf[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0   # s_memtime ticks at 100 MHz -> us
    d = np.diff(t, axis=1) / 100.0
    print(label, "blocks", nblocks, "rc", rc)
    print("  block start (us): min %.1f p50 %.1f p90 %.1f max %.1f" % (start.min(), np.median(start), np.percentile(start, 90), start.max()))
    for i, nm in enumerate(["prologue", "loop", "epilogue", "store-drain"]):
        print("  %-12s mean %.2f p50 %.2f p90 %.2f max %.2f us" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
    print("  end of last block %.1f us" % ((t[:, 4].max() - t0) / 100.0))
    xcc = buf[:, 6] & 0xf
    print("  blocks per xcc", np.bincount(xcc.astype(np.int64), minlength=8))
def conv(n, hw, cin, cout, nb):
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev)
    for _ in range(2): E.conv3x3(x, w, b, out=out)
    dump(nb, "conv %s" % ((n, hw, cin, cout),))
def gemm(m, n, k, nb):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev); r = E.new_act(1, 1, m, n, dev); b = torch.zeros(n, device=dev)
    for _ in range(2): E.gemm(a, w, n, bias=b, residual=r, out=out)
    dump(nb, "gemm %s" % ((m, n, k),))
conv(1, 192, 512, 512, 288 * 4)
conv(2, 96, 320, 320, 144 * 5)
gemm(18432, 2560, 320, 144 * 20)
gemm(4608, 640, 640, 72 * 5)
