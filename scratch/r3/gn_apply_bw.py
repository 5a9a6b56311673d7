"""gn_apply_kernel bandwidth on the VAE's large tensors: rows-per-block sweep (option gn_apply_rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stable_diffusion_amd import engine as E, _lib as L
lib = L.load()
DEV = "cuda"
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for (n, hw, c) in [(1, 768 * 768, 128), (4, 768 * 768, 128), (1, 768 * 768, 256), (1, 384 * 384, 256), (1, 192 * 192, 512), (1, 9216, 320), (1, 9216, 640), (1, 2304, 640)]:
    x = torch.randn(n, hw, c, device=DEV).to(E.act_dtype())
    y = torch.empty_like(x)
    ss = torch.randn(n, c, 2, device=DEV)
    out = []
    for rows in (0, 1, 2, 4, 8, 16, 32, 64):
        assert lib.sdmi_set_option(b"gn_apply_rows", rows) == 0
        t = timed(lambda: L.call("sdmi_groupnorm_apply", x.data_ptr(), c, ss.data_ptr(), y.data_ptr(), c, n, hw, c, L.F_SILU, E._stream()))
        out.append("%d:%.1fus(%.2fTB/s)" % (rows, t, 2 * x.numel() * 2 / t / 1e6))
    lib.sdmi_set_option(b"gn_apply_rows", 0)
    print("n=%d hw=%d c=%d  " % (n, hw, c) + "  ".join(out), flush=True)
