"""Fixed vs per-K-tile cost of the GEMM kernel: a few (M, N) with K swept; run under rocprofv3 --kernel-trace."""
import sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
def gemm(m, n, k, res, flush):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev); r = E.new_act(1, 1, m, n, dev); b = torch.zeros(n, device=dev)
    for _ in range(3):
        if flush: big.zero_()
        E.gemm(a, w, n, bias=b, residual=r if res else None, out=out)
for flush in (1, 0):
    for m, n in [(4608, 640), (18432, 320), (1152, 1280)]:
        for k in (64, 320, 640, 1280, 2560):
            for res in (0, 1):
                gemm(m, n, k, res, flush)
torch.cuda.synchronize()
