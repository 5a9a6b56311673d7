"""A handful of GEMM / conv shapes, 3 launches each, for rocprofv3 --kernel-trace (true GPU durations)."""
import sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)   # flush caches between shapes
def gemm(m, n, k, res=True, bias=True):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev); r = E.new_act(1, 1, m, n, dev); b = torch.zeros(n, device=dev)
    for _ in range(3):
        big.zero_()
        E.gemm(a, w, n, bias=b if bias else None, residual=r if res else None, out=out)
for m, n, k in [(18432, 320, 320), (18432, 960, 320), (18432, 320, 1280), (4608, 640, 640), (4608, 1920, 640), (1152, 1280, 1280), (1152, 3840, 1280), (154, 640, 1024), (154, 2560, 1024), (288, 1280, 1280), (288, 3840, 1280)]:
    gemm(m, n, k)
torch.cuda.synchronize()
