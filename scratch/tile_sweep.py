"""Same problems under each forced tile variant (SDMI_IGEMM_TILE); run under rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
def gemm(m, n, k):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev); r = E.new_act(1, 1, m, n, dev); b = torch.zeros(n, device=dev)
    for _ in range(3):
        big.zero_()
        E.gemm(a, w, n, bias=b, residual=r, out=out)
def conv(n, hw, cin, cout):
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev)
    for _ in range(3):
        big.zero_()
        E.conv3x3(x, w, b, out=out)
SHAPES = [("g", 18432, 320, 320), ("g", 18432, 2560, 320), ("g", 18432, 960, 320), ("g", 18432, 320, 1280),
          ("g", 4608, 640, 640), ("g", 4608, 5120, 640), ("g", 4608, 1920, 640), ("g", 1152, 10240, 1280), ("g", 1152, 3840, 1280),
          ("c", 2, 96, 320, 320), ("c", 2, 96, 640, 320), ("c", 2, 48, 640, 640), ("c", 2, 48, 1280, 640), ("c", 2, 24, 1280, 1280),
          ("c", 1, 384, 256, 256), ("c", 1, 192, 512, 512), ("c", 1, 768, 128, 128)]
for s in SHAPES:
    if s[0] == "g": gemm(*s[1:])
    else: conv(*s[1:])
torch.cuda.synchronize()
