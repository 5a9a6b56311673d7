import sys, os, time, torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "none"]
exec(open("scratch/tune_sweep.py").read().split("which = sys.argv")[0])
def gemm3(m, n, k):
    a = E.Act(torch.randn(2, m // 2, k, device=dev).to(H), 2, 1, m // 2, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(2, 1, m // 2, n, dev); res = E.new_act(2, 1, m // 2, n, dev)
    fn = lambda: E.gemm(a, w, n, bias=b, residual=res, out=out)
    r = []
    for tile in (-1, 4, 5, 3, 2):
        lib.sdmi_tune_override(tile, 1 if tile >= 0 else -1)
        r.append("%s %.1f" % ("auto" if tile < 0 else "t%d" % tile, measure(fn)))
    lib.sdmi_tune_override(-1, -1)
    print("gemm %5d x %5d x %5d: %s" % (m, n, k, "  ".join(r)), flush=True)
for a in [(4608, 640, 640), (4608, 640, 320), (4608, 640, 1280), (4608, 640, 2560), (1152, 1280, 1280), (1152, 1280, 640), (1152, 1280, 2560), (288, 1280, 1280), (18432, 320, 320), (18432, 320, 1280)]: gemm3(*a)
