import sys, os, time, torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "none"]
exec(open("scratch/tune_sweep.py").read().split("which = sys.argv")[0])
def one(label, fn):
    lib.sdmi_tune_override(-1, -1)
    print("%-34s %6.1f us" % (label, measure(fn)), flush=True)
def conv2(hw, cin, cout, stride=1, ups=False, n=2):
    hin = hw * stride if not ups else hw // 2
    x = E.Act(torch.randn(n, hin * hin, cin, device=dev).to(H), n, hin, hin, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
    one("conv %dx%d %d->%d" % (hw, hw, cin, cout), lambda: E.conv3x3(x, W, b, residual=res, out=out))
def gemm2(m, n, k):
    a = E.Act(torch.randn(2, m // 2, k, device=dev).to(H), 2, 1, m // 2, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(2, 1, m // 2, n, dev); res = E.new_act(2, 1, m // 2, n, dev)
    one("gemm %d x %d x %d" % (m, n, k), lambda: E.gemm(a, w, n, bias=b, residual=res, out=out))
for a in [(24, 640, 1280), (24, 1280, 1280), (24, 2560, 1280), (12, 1280, 1280), (12, 2560, 1280), (48, 1280, 640)]: conv2(*a)
for a in [(1152, 1280, 1280), (1152, 3840, 1280), (1152, 1280, 5120), (288, 1280, 1280), (288, 3840, 1280), (288, 1280, 5120)]: gemm2(*a)
