import sys, os, time, torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "none"]
exec(open("scratch/tune_sweep.py").read().split("which = sys.argv")[0])
exec(open("scratch/nmajor_ab.py").read().split("for a in [(24, 640")[0].split("def one(")[1].join(["def one(", ""]) if False else "")
def one(label, fn):
    lib.sdmi_tune_override(-1, -1)
    print("%-34s %6.1f us" % (label, measure(fn)), flush=True)
def conv2(hw, cin, cout, n=2):
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
    one("conv %dx%d %d->%d" % (hw, hw, cin, cout), lambda: E.conv3x3(x, W, b, residual=res, out=out))
def gemm2(m, n, k, geglu=False):
    a = E.Act(torch.randn(2, m // 2, k, device=dev).to(H), 2, 1, m // 2, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    out = E.new_act(2, 1, m // 2, n // 2 if geglu else n, dev); res = None if geglu else E.new_act(2, 1, m // 2, n, dev)
    one("gemm %d x %d x %d%s" % (m, n, k, " geglu" if geglu else ""), lambda: E.gemm(a, w, n, bias=b, residual=res, geglu=geglu, out=out))
for a in [(96, 320, 320), (96, 640, 320), (96, 960, 320), (48, 640, 640), (48, 1920, 640), (48, 320, 640)]: conv2(*a)
conv2(192, 512, 512, 1); conv2(384, 256, 256, 1)
for a in [(18432, 320, 320), (18432, 960, 320), (18432, 2560, 320, True), (18432, 320, 1280), (4608, 640, 640), (4608, 1920, 640), (4608, 5120, 640, True), (4608, 640, 2560), (1152, 10240, 1280, True), (288, 10240, 1280, True)]: gemm2(*a)
