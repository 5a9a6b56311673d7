"""How much do cold (HBM) weights cost the latency-bound kernels?  Same launch chain with one weight set (hot) or many (cold)."""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
def gemm_case(m, n, k, nw, reps=200):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    ws = [(torch.randn(n, k, device=dev) / k ** 0.5).to(H) for _ in range(nw)]
    out = E.new_act(1, 1, m, n, dev)
    res = {}
    for label, pick in (("hot", lambda i: ws[0]), ("cold", lambda i: ws[i % nw])):
        for i in range(2): E.gemm(a, pick(i), n, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps): E.gemm(a, pick(i), n, out=out)
        res[label] = timed(g.replay) / reps * 1e3
    print("gemm", (m, n, k), "weights %.1f MB x %d: hot %.1f us, cold %.1f us" % (n * k * 2 / 1e6, nw, res["hot"], res["cold"]), flush=True)
def conv_case(n, hw, cin, cout, nw, reps=40):
    x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    Ws = [torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5) for _ in range(nw)]
    b = torch.zeros(cout, device=dev)
    out = E.new_act(n, hw, hw, cout, dev)
    res = {}
    for W in Ws: E.conv3x3(x, W, b, out=out)      # pre-pack
    torch.cuda.synchronize()
    for label, pick in (("hot", lambda i: Ws[0]), ("cold", lambda i: Ws[i % nw])):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps): E.conv3x3(x, pick(i), b, out=out)
        res[label] = timed(g.replay) / reps * 1e3
    print("conv", (n, hw, cin, cout), "weights %.1f MB x %d: hot %.1f us, cold %.1f us" % (cout * cin * 18 / 1e6, nw, res["hot"], res["cold"]), flush=True)
gemm_case(4608, 640, 640, 600)
gemm_case(1152, 1280, 1280, 200)
gemm_case(18432, 320, 320, 600)
gemm_case(1152, 10240, 1280, 20)
conv_case(2, 12, 1280, 1280, 12)
conv_case(2, 24, 1280, 1280, 12)
conv_case(2, 48, 640, 640, 40)
