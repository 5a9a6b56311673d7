"""Do two independent kernel chains captured on two streams run concurrently under HIP graph replay?"""
import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
def mk(m, n, k):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev)
    return a, w, out, n
def chain(args, reps):
    a, w, out, n = args
    for _ in range(reps): E.gemm(a, w, n, out=out)
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for shape in [(1152, 1280, 1280), (576, 1280, 1280), (4608, 640, 640), (18432, 320, 320)]:
    A, B = mk(*shape), mk(*shape)
    chain(A, 2); chain(B, 2); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        chain(A, 100); chain(B, 100)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side): chain(B, 100)
        chain(A, 100)
        cur.wait_stream(side)
    t1, t2 = timed(g1.replay), timed(g2.replay)
    print(shape, "200 gemms: one chain %.2f ms (%.1f us each), two streams %.2f ms  -> x%.2f" % (t1, t1 * 5, t2, t1 / t2), flush=True)
# floor: a chain of tiny kernels
t = torch.full((2,), 500, dtype=torch.long, device=dev)
def tiny(reps):
    for _ in range(reps): E.timestep_embedding(t, 320)
tiny(2); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): tiny(200)
tt = timed(g.replay)
print("200 tiny kernels in a chain: %.2f ms (%.2f us each)" % (tt, tt * 5))
x = E.new_act(2, 24, 24, 1280, dev)
gam = torch.ones(1280, device=dev); bet = torch.zeros(1280, device=dev)
def lnc(reps):
    for _ in range(reps): E.layernorm(x, gam, bet, 1e-5)
lnc(2); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): lnc(200)
tt = timed(g.replay)
print("200 layernorms (1152 x 1280) in a chain: %.2f ms (%.2f us each)" % (tt, tt * 5))
