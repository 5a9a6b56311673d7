import sys, time, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
def timed(g, reps, iters=5):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / reps * 1e6
nq = nkv = 8192
for heads in (1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24):
    b, c = 1, heads * 64
    q = E.Act(torch.randn(b, nq, c, device=dev).to(H), b, 1, nq, c)
    k = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
    v = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
    o = E.new_act(b, 1, nq, c, dev)
    fn = lambda: E.attention(q, k, v, heads, out=o)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5): fn()
    t = timed(g, 5)
    print("blocks %4d (%.2f per CU): %7.1f us  %6.0f TF" % (64 * heads, 64 * heads / 256, t, 4.0 * heads * nq * nkv * 64 / t / 1e6), flush=True)
