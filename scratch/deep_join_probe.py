"""Would joining the two CFG halves into one batch-2 pass at the 24x24 / 12x12 levels pay?  Proxy: the stages
input_blocks[7], [8] (24x24, 1280 ch, ResBlock + transformer), [10], [11] and the middle block (12x12) run (a) as two
concurrent batch-1 passes on two streams (what the sampler does) and (b) as one batch-2 pass, both as captured graphs."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stable_diffusion_amd import _lib, engine as E
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
unet, _ = bench.build_models(dev); unet = unet.to(dev)
H = E.act_dtype()
def seg(n, ctx, tb):
    keep, emb, ldrb = unet._emb_pointers(tb.table)
    h = E.Act(torch.randn(n, 576, 640, device=dev).to(H), n, 24, 24, 640)
    for i in (7, 8):
        h = unet._run_stage(unet.input_blocks[i], h, emb, ldrb, ctx, None)
    g = E.Act(torch.randn(n, 144, 1280, device=dev).to(H), n, 12, 12, 1280)
    for i in (10, 11):
        g = unet._run_stage(unet.input_blocks[i], g, emb, ldrb, ctx, None)
    g = unet.middle_block[0].run(g, emb[id(unet.middle_block[0])], ldrb)
    g = unet.middle_block[1].run(g, ctx)
    g = unet.middle_block[2].run(g, emb[id(unet.middle_block[2])], ldrb)
    return h, g
c1 = [unet.precompute_context(torch.randn(1, 77, 1024, device=dev)) for _ in range(2)]
c2 = unet.precompute_context(torch.randn(2, 77, 1024, device=dev))
t1 = unet.precompute_time_embedding(torch.full((1,), 501, dtype=torch.long, device=dev))
t2 = unet.precompute_time_embedding(torch.full((2,), 501, dtype=torch.long, device=dev))
side = torch.cuda.Stream()
def two_streams():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        a = seg(1, c1[1], t1)
    b = seg(1, c1[0], t1)
    cur.wait_stream(side)
    return a, b
def joined():
    return seg(2, c2, t2)
graphs = {}
for name, fn in (("two batch-1 streams", two_streams), ("one batch-2 pass", joined)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    graphs[name] = (g, keep)
res = {k: [] for k in graphs}
for r in range(6):
    for name, (g, _) in graphs.items():
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, v in res.items():
    print("%-22s deep-level segment ms: median %.3f  [%s]" % (name, float(np.median(v)), " ".join("%.3f" % q for q in v)), flush=True)
