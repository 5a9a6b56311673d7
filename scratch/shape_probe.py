"""One eager UNet step, every C-ABI launch timed by events: time aggregated by (kernel, algorithmic flops) = by shape."""
import sys, collections, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import _lib
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet = unet.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn(B, 9, 96, 96, device=dev); t = torch.full((B,), 501, dtype=torch.long, device=dev)
ctx = unet.precompute_context(torch.randn(B, 77, 1024, device=dev)); te = unet.precompute_time_embedding(t)
for _ in range(2): unet(x, timesteps=te, context=ctx)
torch.cuda.synchronize()
with _lib.KernelTimer(bench.work_of) as kt:
    torch.cuda._sleep(int(60e6))
    unet(x, timesteps=te, context=ctx)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, work, a, c in kt.records:
    k = (name, round(work[0] / 1e9, 1), round(work[1] / 1e6, 1))
    agg[k][0] += 1; agg[k][1] += a.elapsed_time(c) * 1e3
tot = sum(v[1] for v in agg.values())
print("total %.0f us over %d launches" % (tot, len(kt.records)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    tf = k[1] * 1e9 * v[0] / (v[1] * 1e-6) / 1e12 if k[1] else 0
    print("%-52s GF %7.1f MB %6.1f  x%2d  %7.1f us (%5.1f each) %5.1f%%  %4.0f TF" % (k[0], k[1], k[2], v[0], v[1], v[1] / v[0], 100 * v[1] / tot, tf))
