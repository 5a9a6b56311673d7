"""Per-launch timeline of ONE batch-1 UNet forward at 768x768 (one CFG half as the sampler runs it): every C-ABI call
bracketed by HIP events behind a spin kernel; grouped by (kind, rows) to show where a step's time goes."""
import sys, collections, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stable_diffusion_amd import _lib, engine as E

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
unet, vae = bench.build_models(dev)
unet = unet.to(dev)
x = torch.randn(B, 9, L, L, device=dev)
ctx = torch.randn(B, 77, 1024, device=dev)
t = torch.full((B,), 501, dtype=torch.long, device=dev)
cc = unet.precompute_context(ctx)
tb = unet.precompute_time_embedding(t)
unet(x, timesteps=tb, context=cc); torch.cuda.synchronize()

recs = []
def work(name, a):
    return (name, a)
class T(_lib.KernelTimer):
    pass
with _lib.KernelTimer(lambda name, a: (name, a)) as kt:
    torch.cuda._sleep(int(80e6))
    unet(x, timesteps=tb, context=cc)
torch.cuda.synchronize()
agg = collections.OrderedDict()
tot = 0.0
for name, work, e0, e1 in kt.records:
    ms = e0.elapsed_time(e1)
    kname, a = work if isinstance(work[0], str) else (name, ())
    if kname.startswith("conv_win") or (kname.startswith("igemm_kernel") and "true," in kname):
        n, h, w, cin, cout, ho, wo = a[2], a[3], a[4], a[5], a[7], a[8], a[9]
        key = ("conv3x3", n * ho * wo, cout, 9 * cin); fl = 2.0 * n * ho * wo * cout * 9 * cin
    elif kname.startswith("igemm_kernel"):
        m, n, k, b = a[9], a[10], a[11], a[12]
        key = ("gemm" + ("-geglu" if a[20] & 8 else ""), m, n, k); fl = 2.0 * m * n * k * b
    elif kname.startswith("attn64"):
        b, h, nq, nk = a[12], a[13], a[14], a[15]
        key = ("attn", nq, nk, h); fl = 4.0 * b * h * nq * nk * 64
    elif kname == "igemm_splitk_reduce_kernel":
        key = ("splitk_reduce", 0, 0, 0); fl = 0
    elif kname == "sdmi_groupnorm":
        key = ("groupnorm", a[5], a[6], 0); fl = 0
    elif kname == "sdmi_layernorm":
        key = ("layernorm", a[7], a[8], 0); fl = 0
    else:
        key = (kname, 0, 0, 0); fl = 0
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += ms; d[2] += fl
    tot += ms
print("total %.3f ms over %d launches (L=%d, batch %d)" % (tot, len(kt.records), L, B))
print("%-16s %8s %6s %6s %5s %9s %8s %8s" % ("kind", "rows", "N", "K", "calls", "total_us", "avg_us", "TFLOP/s"))
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-16s %8d %6d %6d %5d %9.1f %8.1f %8.1f" % (key[0][:16], key[1], key[2], key[3], d[0], d[1] * 1e3, d[1] * 1e3 / d[0], d[2] / (d[1] * 1e-3) / 1e12 if d[2] else 0))
bykind = collections.Counter()
for key, d in agg.items():
    bykind[key[0]] += d[1]
print({k: round(v, 3) for k, v in bykind.most_common()})
bylevel = collections.Counter()
for key, d in agg.items():
    bylevel[key[1]] += d[1]
print({k: round(v, 3) for k, v in bylevel.most_common(12)})
