"""Same-process A/B of the sampler's UNet step (768x768, CFG batch 2 as two concurrent batch-1 passes, HIP-graph replay):
every configuration gets its own Sampler + captured graph, timing rounds are interleaved across configurations.

    python scratch/ab_step.py name1:opt=val,ENV=val name2:...      opt = sdmi_set_option name, UPPERCASE = engine module flag
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stable_diffusion_amd import _lib, engine as E
from stable_diffusion_amd.sampler import Sampler
dev = torch.device("cuda", 0)
lib = _lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet = unet.to(dev)
L = int(os.environ.get("AB_L", "96"))      # latent size (96 = 768 px); AB_B = images per GPU
B = int(os.environ.get("AB_B", "1"))
x = torch.randn(B, 9, L, L, device=dev).repeat(2, 1, 1, 1)
t = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
uncond, cond = torch.randn(B, 77, 1024, device=dev), torch.randn(B, 77, 1024, device=dev)
te = unet.precompute_time_embedding(t)

OPTION_DEFAULTS = {"conv_plan_r2": 0, "attn_pp": 0, "conv_win_256": 0, "attn_variant": 0, "gemm_small_m_tile": 1, "conv_nmajor_by_rows": 0, "splitk_target": 0, "splitk_max_tiles": 0, "attn_x2_min_nq": 0, "splitk_min_kt_conv": 0, "splitk_min_kt_gemm": 0, "bm128_min": 0, "gemm_t2_min_blocks": 0, "split_colstats_off": 0, "gemm_persist": 0, "conv_win_v1": 0, "gemm_buf_off": 0, "conv_win_ups_off": 0, "gn_apply_rows": 0}
def apply(cfg):
    for k, v in OPTION_DEFAULTS.items():        # options are process-wide: every configuration starts from the defaults
        lib.sdmi_set_option(k.encode(), v)
    for k, v in cfg.items():
        if k.isupper():
            setattr(E, k, type(getattr(E, k))(int(v)))
        else:
            assert lib.sdmi_set_option(k.encode(), int(v)) == 0, k

configs = []
for a in sys.argv[1:] or ["base:"]:
    name, _, rest = a.partition(":")
    configs.append((name, dict(kv.split("=") for kv in rest.split(",") if kv)))
defaults = {}
for _, cfg in configs:
    for k in cfg:
        defaults.setdefault(k, None)
runs = []
for name, cfg in configs:
    apply(cfg)
    s = Sampler(unet, None)
    cc = s.make_context(uncond, cond)
    s._unet_step(x, te, cc, True, same_halves=True)
    torch.cuda.synchronize()
    runs.append((name, cfg, s, cc))
res = {name: [] for name, *_ in runs}
for r in range(6):
    for name, cfg, s, cc in runs:
        s._graph.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): s._graph.replay()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, v in res.items():
    print("%-24s unet step ms: median %.3f  min %.3f   [%s]" % (name, float(np.median(v)), min(v), " ".join("%.3f" % q for q in v)), flush=True)
