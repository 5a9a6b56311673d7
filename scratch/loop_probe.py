"""Where do the ~18 ms per image outside 50 x UNet-step + decode go?  CPU time of each call in the sampling loop."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
from stable_diffusion_amd.sampler import Sampler, ddim_schedule, _f32
from stable_diffusion_amd.synth import make_inputs
import bench
dev = torch.device("cuda", 0)
_lib.load(); torch.set_grad_enabled(False)
unet, vae = bench.build_models(dev)
unet, vae = unet.to(dev), vae.to(dev)
L, b, steps = 96, 1, 50
inp = make_inputs(b, L, steps, seed=100)
g = lambda t: t.to(dev)
cond, uncond, c_concat, x_T = g(inp["cond"]), g(inp["uncond"]), g(inp["c_concat"]), g(inp["x_T"])
noises = [g(n) for n in inp["noises"]]
s = Sampler(unet, vae)
for _ in range(2):
    s.sample(steps, cond, uncond, c_concat, x_T, noises, 9.0, use_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); s.sample(steps, cond, uncond, c_concat, x_T, noises, 9.0, use_graph=True); torch.cuda.synchronize()
print("sample(): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); s.sample(steps, cond, uncond, c_concat, x_T, noises, 9.0, use_graph=True, decode=False); torch.cuda.synchronize()
print("sample(decode=False): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); cc = s.make_context(uncond, cond); torch.cuda.synchronize()
print("make_context: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); s.decode_to_uint8(x_T); torch.cuda.synchronize()
print("decode_to_uint8: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
# the loop, call by call (CPU-side durations)
ts, a_t, a_prev, sig, s1m = ddim_schedule(steps, 1.0)
xc = torch.empty((2, 9, L, L), device=dev); xc[:, :4] = x_T; xc[:, 4:] = c_concat
emb_table = s._time_tables(np.flip(ts), 2, dev); emb_cur = s._emb_static(emb_table, dev)
x = x_T.clone()
s._unet_step(xc, emb_cur, cc, True); torch.cuda.synchronize()
tc = tr = td = 0.0
t_all = time.perf_counter()
for i in range(steps):
    index = steps - i - 1
    t0 = time.perf_counter(); emb_cur.table.copy_(emb_table[i]); t1 = time.perf_counter()
    eps = s._unet_step(xc, emb_cur, cc, True); t2 = time.perf_counter()
    E.cfg_ddim_step(eps, x, noises[i], 9.0, _f32(s1m[index]), _f32(a_t[index]), _f32(a_prev[index]), _f32(sig[index]), 1.0, x_out=x, xc_out=xc)
    t3 = time.perf_counter()
    tc += t1 - t0; tr += t2 - t1; td += t3 - t2
torch.cuda.synchronize()
print("loop: %.1f ms total; CPU per step: copy_ %.1f us, replay %.1f us, cfg_ddim %.1f us" % ((time.perf_counter() - t_all) * 1e3, tc / steps * 1e6, tr / steps * 1e6, td / steps * 1e6))
t0 = time.perf_counter()
for i in range(steps): s._graph.replay()
torch.cuda.synchronize()
print("50 bare replays: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
