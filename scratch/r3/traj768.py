"""The headline configuration BY VALUE: 768x768, batch 1, 50 DDIM steps (eta = 1, injected noise), CFG scale 9 - every latent
of the trajectory and the final image of the HIP sampler (graph replay) against the fp32 CPU oracle (oracle/sd_oracle.py,
the restatement of the reference's sample()).  ~5 minutes of host time for the oracle; a one-off measurement
(profiles/r03_trajectory_768.txt), the test suite runs the same check at 256x256 / 512x512.
    python scratch/r3/traj768.py [latent_size=96] [steps=50]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import stable_diffusion_amd
stable_diffusion_amd.install_dropin()
import sd_oracle as O
from conftest import rel_err
from stable_diffusion_amd.synth import fill_state_dict, make_inputs
from Unet import Unet
from AutoencoderKL import AutoencoderKL
from stable_diffusion_amd.sampler import Sampler

L = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.set_num_threads(min(16, os.cpu_count() or 16))
DEV = "cuda"
unet, vae = Unet(), AutoencoderKL()
sd_u = {k: v.clone() for k, v in fill_state_dict(unet.state_dict(), seed=0).items()}
sd_v = {k: v.clone() for k, v in fill_state_dict(vae.state_dict(), seed=1).items()}
unet, vae = unet.to(DEV).eval(), vae.to(DEV).eval()
inp = make_inputs(1, L, steps, seed=100)
dev = lambda t: t.to(DEV)
trace = []
t0 = time.time()
x, img = Sampler(unet, vae).sample(steps, dev(inp["cond"]), dev(inp["uncond"]), dev(inp["c_concat"]), dev(inp["x_T"]),
                                   [dev(n) for n in inp["noises"]], 9.0, use_graph=True, trace=trace)
torch.cuda.synchronize()
print("HIP sampler: %d steps at latent %dx%d done (%.1f s incl. capture)" % (steps, L, L, time.time() - t0), flush=True)
gpu_trace = [t.float().cpu().numpy() for t in trace]


class Progress(list):
    def append(self, v):
        i = len(self)
        m, l2 = rel_err(gpu_trace[i], v.numpy())
        print("step %2d: latent max-rel %.3e rel-L2 %.3e   (oracle %.0f s)" % (i, m, l2, time.time() - t0), flush=True)
        super().append(v)


ref_trace = Progress()
with torch.no_grad():
    ref_x, ref_img = O.sample(sd_u, sd_v, steps, inp["cond"], inp["uncond"], inp["c_concat"], inp["x_T"], inp["noises"], 9.0, trace=ref_trace)
errs = [rel_err(g, r.numpy()) for g, r in zip(gpu_trace, ref_trace)]
mse = np.mean((img.cpu().numpy().astype(np.float64) - ref_img.astype(np.float64)) ** 2)
psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
d = np.abs(img.cpu().numpy().astype(np.int32) - ref_img.astype(np.int32))
print("%d steps at %dx%d px: worst per-step rel-L2 %.3e (step %d), worst max-rel %.3e, last step rel-L2 %.3e; final image PSNR %.2f dB, "
      "max |diff| %d LSB, %.2f %% of the pixels differ by more than 1 LSB" %
      (steps, 8 * L, 8 * L, max(e[1] for e in errs), int(np.argmax([e[1] for e in errs])), max(e[0] for e in errs), errs[-1][1], psnr,
       int(d.max()), 100.0 * float((d > 1).mean())), flush=True)
