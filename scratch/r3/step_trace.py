"""Workload for a kernel trace of the captured two-stream UNet step in graph replay: 3 warm replays, a marker kernel, then
ONE replay.  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scratch/r3/step_trace.py [opt=val ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from stable_diffusion_amd import _lib, engine as E
from stable_diffusion_amd.sampler import Sampler
dev = torch.device("cuda", 0)
lib = _lib.load(); torch.set_grad_enabled(False)
for a in sys.argv[1:]:
    k, v = a.split("=")
    assert lib.sdmi_set_option(k.encode(), int(v)) == 0, k
unet, vae = bench.build_models(dev)
unet = unet.to(dev)
L = int(os.environ.get("LAT", "96"))
x = torch.randn(1, 9, L, L, device=dev).repeat(2, 1, 1, 1)
t = torch.full((2,), 501, dtype=torch.long, device=dev)
uncond, cond = torch.randn(1, 77, 1024, device=dev), torch.randn(1, 77, 1024, device=dev)
te = unet.precompute_time_embedding(t)
s = Sampler(unet, None)
cc = s.make_context(uncond, cond)
s._unet_step(x, te, cc, True, same_halves=True)
torch.cuda.synchronize()
for _ in range(3): s._graph.replay()
torch.cuda.synchronize()
marker = torch.zeros(1 << 20, device=dev)
marker.add_(1.0)            # marker kernel: the trace is cut after the last elementwise kernel
torch.cuda.synchronize()
s._graph.replay()
torch.cuda.synchronize()
