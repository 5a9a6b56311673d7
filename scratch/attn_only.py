"""The 9216-token, 5-head self-attention of the UNet's 96x96 level (CFG batch 2) a few times: workload of the SQ counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
if len(sys.argv) > 1:
    _lib.load().sdmi_set_option(b"attn_variant", int(sys.argv[1]))
b, heads, nq, nkv = 2, 5, 9216, 9216
c = heads * 64
torch.manual_seed(0)
q = E.Act(torch.randn(b, nq, c, device=dev).to(H), b, 1, nq, c)
k = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
v = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
o = E.new_act(b, 1, nq, c, dev)
for _ in range(5): E.attention(q, k, v, heads, out=o, log2q=True)
torch.cuda.synchronize()
