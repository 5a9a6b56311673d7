import sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
b, heads, nq, nkv = 2, 5, 9216, 9216
c = heads * 64
q = E.Act(torch.randn(b, nq, c, device=dev).to(H), b, 1, nq, c)
k = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
v = E.Act(torch.randn(b, nkv, c, device=dev).to(H), b, 1, nkv, c)
o = E.new_act(b, 1, nq, c, dev)
for _ in range(3): E.attention(q, k, v, heads, out=o)
torch.cuda.synchronize()
