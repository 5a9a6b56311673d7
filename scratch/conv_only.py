"""One UNet conv shape a few times: workload of SQ counter passes.  argv: hw cin cout [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
torch.manual_seed(0)
x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
b = torch.zeros(cout, device=dev)
out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
for _ in range(5): E.conv3x3(x, W, b, residual=res, out=out)
torch.cuda.synchronize()
print(_lib.load().sdmi_last_kernel().decode())
