"""One conv / gemm shape, a few launches, for PMC runs: python scratch/one_conv.py conv 1 512 512 384 | gemm M N K"""
import sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E
dev = "cuda"; H = E.act_dtype()
kind = sys.argv[1]
if kind == "conv":
    n, cin, cout, h = map(int, sys.argv[2:6])
    a = E.Act(torch.randn(n, h*h, cin, device=dev).to(H), n, h, h, cin)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)); b = torch.zeros(cout, device=dev)
    out = E.new_act(n, h, h, cout, dev)
    fn = lambda: E.conv3x3(a, w, b, out=out)
elif kind == "gemm":
    m, n, k = map(int, sys.argv[2:5])
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n, dev)
    fn = lambda: E.gemm(a, w, n, out=out)
else:
    b, heads, nq, nk = map(int, sys.argv[2:6])
    q = E.Act(torch.randn(b, nq, heads*64, device=dev).to(H), b, 1, nq, heads*64)
    kv = E.Act(torch.randn(b, nk, 2*heads*64, device=dev).to(H), b, 1, nk, 2*heads*64)
    o = E.new_act(b, 1, nq, heads*64, dev)
    fn = lambda: E.attention(q, kv.cols(0, heads*64), kv.cols(heads*64, heads*64), heads, out=o)
for _ in range(5):
    fn()
torch.cuda.synchronize()
