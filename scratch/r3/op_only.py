"""Workload of counter / trace passes: runs each op spec of argv a few times.  Specs:
   gemm:M:N:K[:res][:ln]   geglu:M:N:K   conv:HW:CIN:COUT[:BATCH]   attn:B:H:NQ:NKV
(ln = LayerNorm folded in: the producer's row statistics come from a first GEMM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
torch.manual_seed(0)
REP = int(os.environ.get("REP", "5"))
lib = _lib.load()
for kv in filter(None, os.environ.get("SDMI_OPTS", "").split(",")):   # e.g. SDMI_OPTS=attn_pp=-1
    lib.sdmi_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
for spec in sys.argv[1:]:
    f = spec.split(":")
    kind = f[0]
    if kind in ("gemm", "geglu"):
        m, n, k = int(f[1]), int(f[2]), int(f[3])
        opts = f[4:]
        a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H).contiguous()
        b = torch.randn(n, device=dev)
        nout = n // 2 if kind == "geglu" else n
        res = E.Act(torch.randn(1, m, nout, device=dev).to(H), 1, 1, m, nout) if "res" in opts else None
        out = E.new_act(1, 1, m, nout, dev)
        for _ in range(REP):
            E.gemm(a, w, n, bias=b, residual=res, geglu=(kind == "geglu"), out=out)
    elif kind == "conv":
        hw, cin, cout = int(f[1]), int(f[2]), int(f[3])
        n = int(f[4]) if len(f) > 4 else 1
        x = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
        W = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
        b = torch.zeros(cout, device=dev)
        out = E.new_act(n, hw, hw, cout, dev); res = E.new_act(n, hw, hw, cout, dev)
        for _ in range(REP):
            E.conv3x3(x, W, b, residual=res, out=out)
    elif kind == "attn":
        b_, h, nq, nkv = int(f[1]), int(f[2]), int(f[3]), int(f[4])
        q = E.Act((torch.randn(b_, nq, h * 64, device=dev) * E.Q_LOG2_SCALE).to(H), b_, 1, nq, h * 64)
        kk = E.Act(torch.randn(b_, nkv, h * 64, device=dev).to(H), b_, 1, nkv, h * 64)
        v = E.Act(torch.randn(b_, nkv, h * 64, device=dev).to(H), b_, 1, nkv, h * 64)
        for _ in range(REP):
            E.attention(q, kk, v, h, log2q=True)
    torch.cuda.synchronize()
    print(spec, "->", lib.sdmi_last_kernel().decode())
