"""Repeat the statistics check many times: any nondeterministic loss shows up as a bad entry count > 0."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import stable_diffusion_amd; stable_diffusion_amd.install_dropin()
from stable_diffusion_amd import engine as E
T = torch.from_numpy
h16 = lambda x: x.to(E.act_dtype()).float()
for (m, n, k) in [(4608, 640, 1280), (9216, 320, 320), (2304, 1280, 640)]:
    rs = np.random.RandomState(m + n)
    a, w = h16(T(rs.randn(m, k).astype(np.float32))), h16(T((rs.randn(n, k) / np.sqrt(k)).astype(np.float32)))
    b, r = T(rs.randn(n).astype(np.float32)), h16(T(rs.randn(m, n).astype(np.float32)))
    ref = (a.double() @ w.double().t() + b.double() + r.double())
    A = E.Act(a.to(E.act_dtype()).cuda().reshape(1, m, k), 1, 1, m, k)
    R = E.Act(r.to(E.act_dtype()).cuda().reshape(1, m, n), 1, 1, m, n)
    wd = w.to(E.act_dtype()).cuda().contiguous(); bd = b.cuda()
    nbad_r = nbad_c = 0
    first = None
    for it in range(40):
        out = E.gemm(A, wd, n, bias=bd, residual=R, colstats=True, rowstats=True)
        rst, parts = out.rstats
        tn = n // parts
        if first is None:
            want = torch.stack([ref.reshape(m, parts, tn).sum(-1), (ref * ref).reshape(m, parts, tn).sum(-1)], dim=-1)
            cs, pps, cc = out.stats[0]
            tm = m // pps
            wantc = torch.stack([ref.reshape(pps, tm, n).sum(1), (ref * ref).reshape(pps, tm, n).sum(1)], dim=-1)
            first = (rst.clone(), cs.clone())
        d = (rst.double().cpu() - want).abs()
        nbad_r += int((d > 1e-3 * want.abs().max()).sum())
        dc = (out.stats[0][0].double().cpu() - wantc).abs()
        nbad_c += int((dc > 1e-4 * wantc.abs().max()).sum())
        assert torch.equal(rst, first[0]) and torch.equal(out.stats[0][0], first[1]), "not deterministic at iteration %d" % it
    print(m, n, k, "bad rowstats", nbad_r, "bad colstats", nbad_c, "over 40 runs; bit-identical across runs")
