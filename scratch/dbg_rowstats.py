import sys, numpy as np, torch
sys.path.insert(0, ".")
import stable_diffusion_amd; stable_diffusion_amd.install_dropin()
from stable_diffusion_amd import engine as E
T = torch.from_numpy
h16 = lambda x: x.to(E.act_dtype()).float()
for (m, n, k) in [(2304, 320, 320), (4608, 640, 1280), (2304, 1280, 640), (256, 128, 128), (4608, 640, 320), (4608, 320, 1280)]:
    rs = np.random.RandomState(m + n)
    a, w = h16(T(rs.randn(m, k).astype(np.float32))), h16(T((rs.randn(n, k) / np.sqrt(k)).astype(np.float32)))
    b, r = T(rs.randn(n).astype(np.float32)), h16(T(rs.randn(m, n).astype(np.float32)))
    ref = (a.double() @ w.double().t() + b.double() + r.double())
    A = E.Act(a.to(E.act_dtype()).cuda().reshape(1, m, k), 1, 1, m, k)
    R = E.Act(r.to(E.act_dtype()).cuda().reshape(1, m, n), 1, 1, m, n)
    out = E.gemm(A, w.to(E.act_dtype()).cuda().contiguous(), n, bias=b.cuda(), residual=R, colstats=True, rowstats=True)
    torch.cuda.synchronize()
    print(m, n, k, E.L.load().sdmi_last_kernel().decode())
    rst, parts = out.rstats
    tn = n // parts
    want = torch.stack([ref.reshape(m, parts, tn).sum(-1), (ref * ref).reshape(m, parts, tn).sum(-1)], dim=-1)
    d = (rst.double().cpu() - want).abs()
    bad = (d > 1e-3 * want.abs().max()).nonzero()
    print("  parts", parts, "tn", tn, "max abs diff", float(d.max()), "bad entries", bad.shape[0])
    if bad.shape[0]:
        print("  first bad:", bad[:12].tolist())
        rows = sorted(set(int(x[0]) for x in bad))
        print("  bad rows mod 64:", sorted(set(r % 64 for r in rows))[:40], "parts:", sorted(set(int(x[1]) for x in bad)))
        i = bad[0]
        print("  got", rst[i[0], i[1]].tolist(), "want", want[i[0], i[1]].tolist())
    if out.stats:
        cs, pps, cc = out.stats[0]
        tm = m // pps
        wantc = torch.stack([ref.reshape(pps, tm, n).sum(1), (ref * ref).reshape(pps, tm, n).sum(1)], dim=-1)
        dc = (cs.double().cpu() - wantc).abs()
        print("  colstats parts", pps, "max abs diff", float(dc.max()), "rel", float(dc.max() / wantc.abs().max()))
    if bad.shape[0]:
        o = out.buf.reshape(m, n).float().cpu().double()
        for i in bad[:3]:
            r0, p0 = int(i[0]), int(i[1])
            seg_ref = ref[r0, p0 * tn:(p0 + 1) * tn]
            seg_out = o[r0, p0 * tn:(p0 + 1) * tn]
            print("  row", r0, "part", p0, "sum ref", float(seg_ref.sum()), "sumsq ref", float((seg_ref ** 2).sum()), "sumsq out", float((seg_out ** 2).sum()),
                  "max|ref|", float(seg_ref.abs().max()), "max|ref-out|", float((seg_ref - seg_out).abs().max()))
            print("   ref", [round(float(v), 3) for v in seg_ref])
            print("   out", [round(float(v), 3) for v in seg_out])
