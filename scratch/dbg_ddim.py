import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import stable_diffusion_amd
import sd_oracle as O
from stable_diffusion_amd import engine as E
rs = np.random.RandomState(8)
b = 3
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
x, eu, ec, nz = (_t(rs.randn(b, 4, 16, 16)) for _ in range(4))
sch = O.ddim_schedule(50)
for index in (0, 17, 49):
    a_t, a_p, sg, s1 = sch["alphas"][index], sch["alphas_prev"][index], sch["sigmas"][index], sch["sqrt_one_minus_alphas"][index]
    print(type(a_t), type(a_p), type(sg), sg.dtype, type(s1), torch.full((1,), sg).dtype)
    ref, ref0 = O.cfg_ddim_update(x, eu, ec, 9.0, a_t, a_p, sg, s1, nz)
    f = lambda v: float(torch.full((1,), v).item())
    p0 = torch.empty((b, 4, 16, 16), device="cuda")
    epsd, xd, nd = torch.cat([eu, ec]).cuda(), x.cuda(), nz.cuda()
    out = E.cfg_ddim_step(epsd, xd, nd, 9.0, f(s1), f(a_t), f(a_p), f(sg), pred_x0=p0)
    d = (out.cpu() - ref).abs(); d0 = (p0.cpu() - ref0).abs()
    print(index, "x:", d.max().item(), (d > 0).sum().item(), "px0:", d0.max().item(), (d0 > 0).sum().item())
    # numpy fp32 emulation
    F = np.float32
    e = eu.numpy() + F(9.0) * (ec.numpy() - eu.numpy())
    px0 = (x.numpy() - F(f(s1)) * e) / np.sqrt(F(f(a_t)))
    print("  numpy px0 vs ref0", np.abs(px0 - ref0.numpy()).max(), " vs gpu", np.abs(px0 - p0.cpu().numpy()).max())
    dirc = np.sqrt(F(1.0) - F(f(a_p)) - F(f(sg)) * F(f(sg)))
    xn = np.sqrt(F(f(a_p))) * px0 + dirc * e + F(f(sg)) * nz.numpy() * F(1.0)
    print("  numpy x vs ref", np.abs(xn - ref.numpy()).max(), " vs gpu", np.abs(xn - out.cpu().numpy()).max())
