"""In-kernel split-K reduction vs the separate reduce kernel on the UNet's split shapes: must agree bit for bit."""
import os, sys, torch
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib
dev = "cuda"; H = E.act_dtype()
torch.manual_seed(0)
def run(n, hw, cin, cout, mode):
    os.environ["SDMI_SPLITK_INKERNEL"] = mode  # read per call by the library
    out = E.new_act(n, hw, hw, cout, dev)
    E.conv3x3(X, W, B, residual=R, out=out)
    torch.cuda.synchronize()
    return out.buf.clone(), _lib.load().sdmi_last_kernel().decode()
for (n, hw, cin, cout) in [(2, 48, 640, 640), (2, 48, 1280, 640), (2, 24, 1280, 1280), (2, 24, 2560, 1280), (2, 12, 1280, 1280), (2, 12, 2560, 1280), (1, 12, 1280, 1280)]:
    X = E.Act(torch.randn(n, hw * hw, cin, device=dev).to(H), n, hw, hw, cin)
    W = torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5
    B = torch.randn(cout, device=dev)
    R = E.new_act(n, hw, hw, cout, dev); R.buf.copy_(torch.randn_like(R.buf.float()).to(H))
    ref, k0 = run(n, hw, cin, cout, "0")
    bad = 0
    for it in range(10):
        got, k1 = run(n, hw, cin, cout, "1")
        bad += int(not torch.equal(got.view(torch.int16), ref.view(torch.int16)))
    print((n, hw, cin, cout), "|", k0, "|", k1, "| mismatching runs:", bad, flush=True)
print("done")
