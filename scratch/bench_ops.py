"""Per-shape timing of the conv / gemm / attention entry points at the 768x768 UNet + VAE shapes."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd import engine as E, _lib as L
dev = "cuda"
H = E.act_dtype()

def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def conv(n, cin, cout, h, up=False, stride=1):
    a = E.Act(torch.randn(n, h*h, cin, device=dev).to(H), n, h, h, cin)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)); b = torch.zeros(cout, device=dev)
    ho = h*2 if up else (h // stride)
    out = E.new_act(n, ho, ho, cout, dev)
    ms = timeit(lambda: E.conv3x3(a, w, b, stride=stride, upsample=up, out=out))
    fl = 2.0*n*ho*ho*cout*9*cin
    print("conv %4d->%4d @%3d n=%d%s%s: %8.3f ms  %7.1f TF/s" % (cin, cout, h, n, " up" if up else "", " s2" if stride == 2 else "", ms, fl/ms/1e9))
    return ms

def gemm(m, n, k, geglu=False):
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    out = E.new_act(1, 1, m, n // 2 if geglu else n, dev)
    ms = timeit(lambda: E.gemm(a, w, n, geglu=geglu, out=out))
    print("gemm M=%6d N=%5d K=%5d%s: %8.3f ms  %7.1f TF/s" % (m, n, k, " geglu" if geglu else "", ms, 2.0*m*n*k/ms/1e9))
    return ms

def attn(b, heads, nq, nk):
    q = E.Act(torch.randn(b, nq, heads*64, device=dev).to(H), b, 1, nq, heads*64)
    kv = E.Act(torch.randn(b, nk, 2*heads*64, device=dev).to(H), b, 1, nk, 2*heads*64)
    o = E.new_act(b, 1, nq, heads*64, dev)
    ms = timeit(lambda: E.attention(q, kv.cols(0, heads*64), kv.cols(heads*64, heads*64), heads, out=o))
    print("attn b=%d h=%2d nq=%5d nk=%5d: %8.3f ms  %7.1f TF/s" % (b, heads, nq, nk, ms, 4.0*b*heads*nq*nk*64/ms/1e9))
    return ms

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "conv"):
    print("--- UNet convs (Bc=2) ---")
    tot = 0
    for cnt, args in [(7, (2,320,320,96)), (6, (2,640,640,48)), (7, (2,1280,1280,24)), (11, (2,1280,1280,12)), (3, (2,2560,1280,12)), (2, (2,2560,1280,24)),
                      (1, (2,1920,1280,24)), (1, (2,1920,640,48)), (1, (2,1280,640,48)), (1, (2,960,640,48)), (1, (2,960,320,96)), (2, (2,640,320,96)),
                      (1, (2,320,640,48)), (1, (2,640,1280,24))]:
        tot += cnt * conv(*args)
    tot += conv(2, 640, 640, 48, up=True) + conv(2, 1280, 1280, 24, up=True) + conv(2, 1280, 1280, 12, up=True)
    tot += conv(2, 320, 320, 96, stride=2) + conv(2, 640, 640, 48, stride=2) + conv(2, 1280, 1280, 24, stride=2)
    print("UNet conv total per step: %.2f ms" % tot)
    print("--- VAE convs (B=1) ---")
    tot = 0
    for cnt, args in [(10, (1,512,512,96)), (7, (1,512,512,192)), (1, (1,512,512,384)), (1, (1,512,256,384)), (5, (1,256,256,384)), (1, (1,256,256,768)),
                      (1, (1,256,128,768)), (5, (1,128,128,768)), (1, (1,128,8,768))]:
        tot += cnt * conv(*args)
    print("VAE conv total per decode (w/o upsample variants): %.2f ms" % tot)
if which in ("all", "gemm"):
    print("--- UNet linears (Bc=2) ---")
    tot = 0
    for cnt, args in [(10, (18432, 960, 320)), (20, (18432, 320, 320)), (5, (18432, 2560, 320, True)), (5, (18432, 320, 1280)),
                      (10, (4608, 1920, 640)), (20, (4608, 640, 640)), (5, (4608, 5120, 640, True)), (5, (4608, 640, 2560)),
                      (10, (1152, 3840, 1280)), (20, (1152, 1280, 1280)), (5, (1152, 10240, 1280, True)), (5, (1152, 1280, 5120)),
                      (2, (288, 3840, 1280)), (4, (288, 1280, 1280)), (1, (288, 10240, 1280, True)), (1, (288, 1280, 5120)),
                      (5, (154, 640, 1024)), (5, (154, 1280, 1024)), (6, (154, 2560, 1024)),
                      (3, (288, 1280, 2560)), (2, (1152, 1280, 2560)), (1, (1152, 1280, 1920)), (1, (4608, 640, 1920)), (1, (4608, 640, 1280)), (1, (4608, 640, 960)), (1, (18432, 320, 960)), (2, (18432, 320, 640))]:
        tot += cnt * gemm(*args)
    print("UNet gemm total per step: %.2f ms" % tot)
if which in ("all", "attn"):
    print("--- attention (Bc=2) ---")
    tot = 0
    for cnt, args in [(5, (2,5,9216,9216)), (5, (2,10,2304,2304)), (5, (2,20,576,576)), (1, (2,20,144,144)), (5, (2,5,9216,77)), (5, (2,10,2304,77)), (5, (2,20,576,77)), (1, (2,20,144,77))]:
        tot += cnt * attn(*args)
    print("attention total per step: %.2f ms" % tot)
