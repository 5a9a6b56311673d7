"""Where a GEMM block's time goes (IGEMM_STAMP build: s_memrealtime at kernel start, end of prologue, first K-tile landed, end
of the K loop, stores done; wave 0 of every block).  argv: m:n:k[:res|geglu]"""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stable_diffusion_amd import engine as E, _lib
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
dev = "cuda"; H = E.act_dtype()
for spec in sys.argv[1:] or ["9216:960:320", "9216:320:320:res", "9216:2560:320:geglu", "9216:320:1280:res", "2304:1920:640", "576:1280:1280:res"]:
    f = spec.split(":"); m, n, k = int(f[0]), int(f[1]), int(f[2]); opts = f[3:]
    geglu = "geglu" in opts
    a = E.Act(torch.randn(1, m, k, device=dev).to(H), 1, 1, m, k)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(H)
    b = torch.zeros(n, device=dev)
    nout = n // 2 if geglu else n
    res = E.new_act(1, 1, m, nout, dev) if "res" in opts else None
    out = E.new_act(1, 1, m, nout, dev)
    for _ in range(3): E.gemm(a, w, n, bias=b, residual=res, geglu=geglu, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); E.gemm(a, w, n, bias=b, residual=res, geglu=geglu, out=out); e1.record(); torch.cuda.synchronize()
    name = lib.sdmi_last_kernel().decode()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    assert raw.sdmi_debug_igemm_stamps(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    tm = {"igemm_kernel<128, 64": (128, 64), "igemm_kernel<128, 128": (128, 128), "igemm_kernel<64, 64": (64, 64), "igemm_kernel<64, 128": (64, 128)}
    bm, bn = next(v for kk, v in tm.items() if name.startswith(kk))
    nb = ((m + bm - 1) // bm) * ((n + bn - 1) // bn)
    sx = buf[:min(nb, 8192), :].astype(np.int64) * 10
    s = sx[:, :5]
    t0 = s[:, 0].min()
    d = np.diff(s, axis=1)
    print("%-26s %-44s %4d blocks, launch %.1f us (event pair), span of the stamps %.1f us" % (spec, name, nb, e0.elapsed_time(e1) * 1e3, (s[:, 4].max() - t0) / 1e3))
    print("    per block, ns, mean (median): prologue %.0f (%.0f) | wait for the first K-tile %.0f (%.0f) | K loop %.0f (%.0f) | epilogue + stores %.0f (%.0f) | block %.0f" %
          (d[:, 0].mean(), np.median(d[:, 0]), d[:, 1].mean(), np.median(d[:, 1]), d[:, 2].mean(), np.median(d[:, 2]), d[:, 3].mean(), np.median(d[:, 3]), (s[:, 4] - s[:, 0]).mean()))
    if sx[:, 5].max() > 0:
        print("    prologue split, ns: start -> tile decode done %.0f | bias / residual prefetch issued %.0f | first DMAs issued %.0f | rest %.0f" % ((sx[:, 5] - sx[:, 0]).mean(), (sx[:, 6] - sx[:, 5]).mean(), (sx[:, 7] - sx[:, 6]).mean(), (sx[:, 1] - sx[:, 7]).mean()))
    st = np.sort(s[:, 0] - t0)
    print("    block start times, ns: 10%% %d  50%% %d  90%% %d  last %d;  end of the last block %d" % (st[len(st) // 10], st[len(st) // 2], st[len(st) * 9 // 10], st[-1], s[:, 4].max() - t0))
