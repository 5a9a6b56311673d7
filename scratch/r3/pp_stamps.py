"""Phase timing of attn64pp_kernel from the PPX=9 build (s_memrealtime stamps, 10 ns): per wave, per block: matrix issue,
wait+barrier, vector work, wait+barrier."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stable_diffusion_amd import engine as E, _lib
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
dev = "cuda"; H = E.act_dtype()
b, h, nq, nkv = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2:5:9216:9216").split(":"))
lib.sdmi_set_option(b"attn_pp", 1)
q = E.Act((torch.randn(b, nq, h * 64, device=dev) * E.Q_LOG2_SCALE).to(H), b, 1, nq, h * 64)
k = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
v = E.Act(torch.randn(b, nkv, h * 64, device=dev).to(H), b, 1, nkv, h * 64)
for _ in range(3): E.attention(q, k, v, h, log2q=True)
torch.cuda.synchronize()
buf = np.zeros((2, 8, 400), dtype=np.uint64)
assert raw.sdmi_debug_pp_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
names = ["MFMA issue", "stage", "wait+bar", "softmax", "wait+bar"]
NS = len(names)
for blk in range(2):
    print("block", blk)
    for w in range(8):
        s = buf[blk, w].astype(np.int64)
        n = int((s > 0).sum()) // NS
        s = s[:n * NS].reshape(-1, NS) * 10   # ns
        prev_end = np.concatenate([[s[0, 0]], s[:-1, NS - 1]])
        d = np.concatenate([(s[:, 0] - prev_end)[:, None], np.diff(s, axis=1)], axis=1)
        print("  wave %d (%d blocks) ns per block: " % (w, n) + "  ".join("%s %.0f" % (names[j], d[4:70, j].mean()) for j in range(NS)) + "   total %.0f" % d[4:70].sum(axis=1).mean())
