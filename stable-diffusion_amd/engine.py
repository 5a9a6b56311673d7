"""Host-side engine: NHWC-bf16 activation views, weight pre-pack cache and thin wrappers over the C-ABI.

PyTorch is used for device memory (torch.empty), the current HIP stream and parameter storage only; all
arithmetic runs in libsdmi355.so.  There is no fallback: a missing library raises (see _lib.load).
"""
import ctypes
import math
import os
import weakref

import torch

from . import _lib as L



def act_dtype():
    """torch dtype of the library's 16-bit storage type (fp16 unless built with -DSDMI_BF16)."""
    return torch.bfloat16 if L.load().sdmi_act_dtype() == 1 else torch.float16



def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(t, what):
    if not t.is_cuda:
        raise L.SdmiError("%s: tensor is on %s; the HIP path needs a cuda (ROCm) tensor and has no CPU fallback"
                          % (what, t.device))


class Act(object):
    """(n, h, w, c) activation stored NHWC bf16: buf[n, h*w, ld_total], columns [off, off+c).

    stats: {column offset: (colstats tensor, row parts per sample, channels)} left by the kernels that produced column
    ranges of the buffer (shared by every view of it): per-channel sums for the GroupNorm that follows.
    rstats: (rowstats tensor, column parts) left by the GEMM that produced this view: per-row sums for a LayerNorm."""
    __slots__ = ("buf", "n", "h", "w", "c", "off", "stats", "rstats", "gn", "normed")

    def __init__(self, buf, n, h, w, c, off=0, stats=None):
        self.buf, self.n, self.h, self.w, self.c, self.off = buf, n, h, w, c, off
        self.stats = {} if stats is None else stats
        self.rstats = None
        self.gn = None      # pending GroupNorm + SiLU: the (n, c, 2) scale/shift table the consuming conv applies itself
        self.normed = None  # (key, Act): GroupNorm(+SiLU) of this tensor already written by its producer (conv3x3(gn_next=...))

    @property
    def hw(self):
        return self.h * self.w

    @property
    def ld(self):
        return self.c if self.buf is None else self.buf.shape[-1]

    @property
    def ptr(self):
        if self.buf is None:
            raise L.SdmiError("this activation was never materialised: its producer wrote only the GroupNorm that follows it "
                              "(conv3x3(..., gn_next=..., keep_raw=False))")
        return self.buf.data_ptr() + 2 * self.off

    @property
    def device(self):
        return self.normed[1].device if self.buf is None else self.buf.device

    def cols(self, off, c):
        return Act(self.buf, self.n, self.h, self.w, c, self.off + off, self.stats)


def new_act(n, h, w, c, device):
    return Act(torch.empty((n, h * w, c), dtype=act_dtype(), device=device), n, h, w, c)


def from_nchw(x):
    """NCHW fp32 tensor -> Act (module boundary, SURVEY 8(b) tensor conventions)."""
    _need_cuda(x, "from_nchw")
    x = x.contiguous().float()
    n, c, h, w = x.shape
    cpad = (c + 7) // 8 * 8
    a = new_act(n, h, w, cpad, x.device)
    if cpad != c:
        a.buf.zero_()
    L.call("sdmi_nchw_f32_to_nhwc_h16", x.data_ptr(), a.ptr, n, c, h * w, a.ld, _stream())
    a.c = c
    return a


def to_nchw(a):
    out = torch.empty((a.n, a.c, a.h, a.w), dtype=torch.float32, device=a.device)
    L.call("sdmi_nhwc_h16_to_nchw_f32", a.ptr, out.data_ptr(), a.n, a.c, a.hw, a.ld, _stream())
    return out


def tokens_from_f32(x):
    """(B, T, C) fp32 -> Act(n=B, h=1, w=T, c=C) bf16 (text context entering to_k / to_v)."""
    _need_cuda(x, "tokens_from_f32")
    x = x.contiguous().float()
    b, t, c = x.shape
    a = new_act(b, 1, t, c, x.device)
    L.call("sdmi_f32_to_h16_rows", x.data_ptr(), a.ptr, b * t, c, a.ld, _stream())
    return a


# --------------------------------------------------------------------------- weight pre-pack cache
class _PackCache(object):
    """Packed bf16 copies of parameters, rebuilt when a parameter's storage or version changes
    (load_state_dict / .to(device)); parameter names and shapes are never touched (checkpoint row)."""

    def __init__(self):
        self._d = {}

    def get(self, params, tag, build):
        key = (tag,) + tuple(id(p) for p in params)
        sig = tuple((p.data_ptr(), p._version, p.device) for p in params)
        hit = self._d.get(key)
        if hit is not None and hit[0] == sig and all(r() is p for r, p in zip(hit[1], params)):
            return hit[2]
        val = build()
        self._d[key] = (sig, [weakref.ref(p) for p in params], val)
        return val

    def clear(self):
        self._d.clear()


PACK = _PackCache()


def _f32c(p, what):
    _need_cuda(p, what)
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise L.SdmiError("%s: expected a contiguous fp32 parameter" % what)
    return p


def packed_conv3x3(weight):
    def build():
        w = _f32c(weight, "conv weight")
        co, ci = w.shape[0], w.shape[1]
        out = torch.empty((co, 9 * ci), dtype=act_dtype(), device=w.device)
        L.call("sdmi_pack_conv3x3", w.data_ptr(), out.data_ptr(), co, ci, _stream())
        return out
    return PACK.get((weight,), "c3", build)


# softmax(q k^T / 8) in base 2: the factor the query projection carries when attention runs on sdmi_attention_log2q
Q_LOG2_SCALE = 0.125 * 1.4426950408889634


def _scaled(w, scale):
    """fp32 weight times a per-projection constant, before the single rounding to 16 bits (pack time only)."""
    return w if scale is None or scale == 1.0 else (w * scale).contiguous()


def packed_linear(weights, geglu=False, scales=None):
    """One or more (N_i, K[,1,1]) fp32 weights stacked along N -> bf16 [sum N_i][K].  scales: one constant per weight,
    multiplied in before rounding (the query projection's scale * log2(e))."""
    weights = tuple(weights)
    scales = tuple(scales) if scales is not None else (None,) * len(weights)

    def build():
        k = weights[0].shape[1]
        ntot = sum(w.shape[0] for w in weights)
        out = torch.empty((ntot, k), dtype=act_dtype(), device=weights[0].device)
        r = 0
        for w, sc in zip(weights, scales):
            ws = _scaled(_f32c(w, "linear weight"), sc)
            L.call("sdmi_pack_linear", ws.data_ptr(), out.data_ptr() + 2 * r * k, w.shape[0], k, 1 if geglu else 0, _stream())
            r += w.shape[0]
        return out
    return PACK.get(weights, ("lg" if geglu else "l") + repr(scales if any(scales) else ""), build)


def stacked_f32(params, scales=None):
    """Concatenate 1-D fp32 parameters (stacked biases of fused projections); scales: one constant per parameter."""
    params = tuple(params)
    scales = tuple(scales) if scales is not None else (None,) * len(params)
    return PACK.get(params, "cat" + repr(scales if any(scales) else ""),
                    lambda: torch.cat([_scaled(_f32c(p, "bias"), sc) for p, sc in zip(params, scales)]).contiguous())


# --------------------------------------------------------------------------- split-K workspace
_WS = {}
_WS_RETIRED = []
_WS_WARNED = []


def _workspace(device, nbytes):
    """Per-(device, stream) fp32 scratch for split-K partials, grown on demand (kernels on one stream run in order, so one
    buffer serves every launch of that stream).  Returns (ptr, bytes); (0, 0) when nothing is needed."""
    if nbytes <= 0:
        return 0, 0
    key = (device, torch.cuda.current_stream(device).cuda_stream)   # concurrent streams must not share partials
    t = _WS.get(key)
    if t is None or t.numel() * 4 < nbytes:
        if t is not None:
            _WS_RETIRED.append(t)       # a captured HIP graph may still address it: never freed (sizes double, so the
            nbytes = max(nbytes, 2 * t.numel() * 4)   # retired buffers add up to less than the live one)
        t = torch.empty(((max(nbytes, 64 << 20) + 3) // 4,), dtype=torch.float32, device=device)
        if WS_FLAGS:
            # opt-in in-launch split-K reduction: a bank of arrival counters per workspace (16 per device).  A retired
            # workspace keeps its bank (a captured graph may still run launches that count in it); when no bank is free this
            # workspace keeps the separate reduce kernel, said once
            import ctypes as C
            if L.load().sdmi_ws_tickets_init(C.c_void_p(t.data_ptr()), C.c_void_p(_stream())) != 0 and not _WS_WARNED:
                _WS_WARNED.append(True)
                import warnings
                warnings.warn("SDMI_SPLITK_INKERNEL=1: %s; this workspace runs the split-K reduce kernel" %
                              L.load().sdmi_last_error().decode())
        _WS[key] = t
    return t.data_ptr(), t.numel() * 4


# SDMI_SPLITK_INKERNEL=1: split-K launches finish inside the launch (SDMI_F_WS_TICKETS; the workspace above is private to a
# stream, as that needs) instead of by the separate reduce kernel.  Off: bit-identical but measured slower (UNet step 8.82
# against 8.28 ms, single convs +4 ... 15 us: profiles/r02_splitk_tickets.txt)
WS_FLAGS = L.F_WS_TICKETS if os.environ.get("SDMI_SPLITK_INKERNEL", "0") == "1" else 0


def _check_out(res, n, h, w, c, what):
    """A caller-supplied `out=` view must have exactly the shape the op produces: the kernels write n*h*w rows of c
    channels into it unchecked (the reference raises at torch.cat for the same mismatch, Unet.py:132)."""
    if (res.n, res.h, res.w, res.c) != (n, h, w, c):
        raise L.SdmiError("%s: `out` is (n=%d, h=%d, w=%d, c=%d) but the result is (n=%d, h=%d, w=%d, c=%d)" %
                          (what, res.n, res.h, res.w, res.c, n, h, w, c))


# --------------------------------------------------------------------------- range audit (debug aid)
# The fp16 build saturates on store: a checkpoint whose activations pass +-65504 somewhere (VAE up-blocks are the usual
# place) is clipped there, silently.  With RANGE_AUDIT = True every conv / GEMM result is scanned by sdmi_count_saturated and
# RANGE_LOG collects (op, rows, channels, clipped elements) for the layers that clip: run one image that way when a new
# checkpoint is loaded; any entry means "use the bf16 build" (-DSDMI_BF16).  Off in production (one extra launch per op;
# not capturable: it reads the counter back).
RANGE_AUDIT = False
RANGE_LOG = []


def _audit(what, res):
    if not RANGE_AUDIT or not isinstance(res, Act) or act_dtype() != torch.float16:
        return
    if res.buf is None:                 # only the GroupNorm of the result was written (conv3x3 gn_next, keep_raw=False)
        res = res.normed[1]
    cnt = torch.zeros((1,), dtype=torch.int64, device=res.device)
    L.call("sdmi_count_saturated", res.ptr, res.ld, res.n * res.hw, res.c, cnt.data_ptr(), _stream())
    n = int(cnt.item())
    if n:
        RANGE_LOG.append((what, res.n * res.hw, res.c, n))


# --------------------------------------------------------------------------- ops
USE_COLSTATS = os.environ.get("SDMI_COLSTATS", "1") == "1"     # GroupNorm statistics from the producer's epilogue (A/B switch)
USE_LNFOLD = os.environ.get("SDMI_LNFOLD", "1") == "1"         # LayerNorm folded into the GEMM that follows it (A/B switch)
# GroupNorm + SiLU applied by the consuming conv to its staged input (north-star item (b)); only convs whose one block column
# covers every output channel qualify (the VAE's 128-channel level).  Default from the measurement in DESIGN.md section 5.
CONV_IN_COLSTATS = os.environ.get("SDMI_CONV_IN_COLSTATS", "1") == "1"   # the UNet's first conv leaves GroupNorm statistics too (A/B switch)
GN_IN_CONV = os.environ.get("SDMI_GN_IN_CONV", "0") == "1"
# the split-K reduce of a conv also writes the GroupNorm that follows it (24x24 / 12x12 levels).  Off: bit-identical and measured
# SLOWER (captured step 7.81 -> 7.86 ms; per pair 40 -> 45 us at 24x24, 28.6 -> 29.8 at 12x12: one block per (sample, group) pulls
# every split's slab of partials through 32 CUs, where the plain reduce kernel uses the whole chip; profiles/r03_gn_out_ab.txt)
GN_OUT_FUSE = os.environ.get("SDMI_GN_OUT_FUSE", "0") == "1"


def _gn_small(hw, c, groups):
    """sdmi_groupnorm's rule for its fused single-launch kernel (csrc/norm.hip): such tensors gain nothing from colstats."""
    cpg = c // groups
    dwords = hw * (cpg // 2)
    return dwords <= 12 * 1024 or (dwords <= 24 * 1024 and cpg >= 40)


def _stat_segments(a):
    """The colstats entries that tile columns [a.off, a.off + a.c) of a's buffer exactly, in order; None if they do not."""
    segs, pos = [], a.off
    while pos < a.off + a.c:
        e = a.stats.get(pos)
        if e is None or len(segs) == 2:
            return None
        segs.append(e)
        pos += e[2]
    return segs if pos == a.off + a.c else None


def _gn_key(weight, bias, eps, silu, groups):
    return (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, float(eps), bool(silu), int(groups))


def groupnorm(a, weight, bias, eps, silu, groups=32, conv_cout=None):
    """GroupNorm(+SiLU): one fused launch when a (sample, group) slice fits a block; else, when the kernels that produced
    `a` left per-channel statistics (colstats), a fold of those + apply; else stats + apply (csrc/norm.hip).
    conv_cout: the result feeds ONLY a stride-1 pad-1 3x3 conv with that many output channels; where that conv can apply the
    normalisation to its own staged input (sdmi_epilogue_ext.gn_in, GN_IN_CONV) the apply pass is skipped: the returned view
    is the RAW tensor carrying the scale/shift table (Act.gn), which conv3x3 hands to the kernel."""
    if a.normed is not None:
        key, y = a.normed
        if key == _gn_key(weight, bias, eps, silu, groups):
            return y
    dev = a.device
    small = _gn_small(a.hw, a.c, groups)
    segs = _stat_segments(a) if (USE_COLSTATS and not small) else None
    defer = (GN_IN_CONV and conv_cout is not None and silu and not small and
             L.load().sdmi_conv3x3_gn_in_ok(a.n, a.h, a.w, a.c, conv_cout, a.ld) == 1)
    if segs is not None:
        import ctypes as C
        ns = len(segs)
        ss = torch.empty((L.load().sdmi_groupnorm_colstats_ss_bytes(a.n, a.c, groups) // 4,), dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * 2)(*[t.data_ptr() for t, _, _ in segs] + [0] * (2 - ns))
        ivec = lambda v: (C.c_int * 2)(*(list(v) + [0] * (2 - ns)))
        c0, acc = [], 0
        for _, _, cc in segs:
            c0.append(acc)
            acc += cc
        seg_args = (ns, C.cast(ptrs, C.c_void_p), C.cast(ivec(p for _, p, _ in segs), C.c_void_p), C.cast(ivec(cc for _, _, cc in segs), C.c_void_p),
                    C.cast(ivec([0] * ns), C.c_void_p), C.cast(ivec(c0), C.c_void_p), C.cast(ivec(cc for _, _, cc in segs), C.c_void_p),
                    ss.data_ptr(), _stream())
        gb = (_f32c(weight, "gn weight").data_ptr(), _f32c(bias, "gn bias").data_ptr(), float(eps))
        if defer:
            L.call("sdmi_groupnorm_colstats_finalize", a.n, a.hw, a.c, groups, *(gb + seg_args))
            y = Act(a.buf, a.n, a.h, a.w, a.c, a.off, a.stats)
            y.gn = ss
            return y
        y = new_act(a.n, a.h, a.w, a.c, dev)
        L.call("sdmi_groupnorm_from_colstats", a.ptr, a.ld, y.ptr, y.ld, a.n, a.hw, a.c, groups, *(gb + (L.F_SILU if silu else 0,) + seg_args))
        return y
    ss = torch.empty((a.n, a.c, 2), dtype=torch.float32, device=dev)
    ws = torch.empty((L.load().sdmi_groupnorm_ws_bytes(a.n, a.hw, a.c) // 4,), dtype=torch.float32, device=dev)
    if defer:
        L.call("sdmi_groupnorm_stats", a.ptr, a.n, a.hw, a.c, a.ld, groups, _f32c(weight, "gn weight").data_ptr(),
               _f32c(bias, "gn bias").data_ptr(), float(eps), ss.data_ptr(), ws.data_ptr(), _stream())
        y = Act(a.buf, a.n, a.h, a.w, a.c, a.off, a.stats)
        y.gn = ss
        return y
    y = new_act(a.n, a.h, a.w, a.c, dev)
    L.call("sdmi_groupnorm", a.ptr, a.ld, y.ptr, y.ld, a.n, a.hw, a.c, groups, _f32c(weight, "gn weight").data_ptr(),
           _f32c(bias, "gn bias").data_ptr(), float(eps), L.F_SILU if silu else 0, ss.data_ptr(), ws.data_ptr(), _stream())
    return y


def layernorm(a, weight, bias, eps=1e-5):
    y = new_act(a.n, a.h, a.w, a.c, a.device)
    L.call("sdmi_layernorm", a.ptr, a.ld, _f32c(weight, "ln weight").data_ptr(), _f32c(bias, "ln bias").data_ptr(),
           float(eps), y.ptr, y.ld, a.n * a.hw, a.c, _stream())
    return y


def conv3x3(a, weight, bias, stride=1, pad=(1, 1, 1, 1), upsample=False, rowbias=None, ldrb=0, residual=None,
            out=None, out_nchw=False, out_u8_image=False, gn_next=None, keep_raw=True):
    """3x3 conv. pad = (left, right, top, bottom) as F.pad takes it. rowbias: fp32 tensor/ptr (per-sample channel add).
    gn_next = (weight, bias, eps, silu[, groups]) of the GroupNorm that consumes the result: where the launch splits K and the
    result is small (the 24x24 / 12x12 levels) the split-K reduce writes that GroupNorm too (sdmi_epilogue_ext.gn_out, GN_OUT_FUSE)
    and the returned Act carries it (Act.normed; engine.groupnorm returns it).  keep_raw=False: nobody but that GroupNorm
    reads the result, so the raw tensor is not written at all."""
    wp = packed_conv3x3(weight)
    cout, cin = weight.shape[0], weight.shape[1]
    if cin != a.c:
        raise L.SdmiError("conv3x3: input has %d channels, weight expects %d" % (a.c, cin))
    pl, pr, pt, pb = pad
    gno = None
    if a.gn is not None and (stride != 1 or upsample or tuple(pad) != (1, 1, 1, 1)):
        raise L.SdmiError("conv3x3: the input carries a pending GroupNorm (Act.gn) that only a stride-1 pad-1 conv can apply")
    hin, win = (a.h * 2, a.w * 2) if upsample else (a.h, a.w)
    ho = (hin + pt + pb - 3) // stride + 1
    wo = (win + pl + pr - 3) // stride + 1
    flags = 0
    if bias is not None:
        flags |= L.F_BIAS
    if upsample:
        flags |= L.F_UPSAMPLE2X
    rb_ptr = 0
    if rowbias is not None:
        flags |= L.F_ROWBIAS
        rb_ptr = rowbias if isinstance(rowbias, int) else rowbias.data_ptr()
    res_ptr, ldr = 0, 0
    if residual is not None:
        flags |= L.F_RESIDUAL
        res_ptr, ldr = residual.ptr, residual.ld
    if out_u8_image:
        # uint8 (n, ho, wo, cout) image: clamp((y + 1) / 2, 0, 1) * 255 in the conv epilogue
        flags |= L.F_OUT_U8_IMAGE
        res = torch.empty((a.n, ho, wo, cout), dtype=torch.uint8, device=a.device)
        optr, ldc = res.data_ptr(), 0
    elif out_nchw is not False and out_nchw is not None:
        flags |= L.F_OUT_NCHW_F32
        if out_nchw is True:
            res = torch.empty((a.n, cout, ho, wo), dtype=torch.float32, device=a.device)
        else:                           # a caller's fp32 NCHW tensor (e.g. one half of a CFG eps buffer)
            res = out_nchw
            if tuple(res.shape) != (a.n, cout, ho, wo) or res.dtype != torch.float32 or not res.is_contiguous():
                raise L.SdmiError("conv3x3: out_nchw tensor must be contiguous fp32 %s, got %s %s" %
                                  ((a.n, cout, ho, wo), tuple(res.shape), res.dtype))
        optr, ldc = res.data_ptr(), 0
    else:
        gno = None
        if gn_next is not None and GN_OUT_FUSE:
            groups = gn_next[4] if len(gn_next) > 4 else 32
            if L.load().sdmi_conv3x3_gn_out_ok(a.n, a.h, a.w, cin, cout, ho, wo, stride, pt, pl, flags, groups) == 1:
                gy = new_act(a.n, ho, wo, cout, a.device)
                gno = L.GnOut(y=gy.ptr, ldy=gy.ld, gamma=_f32c(gn_next[0], "gn weight").data_ptr(), beta=_f32c(gn_next[1], "gn bias").data_ptr(),
                              eps=float(gn_next[2]), groups=groups, flags=L.F_SILU if gn_next[3] else 0, keep_raw=1 if (keep_raw or out is not None) else 0)
        if gno is not None and not gno.keep_raw:
            res = Act(None, a.n, ho, wo, cout)
            optr, ldc = 0, cout
        else:
            res = out if out is not None else new_act(a.n, ho, wo, cout, a.device)
            _check_out(res, a.n, ho, wo, cout, "conv3x3")
            optr, ldc = res.ptr, res.ld
    ws, wsb = _workspace(a.device, L.load().sdmi_conv3x3_ws_bytes(a.n, cin, cout, ho, wo))
    ext = None
    if isinstance(res, Act):
        res.normed = None               # a caller's `out=` view may be reused: whatever an earlier producer left is stale now
    if isinstance(res, Act) and gno is not None:
        res.stats.pop(res.off, None)
        res.normed = (_gn_key(gn_next[0], gn_next[1], gn_next[2], gn_next[3], groups), gy)
        ext = L.EpilogueExt(gn_out=ctypes.pointer(gno))
    elif USE_COLSTATS and isinstance(res, Act) and ho * wo >= 2304:
        # leave per-channel statistics for the GroupNorm that follows (large tensors only: small ones use the fused kernel)
        rp = ctypes.c_int(0)
        L.call("sdmi_conv3x3_stats_parts", a.n, a.h, a.w, cin, cout, ho, wo, stride, pt, pl, flags, ctypes.addressof(rp))
        if rp.value > 0:
            cs = torch.empty((rp.value, cout, 2), dtype=torch.float32, device=a.device)
            ext = L.EpilogueExt(colstats=cs.data_ptr())
            res.stats[res.off] = (cs, rp.value // a.n, cout)
    if ext is None and isinstance(res, Act):
        res.stats.pop(res.off, None)
    if a.gn is not None:
        flags |= L.F_SILU
        ext = ext or L.EpilogueExt()
        ext.gn_in = a.gn.data_ptr()
    L.call("sdmi_conv3x3_ex", a.ptr, a.ld, a.n, a.h, a.w, cin, wp.data_ptr(), cout, ho, wo, stride, pt, pl,
           _ptr(bias), rb_ptr, ldrb, res_ptr, ldr, optr, ldc, flags | (WS_FLAGS if ws else 0), ws, wsb, _stream(),
           ctypes.addressof(ext) if ext is not None else 0)
    _audit("conv3x3 %d->%d @%dx%d" % (cin, cout, ho, wo), res)
    return res


def gemm(a, wp, n_out, bias=None, residual=None, geglu=False, out=None, out_shape=None, colstats=False, rowstats=False, ln=None):
    """Token GEMM: rows = n*hw of `a`, W packed [n_out][a.c]. Returns an Act with c = n_out (or n_out/2 for GEGLU).
    colstats / rowstats: also leave per-channel / per-row statistics for a GroupNorm / LayerNorm that follows (where the
    problem can: res.stats / res.rstats tell).  ln = (u, eps): LayerNorm folded in (wp and bias from packed_linear_ln;
    `a` must carry rstats)."""
    m = a.n * a.hw
    cres = n_out // 2 if geglu else n_out
    res = out if out is not None else new_act(a.n, a.h, a.w, cres, a.device)
    if (res.n * res.hw, res.c) != (m, cres):
        raise L.SdmiError("gemm: `out` holds %d rows x %d channels, the product has %d x %d" % (res.n * res.hw, res.c, m, cres))
    flags = 0
    if bias is not None:
        flags |= L.F_BIAS
    if geglu:
        flags |= L.F_GEGLU
    res_ptr, ldr = 0, 0
    if residual is not None:
        flags |= L.F_RESIDUAL
        res_ptr, ldr = residual.ptr, residual.ld
    ext = None
    if ln is not None:
        u, eps = ln
        rst, parts = a.rstats
        flags |= L.F_LNFOLD
        ext = L.EpilogueExt(ln_rowstats=rst.data_ptr(), ln_u=u.data_ptr(), ln_parts=parts, ln_eps=float(eps))
    res.rstats = None
    res.normed = None
    keep_cs = False
    if (colstats and USE_COLSTATS and a.hw >= 2304) or rowstats:
        rp, cp = ctypes.c_int(0), ctypes.c_int(0)
        L.call("sdmi_gemm_stats_parts", m, n_out, a.c, 1, a.hw, flags, ctypes.addressof(rp), ctypes.addressof(cp))
        if colstats and USE_COLSTATS and a.hw >= 2304 and rp.value > 0:
            cs = torch.empty((rp.value, n_out, 2), dtype=torch.float32, device=a.device)
            ext = ext or L.EpilogueExt()
            ext.colstats = cs.data_ptr()
            res.stats[res.off] = (cs, rp.value // a.n, n_out)
            keep_cs = True
        if rowstats and cp.value > 0:
            rs = torch.empty((m, cp.value, 2), dtype=torch.float32, device=a.device)
            ext = ext or L.EpilogueExt()
            ext.rowstats = rs.data_ptr()
            res.rstats = (rs, cp.value)
    if not keep_cs:
        res.stats.pop(res.off, None)
    ws, wsb = _workspace(a.device, L.load().sdmi_gemm_ws_bytes(m, n_out, a.c, 1, a.hw, flags))
    L.call("sdmi_gemm_ex", a.ptr, a.ld, 0, wp.data_ptr(), wp.shape[1], 0, res.ptr, res.ld, 0, m, n_out, a.c, 1,
           _ptr(bias), 0, 0, a.hw, res_ptr, ldr, 1.0, flags | (WS_FLAGS if ws else 0), ws, wsb, _stream(),
           ctypes.addressof(ext) if ext is not None else 0)
    _audit("gemm %dx%dx%d" % (m, n_out, a.c), res)
    return res


def packed_linear_ln(weights, ln_weight, ln_bias, bias=None, geglu=False, scales=None):
    """Weights of the Linear(s) that follow a LayerNorm, with the LayerNorm folded in (sdmi_pack_linear_ln): returns
    (packed h16 [sum N_i][K] with gamma folded, u fp32 [sum N_i] in packed row order, v fp32 [sum N_i] = beta.W + bias)."""
    weights = tuple(weights)
    biases = tuple(bias) if isinstance(bias, (tuple, list)) else ((bias,) if bias is not None else ())
    if geglu and len(weights) != 1:
        raise L.SdmiError("packed_linear_ln: GEGLU packs one weight")
    scales = tuple(scales) if scales is not None else (None,) * len(weights)
    if biases and any(scales):
        raise L.SdmiError("packed_linear_ln: scaled projections carry no bias")

    def build():
        k = weights[0].shape[1]
        ntot = sum(w.shape[0] for w in weights)
        dev = weights[0].device
        wp = torch.empty((ntot, k), dtype=act_dtype(), device=dev)
        u = torch.empty((ntot,), dtype=torch.float32, device=dev)
        v = torch.empty((ntot,), dtype=torch.float32, device=dev)
        g, b = _f32c(ln_weight, "ln weight"), _f32c(ln_bias, "ln bias")
        r = 0
        for i, w in enumerate(weights):
            ws = _scaled(_f32c(w, "linear weight"), scales[i])
            bi = _f32c(biases[i], "linear bias").data_ptr() if biases else 0
            L.call("sdmi_pack_linear_ln", ws.data_ptr(), g.data_ptr(), b.data_ptr(), bi, wp.data_ptr() + 2 * r * k,
                   u.data_ptr() + 4 * r, v.data_ptr() + 4 * r, w.shape[0], k, 1 if geglu else 0, _stream())
            r += w.shape[0]
        return wp, u, v
    return PACK.get(weights + (ln_weight, ln_bias) + biases,
                    "lnl" + ("g" if geglu else "") + repr(scales if any(scales) else ""), build)


def gemm_ln(a, ln_mod, weights, bias=None, geglu=False, scales=None, **kw):
    """LayerNorm(a) @ W^T (+bias): folded into one GEMM when `a` carries row statistics from its producer (gemm(...,
    rowstats=True)), else the LayerNorm kernel followed by the plain GEMM."""
    weights = tuple(weights)
    n_out = sum(w.shape[0] for w in weights)
    if USE_LNFOLD and a.rstats is not None:
        wp, u, v = packed_linear_ln(weights, ln_mod.weight, ln_mod.bias, bias, geglu, scales)
        return gemm(a, wp, n_out, bias=v, geglu=geglu, ln=(u, ln_mod.eps), **kw)
    x = layernorm(a, ln_mod.weight, ln_mod.bias, ln_mod.eps)
    b = stacked_f32(tuple(bias)) if isinstance(bias, (tuple, list)) else bias
    return gemm(x, packed_linear(weights, geglu=geglu, scales=scales), n_out, bias=b, geglu=geglu, **kw)


def attention(q, k, v, heads, out=None, log2q=False):
    """softmax(q k^T / 8) v per head; q/k/v are column views (Act.cols) of token buffers.  log2q: q already carries
    Q_LOG2_SCALE (folded into the to_q weights at pack time)."""
    o = out if out is not None else new_act(q.n, q.h, q.w, heads * 64, q.device)
    if (o.n, o.hw, o.c) != (q.n, q.hw, heads * 64) or (k.n, v.n, v.hw) != (q.n, q.n, k.hw):
        raise L.SdmiError("attention: q (%d,%d) / k (%d,%d) / v (%d,%d) / out (%d,%d,%d) do not match" %
                          (q.n, q.hw, k.n, k.hw, v.n, v.hw, o.n, o.hw, o.c))
    args = (q.ptr, q.ld, q.hw * q.ld, k.ptr, k.ld, k.hw * k.ld, v.ptr, v.ld, v.hw * v.ld,
            o.ptr, o.ld, o.hw * o.ld, q.n, heads, q.hw, k.hw)
    if log2q:
        L.call("sdmi_attention_log2q", *(args + (_stream(),)))
    else:
        L.call("sdmi_attention", *(args + (0.125, _stream())))
    return o


# cross-attention with its LayerNorm + to_q projection inside the attention kernel (sdmi_cross_attention_lnq): one launch
# less per cross-attention.  XATTN_FUSE_Q_MAX_C: widest token stream it is used for (the kernel stages the head's weights in the
# LDS whole: 320 or 640 channels); 0 = never
XATTN_FUSE_Q_MAX_C = int(os.environ.get("SDMI_XATTN_FUSE_Q_MAX_C", "0"))


def cross_attention_lnq(a, ln_mod, to_q_weight, k, v, heads, out=None):
    """softmax(to_q(LayerNorm(a)) k^T / 8) v per head in ONE launch; None when this call cannot take the fused kernel (the
    caller then runs gemm_ln + attention).  `a` must carry row statistics (gemm(..., rowstats=True))."""
    if not (USE_LNFOLD and a.rstats is not None and a.c in (320, 640) and a.c <= XATTN_FUSE_Q_MAX_C):
        return None
    wp, u, vv = packed_linear_ln((to_q_weight,), ln_mod.weight, ln_mod.bias, None, False, (Q_LOG2_SCALE,))
    rst, parts = a.rstats
    o = out if out is not None else new_act(a.n, a.h, a.w, heads * 64, a.device)
    if (o.n, o.hw, o.c) != (a.n, a.hw, heads * 64) or (k.n, v.n, v.hw) != (a.n, a.n, k.hw) or wp.shape[0] != heads * 64:
        raise L.SdmiError("cross_attention_lnq: shapes do not match")
    L.call("sdmi_cross_attention_lnq", a.ptr, a.ld, a.hw * a.ld, a.c, wp.data_ptr(), wp.shape[1], u.data_ptr(), vv.data_ptr(),
           rst.data_ptr(), parts, float(ln_mod.eps), k.ptr, k.ld, k.hw * k.ld, v.ptr, v.ld, v.hw * v.ld, o.ptr, o.ld, o.hw * o.ld,
           a.n, heads, a.hw, k.hw, _stream())
    return o


def linear_small_m(x, wp, bias, silu_in=False):
    """fp32 (M<=64, K) x bf16 [N][K] -> fp32 (M, N)."""
    x = x.contiguous()
    m, k = x.shape
    n = wp.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    for m0 in range(0, m, 64):      # the kernel takes at most 64 rows per launch (larger CFG batches: several launches)
        L.call("sdmi_linear_small_m", x.data_ptr() + 4 * m0 * k, wp.data_ptr(), _ptr(bias), out.data_ptr() + 4 * m0 * n,
               min(64, m - m0), n, k, L.F_SILU if silu_in else 0, _stream())
    return out


_FREQS = {}


def timestep_embedding(t, dim, max_period=10000):
    """Unet.py:7-25 on device; the frequency table is the host fp32 expression of the reference."""
    _need_cuda(t, "timestep_embedding")
    half = dim // 2
    key = (t.device, half, max_period)
    if key not in _FREQS:
        # exp(-ln(max_period) i/half) evaluated in fp64 and rounded once: the same table on every host (torch's
        # fp32 exp differs in the last ulp between CPU back ends, which moves cos(t*f) by up to ~6e-5 at t ~ 1000)
        f = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float64) / half).float()
        _FREQS[key] = f.to(t.device)
    t = t.to(torch.int64).contiguous()
    out = torch.zeros((t.shape[0], dim), dtype=torch.float32, device=t.device) if dim % 2 else \
        torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    if dim % 2:
        raise L.SdmiError("timestep_embedding: odd dim unsupported")
    L.call("sdmi_timestep_embedding", t.data_ptr(), _FREQS[key].data_ptr(), out.data_ptr(), t.shape[0], half, _stream())
    return out


def attention_d512(q, k, v, out=None, log2q=False):
    """softmax(q k^T / sqrt(512)) v, one head of dim 512, flash-style (no N x N buffer): q/k/v are column views of token
    buffers.  log2q: q already carries 512^-0.5 * log2(e)."""
    if (q.c, k.c, v.c) != (512, 512, 512) or (k.n, v.n, v.hw) != (q.n, q.n, k.hw):
        raise L.SdmiError("attention_d512: q (%d,%d,%d) / k (%d,%d,%d) / v (%d,%d,%d)" % (q.n, q.hw, q.c, k.n, k.hw, k.c, v.n, v.hw, v.c))
    o = out if out is not None else new_act(q.n, q.h, q.w, 512, q.device)
    _check_out(o, q.n, q.h, q.w, 512, "attention_d512")
    if q.hw != k.hw:
        raise L.SdmiError("attention_d512: self-attention only (%d queries, %d keys)" % (q.hw, k.hw))
    args = (q.ptr, q.ld, q.hw * q.ld, k.ptr, k.ld, k.hw * k.ld, v.ptr, v.ld, v.hw * v.ld, o.ptr, o.ld, o.hw * o.ld, q.n, q.hw)
    if log2q:
        L.call("sdmi_attention_d512_log2q", *(args + (_stream(),)))
    else:
        L.call("sdmi_attention_d512", *(args + (512.0 ** -0.5, _stream())))
    return o


def cfg_ddim_step(eps, x, noise, guidance, s1m, a_t, a_prev, sigma, temperature=1.0, x_out=None, pred_x0=None,
                  xc_out=None):
    """Fused CFG + DDIM update (LatentDiffusionInpainting.py:198-212). eps: (2b,4,h,w) fp32 [uncond|cond]."""
    b = x.shape[0]
    chw = x[0].numel()
    if x_out is None:
        x_out = torch.empty_like(x)
    L.call("sdmi_cfg_ddim_step", eps.data_ptr(), x.data_ptr(), noise.data_ptr(), float(guidance), float(s1m), float(a_t),
           float(a_prev), float(sigma), float(temperature), x_out.data_ptr(), _ptr(pred_x0), _ptr(xc_out),
           0 if xc_out is None else xc_out.shape[1], b, chw, _stream())
    return x_out
