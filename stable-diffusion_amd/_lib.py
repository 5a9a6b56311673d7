"""ctypes binding of include/sdmi.h (libsdmi355.so).

The library is the product path: if it is missing this module raises at first use - there is no
PyTorch / CPU fallback (a silent fallback would void every parity claim).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDMI_LIB_PATH") or os.path.join(_HERE, "libsdmi355.so")   # env: A/B another build of the library

# flags (sdmi.h)
F_BIAS, F_ROWBIAS, F_RESIDUAL, F_GEGLU, F_OUT_NCHW_F32, F_UPSAMPLE2X, F_SILU, F_OUT_F32, F_OUT_U8_IMAGE = 1, 2, 4, 8, 16, 32, 64, 128, 256
F_LNFOLD = 512
F_WS_TICKETS = 1024


class GnOut(C.Structure):
    """sdmi_gn_out (include/sdmi.h)."""
    _fields_ = [("y", C.c_void_p), ("ldy", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("groups", C.c_int),
                ("flags", C.c_uint32), ("keep_raw", C.c_int)]


class EpilogueExt(C.Structure):
    """sdmi_epilogue_ext (include/sdmi.h)."""
    _fields_ = [("colstats", C.c_void_p), ("rowstats", C.c_void_p), ("ln_rowstats", C.c_void_p), ("ln_u", C.c_void_p),
                ("ln_parts", C.c_int), ("ln_eps", C.c_float), ("gn_in", C.c_void_p), ("gn_out", C.POINTER(GnOut))]


_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> (restype, argtypes); mirrors include/sdmi.h declaration by declaration
PROTOTYPES = {
    "sdmi_version": (_i, []),
    "sdmi_act_dtype": (_i, []),
    "sdmi_last_error": (C.c_char_p, []),
    "sdmi_last_kernel": (C.c_char_p, []),
    "sdmi_pack_conv3x3": (_i, [_vp, _vp, _i, _i, _vp]),
    "sdmi_pack_linear": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sdmi_nchw_f32_to_nhwc_h16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sdmi_nhwc_h16_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sdmi_f32_to_h16_rows": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sdmi_groupnorm_ws_bytes": (_i64, [_i, _i, _i]),
    "sdmi_groupnorm_stats": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "sdmi_groupnorm_apply": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _u32, _vp]),
    "sdmi_groupnorm": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _u32, _vp, _vp, _vp]),
    "sdmi_groupnorm_static": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _u32, _vp]),
    "sdmi_groupnorm_static_max_bytes": (_i64, []),
    "sdmi_groupnorm_static_init": (_i, []),
    "sdmi_count_saturated": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "sdmi_layernorm": (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "sdmi_conv3x3": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _u32, _vp, _i64, _vp]),
    "sdmi_conv3x3_ws_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "sdmi_conv3x3_in": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "sdmi_conv3x3_in_ex": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "sdmi_gemm": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _f, _u32, _vp, _i64, _vp]),
    "sdmi_gemm_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _u32]),
    "sdmi_gemm_ex": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _f, _u32, _vp, _i64, _vp, _vp]),
    "sdmi_conv3x3_ex": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _u32, _vp, _i64, _vp, _vp]),
    "sdmi_conv3x3_gn_out_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _u32, _i]),
    "sdmi_conv3x3_gn_in_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "sdmi_gemm_stats_parts": (_i, [_i, _i, _i, _i, _i, _u32, _vp, _vp]),
    "sdmi_conv3x3_stats_parts": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _u32, _vp]),
    "sdmi_groupnorm_from_colstats": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _u32, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdmi_groupnorm_colstats_ss_bytes": (_i64, [_i, _i, _i]),
    "sdmi_groupnorm_colstats_finalize": (_i, [_i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdmi_pack_linear_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sdmi_tune_override": (None, [_i, _i]),
    "sdmi_tune_conv_window": (None, [_i]),
    "sdmi_prof_mid_event": (None, [_vp]),
    "sdmi_set_option": (_i, [C.c_char_p, _i]),
    "sdmi_probe_mfma": (_i, [_vp, _i, _i, _vp]),
    "sdmi_probe_copy": (_i, [_vp, _vp, _i64, _vp]),
    "sdmi_linear_small_m": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _u32, _vp]),
    "sdmi_timestep_embedding": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sdmi_attention": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _f, _vp]),
    "sdmi_cross_attention_lnq": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp, _vp, _vp, _i, _f, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "sdmi_inpaint_prepare": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sdmi_posterior_sample": (_i, [_vp, _vp, _f, _vp, _i64, _i, _i, _vp]),
    "sdmi_attention_d512": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _f, _vp]),
    "sdmi_attention_d512_log2q": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _vp]),
    "sdmi_attention_causal": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _f, _vp]),
    "sdmi_embed_tokens": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sdmi_gelu_rows": (_i, [_vp, _i, _i, _i, _vp]),
    "sdmi_ws_tickets_init": (_i, [_vp, _vp]),
    "sdmi_ws_tickets_release": (_i, [_vp]),
    "sdmi_attention_log2q": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "sdmi_cfg_ddim_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


class SdmiError(RuntimeError):
    pass


def load():
    """dlopen libsdmi355.so and bind every symbol of sdmi.h; raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdmiError("libsdmi355.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C stable-diffusion_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


# Optional per-entry-point timing (bench.py's roofline leg): when set to a KernelTimer, every call is bracketed
# by HIP events recorded on the stream the kernel is launched on.
TIMER = None


class KernelTimer(object):
    """Collects (entry point, algorithmic work, event pair) for every C-ABI call made while installed."""

    def __init__(self, work_fn):
        self.work_fn, self.records = work_fn, []

    def __enter__(self):
        global TIMER
        TIMER = self
        return self

    def __exit__(self, *exc):
        global TIMER
        TIMER = None

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, work, e0, e1 in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += work[0]
            d["bytes"] += work[1]
        return out


def call(name, *args):
    """Invoke an int-returning entry point; raise SdmiError(sdmi_last_error()) on a non-zero code."""
    lib = load()
    if TIMER is not None and not name.endswith(("_stats_parts", "_bytes")):    # host-side size queries launch nothing
        import torch
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        mid = None
        if name in ("sdmi_conv3x3", "sdmi_gemm", "sdmi_conv3x3_ex", "sdmi_gemm_ex"):
            # a split-K launch records this event between its main kernel and its fixed-order reduce kernel
            mid = torch.cuda.Event(enable_timing=True)
            mid.record(st)                      # creates the handle; the library records it again if it splits K
            lib.sdmi_prof_mid_event(C.c_void_p(mid.cuda_event))
        e0.record(st)
        rc = getattr(lib, name)(*args)
        e1.record(st)
        if mid is not None:
            lib.sdmi_prof_mid_event(None)
            kern = (lib.sdmi_last_kernel().decode() or name).split(" + ")   # kernel symbols as rocprofv3 reports them
            name = kern[0]
            if len(kern) > 1:
                TIMER.records.append((name, TIMER.work_fn(name, args), e0, mid))
                TIMER.records.append((kern[1], (0.0, 0.0), mid, e1))
                e0 = None
        elif name.startswith("sdmi_attention"):
            name = lib.sdmi_last_kernel().decode() or "attn64_kernel"
        if e0 is not None:
            TIMER.records.append((name, TIMER.work_fn(name, args), e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SdmiError("%s -> %d: %s" % (name, rc, lib.sdmi_last_error().decode()))
