"""MI355X-native SD-2.x inpainting hot path (UNet step + AutoencoderKL decode).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI (include/sdmi.h) -> libsdmi355.so
  _lib.py    ctypes binding of that C-ABI (raises if the library is missing: no CPU fallback)
  engine.py  NHWC-bf16 activation views, weight pre-pack cache, op wrappers
  dropin/    Unet.py / AutoencoderKL.py / layers/*.py: the reference's module surface (top-level module names)
  sampler.py DDIM + classifier-free guidance loop on the fused update kernel
  shard.py   one-process-per-GPU image sharding + RCCL all-gather
  synth.py   deterministic synthetic weights / inputs
"""
import os
import sys

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def install_dropin():
    """Put dropin/ first on sys.path so `from Unet import Unet`, `from AutoencoderKL import AutoencoderKL`,
    `from layers.ResBlock import ...` (the reference's import surface, LatentDiffusionInpainting.py:2-3) resolve to
    the HIP-backed modules."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    for name in ("Unet", "AutoencoderKL", "layers", "layers.ResBlock", "layers.SpatialTransformer"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(DROPIN_DIR):
            raise ImportError("module %r is already imported from %s; install_dropin() must run first"
                              % (name, getattr(mod, "__file__", "?")))
