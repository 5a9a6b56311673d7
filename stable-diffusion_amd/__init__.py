"""MI355X-native SD-2.x inpainting hot path (UNet step + AutoencoderKL decode).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI (include/sdmi.h) -> libsdmi355.so
  _lib.py    ctypes binding of that C-ABI (raises if the library is missing: no CPU fallback)
  engine.py  NHWC-bf16 activation views, weight pre-pack cache, op wrappers
  dropin/    Unet.py / AutoencoderKL.py / layers/*.py: the reference's module surface
  sampler.py DDIM + classifier-free guidance loop on the fused update kernel
  shard.py   one-process-per-GPU image sharding + RCCL all-gather
  synth.py   deterministic synthetic weights / inputs
"""
__all__ = ["synth"]
