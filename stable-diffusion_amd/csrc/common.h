// Shared device/host helpers for libsdmi355.so (gfx950 only: wave64, MFMA bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdmi.h"

typedef unsigned short u16;
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-bit storage type of activations and packed weights ("h16" in the ABI).  fp16 by default: same MFMA rate as
// bf16 on gfx950 and 3 more mantissa bits, which is what the parity-vs-fp32 budget needs; -DSDMI_BF16 selects bf16.
#ifdef SDMI_BF16
typedef __bf16 h16;
#define MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define MFMA32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#define MFMA32_MNEMONIC "v_mfma_f32_32x32x16_bf16"
#define DOT2H(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)
#define SDMI_ACT_DTYPE 1
#else
typedef _Float16 h16;
#define MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define MFMA32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#define MFMA32_MNEMONIC "v_mfma_f32_32x32x16_f16"
#define DOT2H(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)
#define SDMI_ACT_DTYPE 0
#endif
typedef h16 hx8 __attribute__((ext_vector_type(8)));
typedef h16 hx2 __attribute__((ext_vector_type(2)));

// host side: error reporting (defined in api.cpp)
void sdmi_set_error(const char* fmt, ...);
int sdmi_check_launch(const char* what);
void sdmi_note_kernel(const char* fmt, ...);  // records the kernel variant the last entry point launched
int sdmi_option(const char* name);           // runtime experiment switch (api.cpp; sdmi_set_option)

#define SDMI_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      sdmi_set_error(__VA_ARGS__);         \
      return SDMI_EINVAL;                  \
    }                                      \
  } while (0)

__device__ __forceinline__ float h2f(u16 v) { return (float)__builtin_bit_cast(h16, v); }
__device__ __forceinline__ h16 f2h16(float f) {
#ifdef SDMI_BF16
  return (h16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
#else
  return (h16)__builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f);  // saturate instead of +-inf
#endif
}
__device__ __forceinline__ u16 f2h(float f) { return __builtin_bit_cast(u16, f2h16(f)); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  hx2 v;
  v[0] = f2h16(lo);
  v[1] = f2h16(hi);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  const hx2 v = __builtin_bit_cast(hx2, w);
  lo = (float)v[0];
  hi = (float)v[1];
}
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  unpack2(v.x, f[0], f[1]); unpack2(v.y, f[2], f[3]); unpack2(v.z, f[4], f[5]); unpack2(v.w, f[6], f[7]);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2h(f[0], f[1]), pack2h(f[2], f[3]), pack2h(f[4], f[5]), pack2h(f[6], f[7]));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so
// neighbouring tiles (which share operand panels) hit the same L2.  Bijective for any nb.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
