// Shared device/host helpers for libsdmi355.so (gfx950 only: wave64, MFMA bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdmi.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// host side: error reporting (defined in api.cpp)
void sdmi_set_error(const char* fmt, ...);
int sdmi_check_launch(const char* what);

#define SDMI_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      sdmi_set_error(__VA_ARGS__);         \
      return SDMI_EINVAL;                  \
    }                                      \
  } while (0)

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
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
