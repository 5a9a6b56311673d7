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
// Runtime experiment switches (api.cpp; sdmi_set_option by name).  The launch paths read them by integer id: an array
// load, not a string search, on every launch.  X(name, default, meaning)
#define SDMI_OPTIONS(X)                                                                                                                       \
  X(conv_nmajor_by_rows, 0, "1: conv tile order by N > M as in round 1 (0: by operand bytes, 9 N > M)")                                          \
  X(splitk_target, 0, "> 0: blocks a split-K launch aims for per sample (default 512; round 1: 384)")                                            \
  X(splitk_max_tiles, 0, "> 0: split K only when a sample has at most this many tiles (default 256; round 1: 128)")                              \
  X(splitk_min_kt_conv, 0, "> 0: K-tiles (of 64) a conv split keeps at least (default 16)")                                                      \
  X(splitk_min_kt_gemm, 0, "> 0: same for plain GEMMs (default 24; round 1: 32)")                                                                \
  X(bm128_min, 0, "> 0: 128-row tiles once they number at least this (default 128)")                                                             \
  X(gemm_t2_min_blocks, 0, "> 0: plain GEMMs take 128 x 64 tiles once those number at least this (default 360; round 1: 600)")                   \
  X(split_colstats_off, 0, "1: split-K launches leave no column statistics (A/B switch)")                                                        \
  X(gemm_persist, 0, "persistent GEMM blocks: 0 off, 1 when a launch has more tiles than resident blocks, 2 always, 3 as 1 for GEGLU only")      \
  X(attn_d512_variant, 0, "d = 512 attention: 0 automatic (8-wave kernel), 1 the 4-wave kernel, 2 the 8-wave kernel")                            \
  X(gemm_small_m_tile, 1, "1: 64x64 tiles for K >= 2048 GEMMs with fewer than 1024 rows (0: 64x128 as in round 1)")                              \
  X(attn_x2_min_nq, 0, "> 0: query rows from which long self-attention takes the 64-rows-per-wave kernel (default 2048)")                        \
  X(attn_variant, 0, "attention kernel variant (0 = default)")                                                                                   \
  X(gemm_buf_off, 0, "1: GEMMs stage with the round-2 pointer-per-lane loader instead of buffer descriptors, A/B switch")                         \
  X(attn_pipe_off, 0, "1: long self-attention with the lockstep block of round 2 instead of the software-pipelined one, A/B switch")             \
  X(attn_buf_off, 0, "1: attn64x2_kernel stages K / V with per-lane pointers (round 2) instead of buffer descriptors, A/B switch")                \
  X(attn_pp, 0, "1: long self-attention on the two-group 8-wave kernel attn64pp_kernel (not faster: DESIGN.md, round 3)")                           \
  X(conv_win_256, 0, "1: 256 x 128 shared-window conv tiles wherever the kernel can run the problem (measured slower: an experiment)")              \
  X(gemm_astat, 0, "1: A-stationary GEMM kernel (igemm_astat_kernel) for K <= 384 wherever it can run the problem (measured slower)")           \
  X(conv_plan_r2, 0, "1: split-K / tile rules of the convs as in round 2 (A/B switch for the round-3 rule changes)")                              \
  X(conv_win_v1, 0, "1: shared-window convs on the round-2 kernel (global_load_lds + per-lane pointers), A/B switch")                           \
  X(conv_win_ups_off, 0, "1: nearest-2x upsample convs on the generic gather kernel instead of the shared-window variant, A/B switch")        \
  X(gn_apply_rows, 0, "> 0: rows per block of gn_apply_kernel in units of its 8-pass step (experiment)")
enum SdmiOpt {
#define SDMI_OPT_ENUM(name, def, doc) OPT_##name,
  SDMI_OPTIONS(SDMI_OPT_ENUM)
#undef SDMI_OPT_ENUM
  OPT_COUNT
};
extern int g_sdmi_opt[OPT_COUNT];
static inline int sdmi_opt(SdmiOpt id) { return g_sdmi_opt[id]; }

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
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (10 more vector instructions per element):
// every GroupNorm + SiLU path - gn_apply, gn_fused, the conv that applies the normalisation to its staged input - uses this
// one expression, so they agree bit for bit
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
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

// ---- LDS-DMA through a buffer descriptor (igemm.hip, attention.hip)
constexpr unsigned DMA_OOB = 0x80000000u;  // voffset of a lane that must fetch zeros (>= num_records of every descriptor here)
// one LDS-DMA wave-instruction: 64 lanes x 16 B -> 1 KB of LDS at byte address `lds` (wave-uniform); not counted by the
// compiler (inline asm): the loops wait with their own s_waitcnt vmcnt
__device__ __forceinline__ void dma_buf16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* smem_ptr) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem_ptr;
}
