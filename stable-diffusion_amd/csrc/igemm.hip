// igemm.hip - bf16 MFMA GEMM  out[M][N] = A[M][K] * W[N][K]^T  with an optional on-the-fly im2col
// A-loader (3x3 convolution, stride 1/2, explicit padding, nearest-2x upsample gather) and a fused
// epilogue (alpha, bias, per-sample channel bias = timestep embedding, residual, GEGLU, NCHW-fp32 out).
//
// Replaces nn.Conv2d(k=3) / nn.Linear / nn.Conv2d(k=1) / F.interpolate / F.pad / GEGLU of the reference
// (layers/ResBlock.py:17-19,:36-37,:67-76; layers/SpatialTransformer.py:22-26,:62-66,:110-129,:199,:204).
//
// Structure (gfx950): WM x WN waves over a BM x BN tile (table kVariants), BK = 64 per stage.  Both operands are
// K-contiguous and staged by LDS-DMA (global_load_lds_dwordx4, 16 B per lane) into an NSTAGE-slot LDS ring whose XOR
// swizzle (conflict-free ds_read_b128 fragment reads) is applied on the source address.  The K loop is ordered by one
// counted s_waitcnt vmcnt(N) + one raw s_barrier per K-tile, fragments are double-buffered in registers (every
// ds_read batch is issued ahead of an MFMA batch that does not need it).  v_mfma_f32_16x16x32_f16 with the WEIGHT
// fragment as the A operand, so a lane ends up with 4 consecutive output channels of one pixel; the epilogue widens
// that to 8 channels / 128-byte row segments with two ds_swizzle lane exchanges (no LDS memory, no barrier), with its
// bias / residual operands loaded before the K loop.
// Measured (per-block phase timers, MI355X): the K loop runs at the L2 -> LDS rate of about 70 GB/s per CU (64 KB in
// flight per CU), not at the MFMA rate; tile shape and prefetch depth are chosen for that.  One block per CU (256x128 with
// 8 waves, 128x256 with 4) measured 20-60 % slower than two co-resident 128x128 blocks: nothing covers the DMA wait.
// Built with -mllvm -amdgpu-mfma-vgpr-form (Makefile): accumulators stay in arch VGPRs, 3-10 % faster here.
#include <stdlib.h>
#include "common.h"
#include <mutex>

struct IgemmParams {
  const u16* a; const u16* w; void* out; const float* bias; const float* rowbias; const u16* res;
  long long sa, sw, sc;
  int lda, ldw, ldc, ldr, ldrb;
  int M, N, K;
  int rows_per_sample;
  float alpha;
  uint32_t flags;
  // conv geometry (CONV only)
  int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, ups, kt_per_tap;
  int nk;
  // split-K: grid.z splits of kt_per_split K-tiles each write fp32 partials [split][batch][M][npad] to ws
  int splits, kt_per_split, npad;
  float* ws;
  unsigned* ticket;  // split-K: per-tile arrival counters (in-kernel reduction by the last arriver); null = reduce kernel
  int nmajor;    // tile order: consecutive tile ids walk M first (an XCD's run of tiles shares weight panels)
  int fast_epi;  // host-checked: shapes / alignments allow the LDS-transposed 16-byte epilogue
  // epilogue extensions (sdmi_epilogue_ext): statistics for the normalisation that follows / a LayerNorm folded in
  float* colstats;         // [row part][N][2]: per-channel (sum, sumsq) over each wave tile's rows
  float* rowstats;         // [M][col_parts][2]: per-row (sum, sumsq) over each wave tile's columns
  const float* ln_stats;   // LNFOLD: [M][ln_parts][2] row partials of the input
  const float* ln_u;       // LNFOLD: [N] column sums of the packed weights
  int ln_parts, col_parts;
  float ln_eps;
  // GroupNorm + SiLU of the INPUT applied to the staged windows (sdmi_epilogue_ext.gn_in; conv_win2_kernel<.., GNA = true>)
  const float* gn_ss;      // [sample][Cin][2] (scale, shift) of the raw input a
};

__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};  // non-const: stays in the global address space (no flat loads)

constexpr int BK = 64;  // K elements per pipeline stage (128-byte tile rows)
// element offset of 16-byte chunk `chunk` (0..7) of tile row `row`; the XOR keeps the 16 lanes of a ds_read_b128 lane
// group on 16 distinct 16-byte slots of the 256-byte bank row (conflict-free)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }
// exact (erf) GELU of the reference's GEGLU (layers/SpatialTransformer.py:37-47, F.gelu):
//   gelu(g) = max(g, 0) - |g| h(|g| / sqrt 2),   h(z) = erfc(z) / 2 = 2^p(z)
// with p a degree-8 polynomial fit of log2(erfc(z) / 2) on [0, 4] (weighted for the absolute error of erfc; negative leading
// coefficient, so p -> -inf and h -> 0 beyond the fit range: no clamp).  One transcendental and 11 plain instructions per
// GELU - round 2's Abramowitz-Stegun 7.1.26 form took two transcendentals and 14 (the library erff 34): the GEGLU GEMMs of the
// 96x96 level have K = 320, five K-tiles per 128 x 128 tile, and their 32 GELUs per thread cost more issue time than their
// MFMAs.  |error| <= 2.7e-7 for |g| <= 12 evaluated in fp32 (A-S: 2.1e-7), exact limits g and -0 for large |g|; small negative
// outputs keep their sign (the product -|g| h, never a difference of near-equal numbers).
__device__ __forceinline__ float gelu_erf(float g) {
  const float z = fabsf(g) * 0.70710678118654752f;
  float p = -4.060196594e-05f;
  p = __builtin_fmaf(p, z, 4.048108837e-04f);
  p = __builtin_fmaf(p, z, -1.351170171e-03f);
  p = __builtin_fmaf(p, z, -1.014353406e-03f);
  p = __builtin_fmaf(p, z, 2.848045249e-02f);
  p = __builtin_fmaf(p, z, -1.485975645e-01f);
  p = __builtin_fmaf(p, z, -9.183867066e-01f);
  p = __builtin_fmaf(p, z, -1.627911755e+00f);
  p = __builtin_fmaf(p, z, -9.999999142e-01f);
  return __builtin_fmaf(-fabsf(g), __builtin_amdgcn_exp2f(p), fmaxf(g, 0.f));
}

// Finish 4 consecutive output channels n0..n0+3 of row m (alpha, bias, per-sample bias, residual) and store them in
// the requested layout.  Shared by the GEMM kernel's register epilogue and the split-K reduce kernel.
__device__ __forceinline__ void epilogue4(const IgemmParams& p, int bz, int m, int n0, float (&y)[4]) {
  const uint32_t fl = p.flags;
  const int hwo = p.rows_per_sample;
  const bool full = n0 + 3 < p.N;
  int sample = 0, pix = m;
  if (fl & (SDMI_F_ROWBIAS | SDMI_F_OUT_NCHW_F32)) { sample = m / hwo; pix = m - sample * hwo; }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] *= p.alpha;
    const int n = (full || n0 + e < p.N) ? n0 + e : n0;
    if (fl & SDMI_F_BIAS) y[e] += p.bias[n];
    if (fl & SDMI_F_ROWBIAS) y[e] += p.rowbias[(long long)sample * p.ldrb + n];
  }
  if (fl & SDMI_F_RESIDUAL) {
    const u16* r = p.res + (long long)bz * p.sc + (long long)m * p.ldr + n0;
    if (full) {
      const uint2 rv = *(const uint2*)r;
      float r0, r1, r2, r3;
      unpack2(rv.x, r0, r1);
      unpack2(rv.y, r2, r3);
      y[0] += r0; y[1] += r1; y[2] += r2; y[3] += r3;
    } else {
      for (int e = 0; e < 4; ++e) if (n0 + e < p.N) y[e] += h2f(r[e]);
    }
  }
  if (fl & SDMI_F_OUT_U8_IMAGE) {
    // torch.clamp((out + 1.0) / 2.0, 0, 1) * 255 -> uint8 (truncation), the reference's fp32 operation order
    unsigned char* o = (unsigned char*)p.out + (long long)m * p.N + n0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (full || n0 + e < p.N) {
        float v = (y[e] + 1.0f) / 2.0f;
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        o[e] = (unsigned char)(v * 255.0f);
      }
  } else if (fl & SDMI_F_OUT_NCHW_F32) {
    float* o = (float*)p.out + ((long long)sample * p.N + n0) * hwo + pix;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (full || n0 + e < p.N) o[(long long)e * hwo] = y[e];
  } else if (fl & SDMI_F_OUT_F32) {
    float* o = (float*)p.out + (long long)bz * p.sc + (long long)m * p.ldc + n0;
    if (full) *(float4*)o = make_float4(y[0], y[1], y[2], y[3]);
    else for (int e = 0; e < 4; ++e) if (n0 + e < p.N) o[e] = y[e];
  } else {
    u16* o = (u16*)p.out + (long long)bz * p.sc + (long long)m * p.ldc + n0;
    if (full) *(uint2*)o = make_uint2(pack2h(y[0], y[1]), pack2h(y[2], y[3]));
    else for (int e = 0; e < 4; ++e) if (n0 + e < p.N) o[e] = f2h(y[e]);
  }
}

// split-K second pass: sum the fp32 partials in split order (deterministic) and run the epilogue.
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmParams p, int batch) {
  const int nq = p.npad >> 2;
  const long long per_split = (long long)batch * p.M * p.npad;
  const long long total = (long long)batch * p.M * nq;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % nq);
    const long long r = i / nq;
    const int m = (int)(r % p.M), bz = (int)(r / p.M);
    const float* src = p.ws + r * p.npad + q * 4;
    float4 a = *(const float4*)src;
    for (int s = 1; s < p.splits; ++s) {
      const float4 b = *(const float4*)(src + s * per_split);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float y[4] = {a.x, a.y, a.z, a.w};
    epilogue4(p, bz, m, q * 4, y);
  }
}

// The same, leaving per-channel (sum, sumsq) of every 32-row part for the GroupNorm that follows (colstats, as the fast
// epilogue of an unsplit launch does): a block owns 32 rows x 32 columns, a thread one row's 4 columns; the 32 rows of a
// column are summed over the 8 rows of a wave by shuffles and over the 4 waves through the LDS, in a fixed order.  batch 1.
__global__ __launch_bounds__(256) void igemm_splitk_reduce_stats_kernel(const IgemmParams p, float* __restrict__ colstats) {
  __shared__ float sm[4][8][8];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int chunks = (p.N + 31) / 32;
  const int part = blockIdx.x / chunks, chunk = blockIdx.x - part * chunks;
  const int m = part * 32 + (t >> 3), n0 = chunk * 32 + (t & 7) * 4;
  const long long per_split = (long long)p.M * p.npad;
  float y[4] = {0.f, 0.f, 0.f, 0.f};
  const bool live = n0 < p.N;  // N % 4 == 0 on this path (host): a thread's 4 columns are all inside or all outside
  if (live) {
    const float* src = p.ws + (long long)m * p.npad + n0;
    float4 a = *(const float4*)src;
    for (int s = 1; s < p.splits; ++s) {
      const float4 b = *(const float4*)(src + s * per_split);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
    epilogue4(p, 0, m, n0, y);
  }
  float sq[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sq[e] = live ? y[e] : 0.f; sq[4 + e] = sq[e] * sq[e]; }
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // over the wave's 8 rows: lanes l, l ^ 8, l ^ 16, l ^ 32 hold the same columns
    sq[e] += __shfl_xor(sq[e], 8);
    sq[e] += __shfl_xor(sq[e], 16);
    sq[e] += __shfl_xor(sq[e], 32);
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[wv][lane][e] = sq[e];
  }
  __syncthreads();
  if (t < 32) {  // column chunk*32 + t: thread t's column sits in lane group t >> 2, element t & 3
    const int g = t >> 2, e = t & 3, n = chunk * 32 + t;
    if (n < p.N) {
      const float s = ((sm[0][g][e] + sm[1][g][e]) + sm[2][g][e]) + sm[3][g][e];
      const float q = ((sm[0][g][4 + e] + sm[1][g][4 + e]) + sm[2][g][4 + e]) + sm[3][g][4 + e];
      *(float2*)(colstats + ((long long)part * p.N + n) * 2) = make_float2(s, q);
    }
  }
}


// Split-K second pass fused with the GroupNorm(+SiLU) that follows the conv (sdmi_epilogue_ext.gn_out; verdict item 5): on the
// 24x24 / 12x12 levels every conv is a split-K launch and the next kernel is a GroupNorm of its (small) result.  One
// 1024-thread block owns one (sample, group): it sums the group's fp32 partials in split order, runs the epilogue (alpha, bias,
// per-sample bias, residual), rounds to the storage type - the values igemm_splitk_reduce_kernel would have stored - keeps
// them in registers, takes the exact two-pass statistics of gn_fused_kernel (csrc/norm.hip) from them and writes the
// normalised tensor; the raw tensor is written too when someone else reads it (keep_raw).  Same summation orders, same
// expressions as the reduce kernel followed by gn_fused_kernel: bit-identical to that pair (tested), one launch and one
// round trip of the tensor less.
struct GnOutParams {
  u16* y; int ldy; const float* gamma; const float* beta; float eps; int groups; int silu; int keep_raw;
};
template <int DPT>
__global__ __launch_bounds__(1024) void igemm_splitk_reduce_gn_kernel(const IgemmParams p, const GnOutParams go) {
  __shared__ float s_red[16];
  __shared__ float s_sc[160], s_sh[160];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int groups = go.groups, cpg = p.N / groups, hw = p.rows_per_sample;
  int g, n;
  {
    const int bid = blockIdx.x;
    if ((groups & 7) == 0) {  // blocks b and b + 8 share an XCD: neighbouring groups (they share 128-byte lines) on one XCD
      const int gpx = groups >> 3, xcd = bid & 7, k = bid >> 3;
      g = xcd * gpx + k % gpx;
      n = k / gpx;
    } else {
      g = bid % groups;
      n = bid / groups;
    }
  }
  const int dpr = cpg >> 1;  // channel pairs per row of this group
  const int total = hw * dpr;
  const int qstep = 1024 / dpr, rstep = 1024 - qstep * dpr;
  const long long per_split = (long long)p.M * p.npad;
  const uint32_t fl = p.flags;
  float2 acc[DPT];
  int off[DPT];  // row * 4096 + pair index (hw <= 2^19, dpr <= 80)
  {
    int row = t / dpr, d = t - row * dpr;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const bool ok = k * 1024 + t < total;
      off[k] = ok ? (row << 12) | d : -1;
      d += rstep; row += qstep;
      if (d >= dpr) { d -= dpr; ++row; }
    }
  }
  const float* const zp = (const float*)g_zero16;
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    const long long m = (long long)n * hw + (off[k] >> 12);
    acc[k] = *(const float2*)(off[k] >= 0 ? p.ws + m * p.npad + g * cpg + 2 * (off[k] & 4095) : zp);
  }
  for (int s = 1; s < p.splits; ++s) {
    float2 b[DPT];
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const long long m = (long long)n * hw + (off[k] >> 12);
      b[k] = *(const float2*)(off[k] >= 0 ? p.ws + s * per_split + m * p.npad + g * cpg + 2 * (off[k] & 4095) : zp);
    }
#pragma unroll
    for (int k = 0; k < DPT; ++k) { acc[k].x += b[k].x; acc[k].y += b[k].y; }
  }
  unsigned v[DPT];
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    const int row = off[k] >> 12, c0 = g * cpg + 2 * (off[k] & 4095);
    const long long m = (long long)n * hw + row;
    float y0 = acc[k].x * p.alpha, y1 = acc[k].y * p.alpha;
    if (off[k] >= 0) {
      if (fl & SDMI_F_BIAS) { y0 += p.bias[c0]; y1 += p.bias[c0 + 1]; }
      if (fl & SDMI_F_ROWBIAS) { y0 += p.rowbias[(long long)n * p.ldrb + c0]; y1 += p.rowbias[(long long)n * p.ldrb + c0 + 1]; }
      if (fl & SDMI_F_RESIDUAL) {
        float r0, r1;
        unpack2(*(const unsigned*)(p.res + m * p.ldr + c0), r0, r1);
        y0 += r0; y1 += r1;
      }
      v[k] = pack2h(y0, y1);
      if (go.keep_raw) *(unsigned*)((u16*)p.out + m * p.ldc + c0) = v[k];
    } else {
      v[k] = 0u;
    }
  }
  // ---- gn_fused_kernel's statistics and apply, on the rounded values
  float sm = 0.f;
#pragma unroll
  for (int k = 0; k < DPT; ++k) { float a, b; unpack2(v[k], a, b); sm += a + b; }
  sm = wave_sum(sm);
  if (lane == 0) s_red[wv] = sm;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += s_red[w];
  const float cnt = (float)hw * (float)cpg;
  const float mean = tot / cnt;
  __syncthreads();
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    float a, b; unpack2(v[k], a, b);
    const float da = a - mean, db = b - mean;
    ss += (off[k] >= 0) ? __builtin_fmaf(db, db, da * da) : 0.f;
  }
  ss = wave_sum(ss);
  if (lane == 0) s_red[wv] = ss;
  __syncthreads();
  float vtot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) vtot += s_red[w];
  const float rstd = 1.0f / sqrtf(vtot / cnt + go.eps);
  if (t < cpg) {
    const float sc = go.gamma[g * cpg + t] * rstd;
    s_sc[t] = sc;
    s_sh[t] = __builtin_fmaf(-mean, sc, go.beta[g * cpg + t]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < DPT; ++k) {
    const int row = off[k] >> 12, d = off[k] >= 0 ? off[k] & 4095 : 0;
    float a, b; unpack2(v[k], a, b);
    float oa = __builtin_fmaf(a, s_sc[2 * d], s_sh[2 * d]), ob = __builtin_fmaf(b, s_sc[2 * d + 1], s_sh[2 * d + 1]);
    if (go.silu) { oa = silu_f(oa); ob = silu_f(ob); }
    if (off[k] >= 0) *(unsigned*)(go.y + ((long long)n * hw + row) * go.ldy + g * cpg + 2 * d) = pack2h(oa, ob);
  }
}


// ---- fast epilogue.  The MFMA layout leaves a lane with 4 channels of one row (8-byte pieces, a dependent bias /
// residual load each).  Lanes l and l^16 hold neighbouring 4-channel pieces of the same row, so for every pair of
// 16-column tiles one ds_swizzle exchange (no LDS memory, no barrier) leaves each lane with 8 consecutive channels:
// the even lane of the pair finishes the first tile, the odd lane the second.  Bias / per-sample bias / residual are
// 16-byte loads issued BEFORE the K loop (epi_prefetch: they land under the main loop), the stores are 16 bytes per
// lane.  Arithmetic order per element is that of epilogue4 (alpha, bias, per-sample bias, residual).
template <int TM, int TN, bool GEGLU>
struct EpiRegs {
  static constexpr int MT = TM / 16, NT = TN / 16;
  static constexpr int VT = GEGLU ? NT / 2 : NT;  // 16-column value tiles per wave row
  static constexpr int NP = VT / 2;               // tile pairs = 8-channel chunks per lane per 16 rows
  static_assert(VT % 2 == 0 && NP >= 1, "fast epilogue pairs 16-column tiles");
  f32x4 bia[NP][2];
  f32x4 gb[GEGLU ? NT : 1];  // GEGLU: value / gate bias in MFMA layout
  f32x4 rb[NP][2];           // per-sample bias of the first sample the tile touches
  u32x4 rres[MT * NP];       // residual, 8 channels per (16-row step, pair)
  int s_first;
};
__device__ __forceinline__ float swz16(float v) {  // value of lane l ^ 16
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
// first output column of the 8-channel chunk this lane finishes for tile pair pr
template <bool GEGLU>
__device__ __forceinline__ int epi_col(int col0, int pr, int lane) {
  const int lq = lane >> 4;
  return (GEGLU ? col0 >> 1 : col0) + (2 * pr + (lq & 1)) * 16 + (lq >> 1) * 8;
}

template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ void epi_prefetch(const IgemmParams& p, EpiRegs<TM, TN, GEGLU>& r, int bz, int row0, int col0, int lane) {
  using R = EpiRegs<TM, TN, GEGLU>;
  const uint32_t fl = p.flags;
  const int l15 = lane & 15, lq = lane >> 4;
  const int nout = GEGLU ? p.N >> 1 : p.N;
  const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
  // (a wave whose rows all lie past M still prefetches: clamp, or it would read a sample that does not exist)
  r.s_first = (fl & SDMI_F_ROWBIAS) ? min(row0, p.M - 1) / p.rows_per_sample : 0;
  if constexpr (GEGLU) {
#pragma unroll
    for (int j = 0; j < R::NT; ++j) r.gb[j] = z;
    if (fl & SDMI_F_BIAS) {
#pragma unroll
      for (int j = 0; j < R::NT; j += 2) {
        const int nb0 = col0 + j * 16;
        const int oc = nb0 < p.N ? (nb0 >> 5) * 16 + lq * 4 : 0;
        r.gb[j] = *(const f32x4*)(p.bias + oc);
        r.gb[j + 1] = *(const f32x4*)(p.bias + nout + oc);
      }
    }
  } else {
    r.gb[0] = z;
  }
#pragma unroll
  for (int pr = 0; pr < R::NP; ++pr) {
    const int oc = epi_col<GEGLU>(col0, pr, lane);
    const int ocol = oc < nout ? oc : 0;
    r.bia[pr][0] = r.bia[pr][1] = r.rb[pr][0] = r.rb[pr][1] = z;
    if constexpr (!GEGLU) {
      if (fl & SDMI_F_BIAS) {
        r.bia[pr][0] = *(const f32x4*)(p.bias + ocol);
        r.bia[pr][1] = *(const f32x4*)(p.bias + ocol + 4);
      }
      if (fl & SDMI_F_ROWBIAS) {
        const float* rp = p.rowbias + (long long)r.s_first * p.ldrb + ocol;
        r.rb[pr][0] = *(const f32x4*)rp;
        r.rb[pr][1] = *(const f32x4*)(rp + 4);
      }
    }
    if (fl & SDMI_F_RESIDUAL) {
      const u16* rbase = p.res + (long long)bz * p.sc + ocol;
#pragma unroll
      for (int i = 0; i < R::MT; ++i) {
        const int m = min(row0 + i * 16 + l15, p.M - 1);
        r.rres[i * R::NP + pr] = *(const u32x4*)(rbase + (long long)m * p.ldr);
      }
    } else {
#pragma unroll
      for (int i = 0; i < R::MT; ++i) r.rres[i * R::NP + pr] = u32x4{0u, 0u, 0u, 0u};
    }
  }
}

// s += v, q += v * v as two plain VALU instructions.  Inline asm on purpose: left to the SLP vectoriser these sums became
// v_pk_mul_f32 / v_pk_add_f32 chains with op_sel half swaps, and on MI355X such a chain rarely (about 1 result in 10^4, not
// reproducible run to run) lost one element's square while the plain sum beside it stayed exact.
__device__ __forceinline__ void stat_acc(float& s, float& q, float v) {
  float t;
  asm("v_mul_f32 %2, %3, %3\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %3" : "+v"(s), "+v"(q), "=&v"(t) : "v"(v));
}

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4): every lane ends up with the row's total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

typedef const float __attribute__((address_space(3)))* lds_cfp;
// lnside: LDS side region of an LNFOLD launch: float2 (mean, rstd) per tile row, then the tile's BN column sums u; lrow0 /
// lcol0 = this wave's first row / column inside the block tile; spart / cpart = index of the wave tile's first 32-row part /
// 32-column part
template <int TM, int TN, bool GEGLU, bool LNC, bool CS>
__device__ __forceinline__ void epi_finish(const IgemmParams& p, const EpiRegs<TM, TN, GEGLU>& r,
                                           f32x4 (&acc)[TM / 16][TN / 16], int bz, int row0, int col0, int lane,
                                           lds_cfp lnside, int ln_bm, int lrow0, int lcol0, int spart, int cpart) {
  using R = EpiRegs<TM, TN, GEGLU>;
  const uint32_t fl = p.flags;
  const int l15 = lane & 15;
  const bool odd = (lane >> 4) & 1;
  const bool lnf = LNC && (fl & SDMI_F_LNFOLD) != 0;  // LNC = false (convolutions): the LayerNorm fold and the row statistics
  const bool cstat = CS && p.colstats != nullptr;     // compile away; CS = false: so do the column statistics
  const lds_cfp lnu = lnside + 2 * ln_bm;
  const int nout = GEGLU ? p.N >> 1 : p.N;
  const int hwo = p.rows_per_sample;
  // rows of a second sample inside this tile (host guarantees rows_per_sample >= BM: at most two samples): their
  // per-sample bias is fetched here, the rare case pays one memory latency
  const int s_split = (r.s_first + 1) * hwo;
  const bool two = (fl & SDMI_F_ROWBIAS) && row0 + TM > s_split && s_split < p.M;
  int ocol[R::NP];
  bool cvalid[R::NP];
  f32x4 rb2[R::NP][2];
#pragma unroll
  for (int pr = 0; pr < R::NP; ++pr) {
    const int oc = epi_col<GEGLU>(col0, pr, lane);
    cvalid[pr] = oc < nout;
    ocol[pr] = cvalid[pr] ? oc : 0;
    rb2[pr][0] = r.rb[pr][0];
    rb2[pr][1] = r.rb[pr][1];
    if constexpr (!GEGLU) {
      if (two) {
        const float* rp = p.rowbias + (long long)(r.s_first + 1) * p.ldrb + ocol[pr];
        rb2[pr][0] = *(const f32x4*)rp;
        rb2[pr][1] = *(const f32x4*)(rp + 4);
      }
    }
  }
  // LNFOLD: the 8 column sums u of each chunk this lane finishes (non-GEGLU; GEGLU: of every 16-column tile, MFMA layout),
  // from the LDS side region
  f32x4 gu[GEGLU ? R::NT : 1];
  if constexpr (GEGLU) {
#pragma unroll
    for (int j = 0; j < R::NT; ++j)
      gu[j] = lnf ? *(const f32x4 __attribute__((address_space(3)))*)(lnu + lcol0 + j * 16 + (lane >> 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    gu[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 lu[R::NP][2];
  float csum[R::NP][8], csq[R::NP][8];  // colstats: per-channel sums over this wave tile's rows
#pragma unroll
  for (int pr = 0; pr < R::NP; ++pr) {
    lu[pr][0] = lu[pr][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!GEGLU) {
      if (lnf) {
        const int lc = lcol0 + (2 * pr + ((lane >> 4) & 1)) * 16 + (lane >> 5) * 8;
        lu[pr][0] = *(const f32x4 __attribute__((address_space(3)))*)(lnu + lc);
        lu[pr][1] = *(const f32x4 __attribute__((address_space(3)))*)(lnu + lc + 4);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[pr][e] = csq[pr][e] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < R::MT; ++i) {
    const int m = row0 + i * 16 + l15;
    float2 st = make_float2(0.f, 1.f);  // LNFOLD: (mean, rstd) of row m
    if (lnf) {
      const lds_cfp sp = lnside + 2 * (lrow0 + i * 16 + l15);
      st = make_float2(sp[0], sp[1]);
    }
    float y[R::NP][8];
#pragma unroll
    for (int pr = 0; pr < R::NP; ++pr) {
      f32x4 va, vb;  // value tiles 2*pr and 2*pr+1 of this 16-row step, MFMA layout
      if constexpr (GEGLU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xa = (acc[i][4 * pr][e] * p.alpha - st.x * gu[4 * pr][e]) * st.y + r.gb[4 * pr][e];
          const float ga = (acc[i][4 * pr + 1][e] * p.alpha - st.x * gu[4 * pr + 1][e]) * st.y + r.gb[4 * pr + 1][e];
          const float xb = (acc[i][4 * pr + 2][e] * p.alpha - st.x * gu[4 * pr + 2][e]) * st.y + r.gb[4 * pr + 2][e];
          const float gb = (acc[i][4 * pr + 3][e] * p.alpha - st.x * gu[4 * pr + 3][e]) * st.y + r.gb[4 * pr + 3][e];
          va[e] = xa * gelu_erf(ga);
          vb[e] = xb * gelu_erf(gb);
        }
      } else {
        va = acc[i][2 * pr];
        vb = acc[i][2 * pr + 1];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float got = swz16(odd ? va[e] : vb[e]);  // even lane hands over its piece of tile b, odd lane of tile a
        y[pr][e] = odd ? got : va[e];
        y[pr][e + 4] = odd ? vb[e] : got;
      }
      if constexpr (!GEGLU) {
        // absent terms are zeros (adding +0 changes nothing that survives the fp16 rounding)
        const bool second = m >= s_split;
        const f32x4 rs0 = second ? rb2[pr][0] : r.rb[pr][0], rs1 = second ? rb2[pr][1] : r.rb[pr][1];
        const u32x4 rv = r.rres[i * R::NP + pr];
        float q[8];
        unpack2(rv[0], q[0], q[1]); unpack2(rv[1], q[2], q[3]); unpack2(rv[2], q[4], q[5]); unpack2(rv[3], q[6], q[7]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (mean 0, rstd 1, u 0 without LNFOLD: the subtraction and the product change no bit)
          y[pr][e] = ((((y[pr][e] * p.alpha - st.x * lu[pr][0][e]) * st.y) + r.bia[pr][0][e]) + rs0[e]) + q[e];
          y[pr][e + 4] = ((((y[pr][e + 4] * p.alpha - st.x * lu[pr][1][e]) * st.y) + r.bia[pr][1][e]) + rs1[e]) + q[e + 4];
        }
      }
    }
    if constexpr (!GEGLU) {
      if (LNC && p.rowstats) {
        // per-row (sum, sumsq) of this wave tile's columns: in-lane over the lane's chunks, then over the 4 lanes that
        // share the row (lane ^ 16, lane ^ 32); fixed order
        // one part per 32 columns (= per tile pair), whatever the tile shape: the parts, and with them the LayerNorm
        // that sums them, do not depend on the variant the launch picked (batch invariance)
#pragma unroll
        for (int pr = 0; pr < R::NP; ++pr) {
          float rs_ = 0.f, rq_ = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) stat_acc(rs_, rq_, y[pr][e]);
          rs_ += __shfl_xor(rs_, 16); rq_ += __shfl_xor(rq_, 16);
          rs_ += __shfl_xor(rs_, 32); rq_ += __shfl_xor(rq_, 32);
          if ((lane >> 4) == 0 && m < p.M && cpart + pr < p.col_parts)
            *(float2*)(p.rowstats + ((long long)m * p.col_parts + cpart + pr) * 2) = make_float2(rs_, rq_);
        }
      }
      if (cstat) {
#pragma unroll
        for (int pr = 0; pr < R::NP; ++pr)
#pragma unroll
          for (int e = 0; e < 8; ++e) stat_acc(csum[pr][e], csq[pr][e], y[pr][e]);
        if (i & 1) {
          // one part per 32 rows, whatever the tile shape (batch invariance, as for the row statistics): fold the 16
          // rows held by the lanes of a DPP row; lane 0 of each row stores its chunk's 8 x (sum, sumsq)
#pragma unroll
          for (int pr = 0; pr < R::NP; ++pr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { csum[pr][e] = row16_sum(csum[pr][e]); csq[pr][e] = row16_sum(csq[pr][e]); }
            // (a wave tile that lies wholly past M - M a multiple of the wave tile but not of the block tile - has no part)
            if (l15 == 0 && cvalid[pr] && (spart + (i >> 1)) * 32 < p.M) {
              float* o = p.colstats + ((long long)(spart + (i >> 1)) * p.N + ocol[pr]) * 2;
#pragma unroll
              for (int e = 0; e < 8; e += 2) *(f32x4*)(o + 2 * e) = f32x4{csum[pr][e], csq[pr][e], csum[pr][e + 1], csq[pr][e + 1]};
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[pr][e] = csq[pr][e] = 0.f;
          }
        }
      }
    }
    if (fl & SDMI_F_OUT_F32) {
#pragma unroll
      for (int pr = 0; pr < R::NP; ++pr)
        if (m < p.M && cvalid[pr]) {
          float* o = (float*)p.out + (long long)bz * p.sc + (long long)m * p.ldc + ocol[pr];
          *(f32x4*)o = f32x4{y[pr][0], y[pr][1], y[pr][2], y[pr][3]};
          *(f32x4*)(o + 4) = f32x4{y[pr][4], y[pr][5], y[pr][6], y[pr][7]};
        }
    } else {
      u32x4 pk[R::NP];
#pragma unroll
      for (int pr = 0; pr < R::NP; ++pr)
        pk[pr] = u32x4{pack2h(y[pr][0], y[pr][1]), pack2h(y[pr][2], y[pr][3]), pack2h(y[pr][4], y[pr][5]), pack2h(y[pr][6], y[pr][7])};
      u16* ob = (u16*)p.out + (long long)bz * p.sc;
      if constexpr (R::NP == 2) {
        // second exchange (lanes l and l^8, packed halves): a store instruction then covers 8 rows x 128 contiguous
        // bytes instead of 16 rows x 64: rows 0-7 of the step go out first, rows 8-15 second
        const bool hi = l15 >= 8;
        u32x4 got;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          got[e] = (unsigned)__builtin_amdgcn_ds_swizzle((int)(hi ? pk[0][e] : pk[1][e]), 0x201F);  // lane ^ 8
        const u32x4 sa = hi ? got : pk[0], sb = hi ? pk[1] : got;
        const int oc = hi ? ocol[1] : ocol[0];
        const bool cv = hi ? cvalid[1] : cvalid[0];
        const int ma = row0 + i * 16 + (l15 & 7), mb = ma + 8;
        if (ma < p.M && cv) *(u32x4*)(ob + (long long)ma * p.ldc + oc) = sa;
        if (mb < p.M && cv) *(u32x4*)(ob + (long long)mb * p.ldc + oc) = sb;
      } else {
#pragma unroll
        for (int pr = 0; pr < R::NP; ++pr)
          if (m < p.M && cvalid[pr]) *(u32x4*)(ob + (long long)m * p.ldc + ocol[pr]) = pk[pr];
      }
    }
  }
}

// Everything after the K loop, shared by the GEMM / generic-conv kernel and the shared-window conv kernel: split-K partials
// (+ optional in-launch reduction) and the epilogue.  row0 / col0 = first output row / column of this wave's tile.
template <int TM, int TN, bool GEGLU, bool LNC, bool CS>
__device__ __forceinline__ void igemm_finish(const IgemmParams& p, f32x4 (&acc)[TM / 16][TN / 16],
                                             EpiRegs<TM, (!GEGLU || TN >= 64) ? TN : 64, GEGLU>& er, bool fast, u16* smem, int bz,
                                             int row0, int col0, lds_cfp lnside, int ln_bm, int lrow0, int lcol0, int spart,
                                             int cpart) {
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr bool FAST_OK = !GEGLU || TN >= 64;
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
  // ---- epilogue: lane holds out[m][n0..n0+3], m = tile row l15, n0 = 4*lq within each 16x16 tile ----
  const uint32_t fl = p.flags;
  if (p.splits > 1) {
    // split-K: every block leaves its fp32 partial tile in the workspace; the block that draws the tile's last ticket
    // sums all of them in split order (same bits whoever is last) and runs the epilogue.  Hand-off = write-through
    // (sc1) stores, every wave drained, the ticket, one agent-scope acquire in the reducing block.
    float* wsb = p.ws + ((long long)blockIdx.z * gridDim.y + bz) * p.M * p.npad;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = row0 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n0 = col0 + j * 16 + lq * 4;
        if (n0 < p.npad) {
          float* dst = wsb + (long long)m * p.npad + n0;
          if (p.ticket)  // write-through (sc1) store: visible to the reducing block without an L2 write-back fence
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[i][j]) : "memory");
          else
            *(f32x4*)dst = acc[i][j];
        }
      }
    }
    if (!p.ticket) return;  // igemm_splitk_reduce_kernel finishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // all partial stores issued and drained; every wave's trailing DMA has landed, the LDS is free
    int* flag = (int*)smem;
    if (t == 0) {
      unsigned* tk = p.ticket + ((long long)bz * gridDim.x + blockIdx.x);
      const unsigned prev = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)p.splits - 1u;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next call
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    if constexpr (FAST_OK) if (fast) epi_prefetch<TM, TN, GEGLU>(p, er, bz, row0, col0, lane);
    const long long per_split = (long long)gridDim.y * p.M * p.npad;
    const float* base = p.ws + (long long)bz * p.M * p.npad;
    for (int s = 0; s < p.splits; ++s) {
      const float* sb = base + s * per_split;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = min(row0 + i * 16 + l15, p.M - 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n0 = min(col0 + j * 16 + lq * 4, p.npad - 4);
          const f32x4 v = *(const f32x4*)(sb + (long long)m * p.npad + n0);
          acc[i][j] = s == 0 ? v : acc[i][j] + v;
        }
      }
    }
  }
  if constexpr (FAST_OK) if (fast) {
    epi_finish<TM, TN, GEGLU, LNC, CS>(p, er, acc, bz, row0, col0, lane, lnside, ln_bm, lrow0, lcol0, spart, cpart);
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = row0 + i * 16 + l15;
    if (m >= p.M) continue;
    if constexpr (GEGLU) {
      const int half = p.N >> 1;
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        const int nb0 = col0 + j * 16;
        if (nb0 >= p.N) continue;
        const int oc = (nb0 >> 5) * 16 + lq * 4;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xv = acc[i][j][e] * p.alpha, gv = acc[i][j + 1][e] * p.alpha;
          if (fl & SDMI_F_BIAS) { xv += p.bias[oc + e]; gv += p.bias[half + oc + e]; }
          y[e] = xv * gelu_erf(gv);
        }
        u16* o = (u16*)p.out + (long long)bz * p.sc + (long long)m * p.ldc + oc;
        *(uint2*)o = make_uint2(pack2h(y[0], y[1]), pack2h(y[2], y[3]));
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n0 = col0 + j * 16 + lq * 4;
        if (n0 >= p.N) continue;
        float y[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        epilogue4(p, bz, m, n0, y);
      }
    }
  }
}

// The second launch bound (2 waves per SIMD for the 4-wave tiles = the 256-register budget they run at anyway) matters:
// with the default 512-register budget hipcc split the accumulators over VGPRs and AGPRs and shuffled them with 74
// v_accvgpr moves per K-tile in the plain-GEMM instantiations (MFMA not accumulating in place).
// GEGLU is a template parameter, not a run-time flag: with the value/gate pairing of accumulators in the same kernel as
// the plain epilogue, hipcc's register coalescing gave up on in-place MFMA accumulation (74 v_accvgpr moves per K-tile).
// BUF (GEMM only, K % 64 == 0): staging through buffer descriptors (dma_buf16): a lane's byte offset inside the block's A / W
// panel is computed once, the K position is a scalar offset, rows past M / N carry an out-of-range offset (zeros) and K-tiles
// past the end go through a descriptor of zero records - no vector instruction in the K loop's staging at all.
// Diagnostic build only (make EXTRA=-DIGEMM_STAMP LIB=<another file>; scratch/r3/igemm_stamps.py reads the stamps): wave 0 of
// every block leaves s_memrealtime (100 MHz) at 0 kernel start, 5 tile decode done, 6 epilogue operands requested, 7 first
// DMAs issued, 1 prologue done, 2 first K-tile landed, 3 K loop done, 4 stores retired.
#ifdef IGEMM_STAMP
__device__ unsigned long long g_igemm_stamp[8192 * 8];
extern "C" int sdmi_debug_igemm_stamps(void* dst, int bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_igemm_stamp), bytes); }
#define IG_STAMP(I) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0 && blockIdx.x < 8192) g_igemm_stamp[blockIdx.x * 8 + (I)] = t_; }
#else
#define IG_STAMP(I)
#endif
template <int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, bool GEGLU, bool BUF = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && NSTAGE * (BM + BN) * BK * 2 <= 80 * 1024 ? 2 : 1)) void igemm_kernel(const IgemmParams p) {
  IG_STAMP(0);
  static_assert(!(BUF && CONV), "buffer-descriptor staging: GEMM only");
  constexpr int NTH = 64 * WM * WN, RP = NTH / 8;          // threads; tile rows staged per DMA pass
  constexpr int TM = BM / WM, TN = BN / WN;                // wave tile
  constexpr int MT = TM / 16, NT = TN / 16, PA = BM / RP, PB = BN / RP;
  static_assert(BM % RP == 0 && BN % RP == 0 && TM % 16 == 0 && TN % 16 == 0, "tile / wave grid mismatch");
  constexpr int STAGE = (BM + BN) * BK;  // elements per pipeline stage
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // NSTAGE * STAGE halves (the only LDS object)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv / WN, wc = wv % WN, l15 = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int brow = (p.nmajor ? bid % tiles_m : bid / tiles_n) * BM, bcol = (p.nmajor ? bid / tiles_m : bid % tiles_n) * BN;
  const int bz = blockIdx.y;
  const u16* A = p.a + (long long)bz * p.sa;
  const u16* Wp = p.w + (long long)bz * p.sw;

  // Staging is LDS-DMA (global_load_lds_dwordx4): a wave-instruction moves 64 x 16 B = 8 tile rows (BK = 64 halves =
  // 128 B per row) straight into LDS: no VGPR round trip, no ds_write.  The LDS image is linear in lane order, so the
  // XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side: the lane that
  // fills physical chunk (t&7) of its row fetches logical chunk (t&7)^f(row).  Out-of-range rows / taps / k-chunks
  // fetch 16 zero bytes from g_zero16, so every load is unconditional.
  const u16* const zp = (const u16*)g_zero16;
  const int r0 = t >> 3;                       // tile row of pass 0 (pass i adds RP)
  const int lchunk = (t & 7) ^ ((r0 >> 1) & 7);  // logical 16-B chunk this lane fetches: f(row) = (row>>1)&7
  const int wave_row0 = __builtin_amdgcn_readfirstlane(wv) * 8;

  // ---- per-thread row bookkeeping, fixed over the K loop.  Everything that depends only on the row is hoisted here:
  // for a conv, the row pointer already includes the output pixel's position and this lane's chunk, and a 9-bit mask
  // says which taps fall inside the image, so a K-tile costs one test, one uniform offset add and one select per row.
  const u16* a_ptr[PA];
  int a_mask[PA];                 // CONV: bit t set = tap t readable; GEMM: 1 = row valid
  int a_oy[PA], a_ox[PA];         // CONV with upsample only
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int gm = brow + r0 + RP * i;
    const bool ok = gm < p.M;
    const int gmc = ok ? gm : 0;
    if constexpr (CONV) {
      const int hw = p.Ho * p.Wo;
      const int img = gmc / hw, rem = gmc - img * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int hlim = p.H << p.ups, wlim = p.W << p.ups;
      int mask = 0;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int iy = oy * p.stride + tp / 3 - p.pad_t, ix = ox * p.stride + tp % 3 - p.pad_l;
        if (ok && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim) mask |= 1 << tp;
      }
      a_mask[i] = mask;
      a_oy[i] = oy;
      a_ox[i] = ox;
      const u16* base = A + (long long)img * p.H * p.W * p.lda + lchunk * 8;
      // no upsample: pointer to tap (0,0) of this output pixel (may lie outside the image: only dereferenced when the
      // tap's mask bit is set); the tap then adds the uniform offset (ky*W + kx)*lda
      a_ptr[i] = p.ups ? base : base + ((long long)(oy * p.stride - p.pad_t) * p.W + (ox * p.stride - p.pad_l)) * p.lda;
    } else {
      a_mask[i] = ok ? 1 : 0;
      a_oy[i] = a_ox[i] = 0;
      a_ptr[i] = A + (long long)gmc * p.lda + lchunk * 8;
    }
  }
  bool b_ok[PB];
  const u16* b_ptr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int gn = bcol + r0 + RP * i;
    b_ok[i] = gn < p.N;
    b_ptr[i] = Wp + (long long)(b_ok[i] ? gn : 0) * p.ldw + lchunk * 8;
  }
  const int k_lim = (CONV ? p.Cin : p.K) - lchunk * 8;  // this lane's chunk is inside K iff (uniform k offset) < k_lim
  // BUF: descriptors of this block's panels, per-lane offsets inside them
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long long)brow * p.lda), 0, BUF ? DMA_OOB : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(Wp + (long long)bcol * p.ldw), 0, BUF ? DMA_OOB : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcZ = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0u, 0x00020000);  // no records: every lane reads zeros
  unsigned a_voff[PA], b_voff[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) a_voff[i] = (BUF && brow + r0 + RP * i < p.M) ? (unsigned)(((r0 + RP * i) * p.lda + lchunk * 8) * 2) : DMA_OOB;
#pragma unroll
  for (int i = 0; i < PB; ++i) b_voff[i] = (BUF && bcol + r0 + RP * i < p.N) ? (unsigned)(((r0 + RP * i) * p.ldw + lchunk * 8) * 2) : DMA_OOB;
  const unsigned lds0 = lds_addr_of(smem);

  const int kt_begin = blockIdx.z * p.kt_per_split;
  const int kt_end = min(p.nk, kt_begin + p.kt_per_split);
  // CONV: K-tiles run channel-chunk-major, the 9 taps of one 64-channel chunk back to back: they re-read the same input
  // patch (shifted), which is then still in the CU's vector cache
  int cb = kt_begin / 9, tap = kt_begin - cb * 9;  // position of the NEXT tile to stage

#define IGEMM_GLDS(SRC, DST) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
  // Stage K-tile KT into ring slot SLOT.  Always issues exactly PA+PB DMA instructions per wave (tiles past kt_end
  // fetch the zero page), so the counted s_waitcnt below is the same constant on every iteration.
#define IGEMM_STAGE_TILE(KT, SLOT)                                                                                \
  {                                                                                                               \
    u16* dA = smem + (SLOT) * STAGE + wave_row0 * BK;                                                             \
    u16* dB = dA + BM * BK;                                                                                       \
    const bool live = (KT) < kt_end;                                                                              \
    if constexpr (CONV) {                                                                                         \
      const int ky = tap / 3, kx = tap - 3 * ky;                                                                  \
      const int c0 = cb * BK;                                                                                     \
      const bool cok = live && c0 < k_lim;                                                                        \
      const int tbit = 1 << tap;                                                                                  \
      if (p.ups) {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                          \
          const int iy = (a_oy[i] + ky - p.pad_t) >> 1, ix = (a_ox[i] + kx - p.pad_l) >> 1;                       \
          const bool ok = (a_mask[i] & tbit) && cok;                                                              \
          const u16* src = ok ? a_ptr[i] + ((long long)(iy * p.W + ix) * p.lda + c0) : zp;                        \
          IGEMM_GLDS(src, dA + i * RP * BK);                                                                      \
        }                                                                                                         \
      } else {                                                                                                    \
        const long long toff = (long long)(ky * p.W + kx) * p.lda + c0;                                           \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                          \
          const bool ok = (a_mask[i] & tbit) && cok;                                                              \
          const u16* src = ok ? a_ptr[i] + toff : zp;                                                             \
          IGEMM_GLDS(src, dA + i * RP * BK);                                                                      \
        }                                                                                                         \
      }                                                                                                           \
      const int wk = tap * p.Cin + c0;                                                                            \
      _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                            \
        const u16* src = (b_ok[i] && cok) ? b_ptr[i] + wk : zp;                                                   \
        IGEMM_GLDS(src, dB + i * RP * BK);                                                                        \
      }                                                                                                           \
      if (++tap == 9) { tap = 0; ++cb; }                                                                          \
    } else if constexpr (BUF) {                                                                                   \
      const int soff = (KT) * (BK * 2);                                                                           \
      const __amdgpu_buffer_rsrc_t ra = live ? rsrcA : rsrcZ, rb = live ? rsrcB : rsrcZ;                          \
      const unsigned la = lds0 + ((SLOT) * STAGE + wave_row0 * BK) * 2;                                           \
      _Pragma("unroll") for (int i = 0; i < PA; ++i) dma_buf16(ra, a_voff[i], soff, la + i * (RP * BK * 2));      \
      _Pragma("unroll") for (int i = 0; i < PB; ++i) dma_buf16(rb, b_voff[i], soff, la + (BM * BK + i * RP * BK) * 2); \
    } else {                                                                                                      \
      const int ko = (KT) * BK;                                                                                   \
      const bool kok = live && ko < k_lim;                                                                        \
      _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                            \
        const u16* src = (a_mask[i] && kok) ? a_ptr[i] + ko : zp;                                                 \
        IGEMM_GLDS(src, dA + i * RP * BK);                                                                        \
      }                                                                                                           \
      _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                            \
        const u16* src = (b_ok[i] && kok) ? b_ptr[i] + ko : zp;                                                   \
        IGEMM_GLDS(src, dB + i * RP * BK);                                                                        \
      }                                                                                                           \
    }                                                                                                             \
  }
  // wait until all but the NSTAGE-2 youngest stage groups of THIS wave have landed and this wave's fragment reads of
  // the slot that is restaged next have returned (lgkmcnt), then meet the other waves
#define IGEMM_WAIT_STAGE(GROUPS)                                                                    \
  {                                                                                                 \
    constexpr int INFLIGHT = (GROUPS) * (PA + PB);                                                  \
    static_assert(INFLIGHT >= 0 && INFLIGHT < 64, "vmcnt is a 6-bit counter");                     \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");                      \
    __builtin_amdgcn_s_barrier();                                                                   \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // NSTAGE-slot ring.  Fragments are double-buffered in registers and the loop is skewed by half a K-tile, so every
  // ds_read batch is issued ahead of an MFMA batch that does not depend on it (the LDS latency hides under 16+ MFMAs)
  // and a K-tile costs one counted s_waitcnt + one raw s_barrier (__syncthreads() would drain the DMA queue):
  //   read kk=1 of tile k | MFMA kk=0 of tile k | wait: tile k+1 landed, slot of tile k fully read; barrier |
  //   stage tile k+NSTAGE into the slot of tile k | read kk=0 of tile k+1 | MFMA kk=1 of tile k
#define IGEMM_READ_FRAGS(AF, WF, SLOT, KK)                                                                       \
  {                                                                                                              \
    const u16* cA = smem + (SLOT) * STAGE;                                                                       \
    const u16* cB = cA + BM * BK;                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AF[i] = *(const hx8*)&cA[lds_off(wr * TM + i * 16 + l15, lq + 4 * (KK))]; \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const hx8*)&cB[lds_off(wc * TN + j * 16 + l15, lq + 4 * (KK))]; \
  }
#define IGEMM_MFMA_ALL(AF, WF)                                                                 \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(WF[j], AF[i], acc[i][j], 0, 0, 0); \
  }
  hx8 af0[MT], wf0[NT], af1[MT], wf1[NT];
  // epilogue operands first: they are in registers long before the K loop ends
  constexpr bool FAST_OK = !GEGLU || TN >= 64;
  EpiRegs<TM, FAST_OK ? TN : 64, GEGLU> er;
  const bool fast = FAST_OK && p.fast_epi;
  IG_STAMP(5);
  if constexpr (FAST_OK) if (fast && p.splits == 1) epi_prefetch<TM, TN, GEGLU>(p, er, bz, brow + wr * TM, bcol + wc * TN, lane);
  IG_STAMP(6);
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE; ++sidx) IGEMM_STAGE_TILE(kt_begin + sidx, sidx);
  IG_STAMP(7);
  // LayerNorm folded into this GEMM (SDMI_F_LNFOLD): per-row (mean, rstd) from the producer's row partials and this tile's
  // column sums u go to a side region of the LDS, behind the ring (same array: one LDS object).  Placed AFTER the prologue's
  // DMA issue: the loads' latency then overlaps the wait for the first K-tile (in front of it, it delayed the whole
  // pipeline by a memory round trip: +2...5 us on the 5-K-tile GEMMs).  Each row's partials are split over the NTH / BM
  // threads that share it and combined by shuffles; the first barrier publishes the region.
  float* const lnside = (float*)(smem + NSTAGE * STAGE);
  if constexpr (!CONV) {
    if (p.ln_stats) {
      {
        constexpr int TPR = NTH / BM;  // threads per tile row (2 or 4, neighbouring lanes)
        const int lrow = t / TPR, sub = t % TPR;
        const int row = min(brow + lrow, p.M - 1);
        const float2* sp = (const float2*)p.ln_stats + (long long)row * p.ln_parts;
        // fixed summation order for every tile shape: four interleaved accumulators a_j = sum of parts j, j+4, j+8, ...,
        // total = (a0 + a1) + (a2 + a3).  A thread owns the accumulators j = sub, sub + TPR, ...
        constexpr int NA = 4 / TPR;
        static_assert(TPR == 2 || TPR == 4, "two or four threads per tile row");
        float a0[NA], a1[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) a0[j] = a1[j] = 0.f;
        // six rounds of loads (24 parts = 768 channels) in flight at once: a second, dependent batch of loads would put
        // another memory round trip in front of every block's K loop
        constexpr int ROUNDS = 6;
        for (int pp = 0; pp < p.ln_parts; pp += 4 * ROUNDS) {
          float2 v[ROUNDS][NA];
#pragma unroll
          for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              const int q = pp + 4 * rr + sub + j * TPR;
              v[rr][j] = q < p.ln_parts ? sp[q] : make_float2(0.f, 0.f);
            }
#pragma unroll
          for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
            for (int j = 0; j < NA; ++j) { a0[j] += v[rr][j].x; a1[j] += v[rr][j].y; }
        }
        // lanes sub ^ 1 hold the odd accumulators: (a0 + a1), (a2 + a3), then their sum
        float s0, s1;
        if constexpr (TPR == 4) {
          s0 = a0[0] + __shfl_xor(a0[0], 1); s1 = a1[0] + __shfl_xor(a1[0], 1);
          s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
        } else {
          const float b0 = a0[0] + __shfl_xor(a0[0], 1), b1 = a1[0] + __shfl_xor(a1[0], 1);   // a0 + a1
          const float c0 = a0[1] + __shfl_xor(a0[1], 1), c1 = a1[1] + __shfl_xor(a1[1], 1);   // a2 + a3
          s0 = b0 + c0; s1 = b1 + c1;
        }
        if (sub == 0) {
          // finished in double: E[x^2] - mean^2 cancels for rows whose |mean| is far above their spread
          const double inv = 1.0 / (double)p.K;
          const double mean = (double)s0 * inv;
          const double var = fmax((double)s1 * inv - mean * mean, 0.0);
          ((float2*)lnside)[lrow] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.ln_eps)));
        }
      }
      for (int c = t; c < BN; c += NTH) lnside[2 * BM + c] = bcol + c < p.N ? p.ln_u[bcol + c] : 0.f;
    }
  }
  IG_STAMP(1);
  IGEMM_WAIT_STAGE(NSTAGE - 1);  // tile kt_begin has landed, NSTAGE-1 younger ones are in flight
  IG_STAMP(2);
  IGEMM_READ_FRAGS(af0, wf0, 0, 0);
  int slot = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
    IGEMM_READ_FRAGS(af1, wf1, slot, 1);
    IGEMM_MFMA_ALL(af0, wf0);
    IGEMM_WAIT_STAGE(NSTAGE - 2);
    IGEMM_STAGE_TILE(kt + NSTAGE, slot);
    IGEMM_READ_FRAGS(af0, wf0, nslot, 0);
    IGEMM_MFMA_ALL(af1, wf1);
    slot = nslot;
  }
  // this wave's trailing (zero-page) DMAs must land before it exits and its LDS is handed to another block
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef IGEMM_READ_FRAGS
#undef IGEMM_MFMA_ALL
#undef IGEMM_STAGE_TILE
#undef IGEMM_WAIT_STAGE
#undef IGEMM_GLDS

  IG_STAMP(3);
  igemm_finish<TM, TN, GEGLU, !CONV, true>(p, acc, er, fast, smem, bz, brow + wr * TM, bcol + wc * TN, (lds_cfp)lnside, BM, wr * TM, wc * TN,
                              (brow + wr * TM) / 32, (bcol + wc * TN) / 32);
#ifdef IGEMM_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IG_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------ persistent plain GEMM
// igemm_kernel pays its fixed part - block launch, argument loads, pointer set-up, the first K-tile's memory round trip,
// the epilogue's stores - once per output tile; on the short-K GEMMs of the transformer blocks (K = 320: five K-tiles)
// that is half a block's life, and a launch of several rounds of blocks (the GEGLU projection: 1440 tiles on 512 slots)
// pays it every round.  Here a block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the staging ring does not
// stop at a tile's end: while tile i's last K-tiles are multiplied and its epilogue runs, the first NSTAGE K-tiles of
// tile i+1 are already on their way.  Same K order, same epilogue: same bits as igemm_kernel.  No split-K, no conv.
template <int BM, int BN, int WM, int WN, int NSTAGE, bool GEGLU>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && NSTAGE * (BM + BN) * BK * 2 <= 80 * 1024 ? 2 : 1)) void igemm_persist_kernel(const IgemmParams p) {
  constexpr int NTH = 64 * WM * WN, RP = NTH / 8;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16, PA = BM / RP, PB = BN / RP;
  static_assert(BM % RP == 0 && BN % RP == 0 && TM % 16 == 0 && TN % 16 == 0, "tile / wave grid mismatch");
  constexpr int STAGE = (BM + BN) * BK;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv / WN, wc = wv % WN, l15 = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, ntile = tiles_m * tiles_n;
  const int bz = blockIdx.y;
  const u16* A = p.a + (long long)bz * p.sa;
  const u16* Wp = p.w + (long long)bz * p.sw;
  const u16* const zp = (const u16*)g_zero16;
  const int r0 = t >> 3;
  const int lchunk = (t & 7) ^ ((r0 >> 1) & 7);
  const int wave_row0 = __builtin_amdgcn_readfirstlane(wv) * 8;
  const int k_lim = p.K - lchunk * 8;
  const int nk = p.nk;

  // ---- staging cursor: the tile whose K-tiles are being fetched (runs NSTAGE K-tiles ahead of the multiplication) ----
  const u16* a_ptr[PA];
  bool a_ok[PA];
  const u16* b_ptr[PB];
  bool b_ok[PB];
  int st_tile = blockIdx.x, s_k = 0;
  bool s_live = st_tile < ntile;
  auto set_ptrs = [&](int tile) {
    const int bid = xcd_remap(tile, ntile);
    const int brow = (p.nmajor ? bid % tiles_m : bid / tiles_n) * BM, bcol = (p.nmajor ? bid / tiles_m : bid % tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int gm = brow + r0 + RP * i;
      a_ok[i] = gm < p.M;
      a_ptr[i] = A + (long long)(a_ok[i] ? gm : 0) * p.lda + lchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int gn = bcol + r0 + RP * i;
      b_ok[i] = gn < p.N;
      b_ptr[i] = Wp + (long long)(b_ok[i] ? gn : 0) * p.ldw + lchunk * 8;
    }
  };
  if (s_live) set_ptrs(st_tile);
#define PGEMM_GLDS(SRC, DST) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
  // always PA + PB DMA instructions per wave (the zero page past the last tile), so the counted waits are constants
#define PGEMM_STAGE_NEXT(SLOT)                                                                \
  {                                                                                           \
    if (s_k == nk) {                                                                          \
      st_tile += gridDim.x;                                                                   \
      s_live = st_tile < ntile;                                                               \
      if (s_live) set_ptrs(st_tile);                                                          \
      s_k = 0;                                                                                \
    }                                                                                         \
    u16* dA = smem + (SLOT) * STAGE + wave_row0 * BK;                                         \
    u16* dB = dA + BM * BK;                                                                   \
    const int ko = s_k * BK;                                                                  \
    const bool kok = s_live && ko < k_lim;                                                    \
    _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                          \
      const u16* src = (a_ok[i] && kok) ? a_ptr[i] + ko : zp;                                 \
      PGEMM_GLDS(src, dA + i * RP * BK);                                                      \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                          \
      const u16* src = (b_ok[i] && kok) ? b_ptr[i] + ko : zp;                                 \
      PGEMM_GLDS(src, dB + i * RP * BK);                                                      \
    }                                                                                         \
    ++s_k;                                                                                    \
  }
#define PGEMM_WAIT_STAGE(GROUPS)                                                                    \
  {                                                                                                 \
    constexpr int INFLIGHT = (GROUPS) * (PA + PB);                                                  \
    static_assert(INFLIGHT >= 0 && INFLIGHT < 64, "vmcnt is a 6-bit counter");                     \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");                      \
    __builtin_amdgcn_s_barrier();                                                                   \
  }
#define PGEMM_READ_FRAGS(AF, WF, SLOT, KK)                                                                       \
  {                                                                                                              \
    const u16* cA = smem + (SLOT) * STAGE;                                                                       \
    const u16* cB = cA + BM * BK;                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AF[i] = *(const hx8*)&cA[lds_off(wr * TM + i * 16 + l15, lq + 4 * (KK))]; \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const hx8*)&cB[lds_off(wc * TN + j * 16 + l15, lq + 4 * (KK))]; \
  }
#define PGEMM_MFMA_ALL(AF, WF)                                                                 \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(WF[j], AF[i], acc[i][j], 0, 0, 0); \
  }
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE; ++sidx) PGEMM_STAGE_NEXT(sidx);

  constexpr bool FAST_OK = !GEGLU || TN >= 64;
  const bool fast = FAST_OK && p.fast_epi;
  float* const lnside = (float*)(smem + NSTAGE * STAGE);
  int slot = 0;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int bid = xcd_remap(tile, ntile);
    const int brow = (p.nmajor ? bid % tiles_m : bid / tiles_n) * BM, bcol = (p.nmajor ? bid / tiles_m : bid % tiles_n) * BN;
    if (tile != (int)blockIdx.x) {
      // the previous tile's stores are out and its reads of the LayerNorm side region are done before anything of this tile
      // is counted or written: loads and stores do not retire in one order, so the counted waits below must not see stores
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    hx8 af0[MT], wf0[NT], af1[MT], wf1[NT];
    EpiRegs<TM, FAST_OK ? TN : 64, GEGLU> er;
    if constexpr (FAST_OK) if (fast) epi_prefetch<TM, TN, GEGLU>(p, er, bz, brow + wr * TM, bcol + wc * TN, lane);
    if (p.ln_stats) {
      {
        constexpr int TPR = NTH / BM;
        const int lrow = t / TPR, sub = t % TPR;
        const int row = min(brow + lrow, p.M - 1);
        const float2* sp = (const float2*)p.ln_stats + (long long)row * p.ln_parts;
        constexpr int NA = 4 / TPR;
        static_assert(TPR == 2 || TPR == 4, "two or four threads per tile row");
        float a0[NA], a1[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) a0[j] = a1[j] = 0.f;
        constexpr int ROUNDS = 6;
        for (int pp = 0; pp < p.ln_parts; pp += 4 * ROUNDS) {
          float2 v[ROUNDS][NA];
#pragma unroll
          for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              const int q = pp + 4 * rr + sub + j * TPR;
              v[rr][j] = q < p.ln_parts ? sp[q] : make_float2(0.f, 0.f);
            }
#pragma unroll
          for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
            for (int j = 0; j < NA; ++j) { a0[j] += v[rr][j].x; a1[j] += v[rr][j].y; }
        }
        float s0, s1;
        if constexpr (TPR == 4) {
          s0 = a0[0] + __shfl_xor(a0[0], 1); s1 = a1[0] + __shfl_xor(a1[0], 1);
          s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
        } else {
          const float b0 = a0[0] + __shfl_xor(a0[0], 1), b1 = a1[0] + __shfl_xor(a1[0], 1);
          const float c0 = a0[1] + __shfl_xor(a0[1], 1), c1 = a1[1] + __shfl_xor(a1[1], 1);
          s0 = b0 + c0; s1 = b1 + c1;
        }
        if (sub == 0) {
          // finished in double: E[x^2] - mean^2 cancels for rows whose |mean| is far above their spread
          const double inv = 1.0 / (double)p.K;
          const double mean = (double)s0 * inv;
          const double var = fmax((double)s1 * inv - mean * mean, 0.0);
          ((float2*)lnside)[lrow] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.ln_eps)));
        }
      }
      for (int c = t; c < BN; c += NTH) lnside[2 * BM + c] = bcol + c < p.N ? p.ln_u[bcol + c] : 0.f;
    }
    PGEMM_WAIT_STAGE(NSTAGE - 1);
    PGEMM_READ_FRAGS(af0, wf0, slot, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
      PGEMM_READ_FRAGS(af1, wf1, slot, 1);
      PGEMM_MFMA_ALL(af0, wf0);
      PGEMM_WAIT_STAGE(NSTAGE - 2);
      PGEMM_STAGE_NEXT(slot);
      PGEMM_READ_FRAGS(af0, wf0, nslot, 0);
      PGEMM_MFMA_ALL(af1, wf1);
      slot = nslot;
    }
    igemm_finish<TM, TN, GEGLU, true, true>(p, acc, er, fast, smem, bz, brow + wr * TM, bcol + wc * TN, (lds_cfp)lnside, BM, wr * TM, wc * TN,
                                            (brow + wr * TM) / 32, (bcol + wc * TN) / 32);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef PGEMM_GLDS
#undef PGEMM_STAGE_NEXT
#undef PGEMM_WAIT_STAGE
#undef PGEMM_READ_FRAGS
#undef PGEMM_MFMA_ALL
}

// ------------------------------------------------------------------------------------------ A-stationary GEMM (round 3)
// The K = 320 GEMMs of the 96x96 transformer blocks (to_q|k|v: N = 960, GEGLU proj: N = 2560; M = 9216 per CFG half) are
// bound by the L2 -> LDS path: a 128 x 128 tile stages (128 + 128) x 320 x 2 B for 5 K-tiles of work, 236 MB per GEGLU launch
// at the ~7.7 TB/s every LDS-staged kernel here tops out at.  Half of that is the activation panel,
// staged again by each of the row tile's 20 column tiles.  Here a block keeps its 128 x K activation panel RESIDENT in the
// LDS (80 KB at K = 320) and walks a run of column tiles with it; only the weight tiles stream (a three-slot ring that runs
// on across column tiles, so the next tile's first K-tiles land under the current tile's epilogue).  Grid = row tiles x
// nsplit column runs, about one block per CU (130 KB of LDS); ids are run-major, so the blocks an XCD holds share their
// weight columns.  Same K order, same arithmetic, same epilogue (igemm_finish) as igemm_kernel: same bits (tested).
// Needs K % 64 = 0, K <= 384, the 16-byte epilogue, batch 1, no split-K.
template <int BN, bool GEGLU>
__global__ __launch_bounds__(256, 1) void igemm_astat_kernel(const IgemmParams p, int nsplit) {
  constexpr int BM = 128, WN = 2, NSTAGE = 3, NTH = 256;
  constexpr int TM = 64, TN = BN / WN, MT = TM / 16, NT = TN / 16, PA = BM / 32, PB = BN / 32;
  constexpr int ASLAB = BM * BK, BSLOT = BN * BK;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // [nk][BM x BK] panel | [NSTAGE][BN x BK] weight ring | LayerNorm side
  const int nk = p.nk;
  u16* const bring = smem + nk * ASLAB;
  float* const lnside = (float*)(bring + NSTAGE * BSLOT);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv / WN, wc = wv % WN, l15 = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * nsplit);
  const int part = bid / tiles_m, brow = (bid % tiles_m) * BM;
  const int c0 = (int)((long long)part * tiles_n / nsplit), c1 = (int)((long long)(part + 1) * tiles_n / nsplit);
  const int r0 = t >> 3;
  const int lchunk = (t & 7) ^ ((r0 >> 1) & 7);
  const int wave_row0 = __builtin_amdgcn_readfirstlane(wv) * 8;
  const unsigned lds0 = lds_addr_of(smem);

  // descriptors end with the last row of the operand: rows of the tile past M / columns past N arrive as zeros
  const int rows_here = min(BM, p.M - brow);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (long long)brow * p.lda), 0, (unsigned)((rows_here - 1) * p.lda + p.K) * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)(((long long)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcZ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
  unsigned a_voff[PA], b_voff[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) a_voff[i] = (unsigned)(((r0 + 32 * i) * p.lda + lchunk * 8) * 2);
#pragma unroll
  for (int i = 0; i < PB; ++i) b_voff[i] = (unsigned)(((r0 + 32 * i) * p.ldw + lchunk * 8) * 2);

  // the whole activation panel, then the first weight tiles
  for (int kt = 0; kt < nk; ++kt)
#pragma unroll
    for (int i = 0; i < PA; ++i) dma_buf16(rsrcA, a_voff[i], __builtin_amdgcn_readfirstlane(kt * (BK * 2)), lds0 + (kt * ASLAB + (wave_row0 + 32 * i) * BK) * 2);
  int s_ct = c0, s_k = 0;  // staging cursor: runs NSTAGE K-tiles ahead of the multiplication, across column tiles
#define AGEMM_STAGE_NEXT(SLOT)                                                                                     \
  {                                                                                                                \
    const bool live_ = s_ct < c1;                                                                                  \
    const int soff_ = __builtin_amdgcn_readfirstlane(live_ ? (s_ct * BN * p.ldw + s_k * BK) * 2 : 0);              \
    _Pragma("unroll") for (int i = 0; i < PB; ++i)                                                                 \
      dma_buf16(live_ ? rsrcB : rsrcZ, b_voff[i], soff_, lds0 + (nk * ASLAB + (SLOT) * BSLOT + (wave_row0 + 32 * i) * BK) * 2); \
    if (++s_k == nk) { s_k = 0; ++s_ct; }                                                                          \
  }
#define AGEMM_WAIT_STAGE(GROUPS)                                                   \
  {                                                                                \
    constexpr int INFLIGHT = (GROUPS) * PB;                                        \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");     \
    __builtin_amdgcn_s_barrier();                                                  \
  }
#define AGEMM_READ_FRAGS(AF, WF, KT, SLOT, KK)                                                                      \
  {                                                                                                                 \
    const u16* cA = smem + (KT) * ASLAB;                                                                            \
    const u16* cB = bring + (SLOT) * BSLOT;                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AF[i] = *(const hx8*)&cA[lds_off(wr * TM + i * 16 + l15, lq + 4 * (KK))]; \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const hx8*)&cB[lds_off(wc * TN + j * 16 + l15, lq + 4 * (KK))]; \
  }
#define AGEMM_MFMA_ALL(AF, WF)                                                                 \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(WF[j], AF[i], acc[i][j], 0, 0, 0); \
  }
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE; ++sidx) AGEMM_STAGE_NEXT(sidx);

  // LayerNorm fold: (mean, rstd) of the tile's rows, once per block (as igemm_kernel: fixed order, finished in double)
  if (p.ln_stats) {
    constexpr int TPR = NTH / BM;  // 2 threads per row
    const int lrow = t / TPR, sub = t % TPR;
    const int row = min(brow + lrow, p.M - 1);
    const float2* sp = (const float2*)p.ln_stats + (long long)row * p.ln_parts;
    constexpr int NA = 4 / TPR;
    float a0[NA], a1[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) a0[j] = a1[j] = 0.f;
    constexpr int ROUNDS = 6;
    for (int pp = 0; pp < p.ln_parts; pp += 4 * ROUNDS) {
      float2 v[ROUNDS][NA];
#pragma unroll
      for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int q = pp + 4 * rr + sub + j * TPR;
          v[rr][j] = q < p.ln_parts ? sp[q] : make_float2(0.f, 0.f);
        }
#pragma unroll
      for (int rr = 0; rr < ROUNDS; ++rr)
#pragma unroll
        for (int j = 0; j < NA; ++j) { a0[j] += v[rr][j].x; a1[j] += v[rr][j].y; }
    }
    const float b0 = a0[0] + __shfl_xor(a0[0], 1), b1 = a1[0] + __shfl_xor(a1[0], 1);
    const float d0 = a0[1] + __shfl_xor(a0[1], 1), d1 = a1[1] + __shfl_xor(a1[1], 1);
    const float s0 = b0 + d0, s1 = b1 + d1;
    if (sub == 0) {
      const double inv = 1.0 / (double)p.K;
      const double mean = (double)s0 * inv;
      const double var = fmax((double)s1 * inv - mean * mean, 0.0);
      ((float2*)lnside)[lrow] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.ln_eps)));
    }
  }

  int slot = 0;
  for (int ct = c0; ct < c1; ++ct) {
    const int bcol = ct * BN;
    if (ct != c0) {
      // the previous tile's stores are out and its reads of the side region are done before anything of this tile is
      // counted or written (loads and stores do not retire in one order: the counted waits below must not see stores)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    hx8 af0[MT], wf0[NT], af1[MT], wf1[NT];
    EpiRegs<TM, TN, GEGLU> er;
    epi_prefetch<TM, TN, GEGLU>(p, er, 0, brow + wr * TM, bcol + wc * TN, lane);
    if (p.ln_stats)
      for (int c = t; c < BN; c += NTH) lnside[2 * BM + c] = bcol + c < p.N ? p.ln_u[bcol + c] : 0.f;
    AGEMM_WAIT_STAGE(NSTAGE - 1);
    AGEMM_READ_FRAGS(af0, wf0, 0, slot, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
      const int nkt = kt + 1 == nk ? 0 : kt + 1;
      AGEMM_READ_FRAGS(af1, wf1, kt, slot, 1);
      AGEMM_MFMA_ALL(af0, wf0);
      AGEMM_WAIT_STAGE(NSTAGE - 2);
      AGEMM_STAGE_NEXT(slot);
      AGEMM_READ_FRAGS(af0, wf0, nkt, nslot, 0);
      AGEMM_MFMA_ALL(af1, wf1);
      slot = nslot;
    }
    igemm_finish<TM, TN, GEGLU, true, true>(p, acc, er, true, smem, 0, brow + wr * TM, bcol + wc * TN, (lds_cfp)lnside, BM, wr * TM, wc * TN,
                                            (brow + wr * TM) / 32, (bcol + wc * TN) / 32);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef AGEMM_STAGE_NEXT
#undef AGEMM_WAIT_STAGE
#undef AGEMM_READ_FRAGS
#undef AGEMM_MFMA_ALL
}

// ------------------------------------------------------------------------------------------ shared-window 3x3 conv
// Stride-1, pad-1 convolutions.  The generic kernel stages a fresh 128 x 64 activation tile for each of the 9 taps; the
// three taps of one kernel row (kx = 0, 1, 2) read the SAME pixels shifted by one, so here ONE window of BM + 2 pixel
// rows is staged per (64-channel chunk, ky) and the three kx taps read it at row offsets 0 / 1 / 2: the activation
// bytes that go through the L2 -> LDS path (what bounds these kernels, see the file header) drop 3x, all staged bytes
// 1.5x.  Vertical padding is decided when a window row is staged (zero page), horizontal padding when a fragment is
// read: lanes whose (pixel, kx) falls off the image row read a row of zeros kept behind each window buffer (the offset
// is precomputed per lane, no work in the loop).  K-tile order (chunk, ky, kx) and all arithmetic are those of the
// generic kernel, so both produce the same bits.  4 waves (2 x 2), two window buffers + a two-slot weight ring:
// 66 KB (BN = 128) or 50 KB (BN = 64) of LDS, 2 - 3 blocks per CU.
// CS: the epilogue can emit per-channel statistics (colstats); a separate instantiation because the 128 x 128 tile sits at
// the 256-register budget and the statistics' accumulators push it into spills.
template <int BM, int BN, bool CS>
__global__ __launch_bounds__(256, 2) void conv_win_kernel(const IgemmParams p) {
  constexpr int TM = BM / 2, TN = BN / 2, MT = TM / 16, NT = TN / 16;
  constexpr int WROWS = BM + 8;           // staged window rows: BM + 2 are read; whole 8-row DMA instructions
  constexpr int WBUF = (WROWS + 1) * BK;  // elements per window buffer, + one row of zeros
  constexpr int BSLOT = BN * BK;
  constexpr int PW = (WROWS + 31) / 32, PB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // [2][WBUF] windows | [2][BSLOT] weight ring
  u16* const bring = smem + 2 * WBUF;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1, l15 = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int brow = (p.nmajor ? bid % tiles_m : bid / tiles_n) * BM, bcol = (p.nmajor ? bid / tiles_m : bid % tiles_n) * BN;
  const u16* const zp = (const u16*)g_zero16;
  const int hw = p.H * p.W;

  if (t < 16) *(u32x4*)(smem + (t >> 3) * WBUF + WROWS * BK + (t & 7) * 8) = u32x4{0u, 0u, 0u, 0u};  // the zero rows

  // ---- staging bookkeeping (fixed over the K loop) ----
  const int r0 = t >> 3;
  const int lchunk = (t & 7) ^ ((r0 >> 1) & 7);  // logical 16-B chunk this lane fetches (swizzle on the source side)
  const int wave_row0 = __builtin_amdgcn_readfirstlane(wv) * 8;
  const u16* w_ptr[PW];
  int w_mask[PW];  // bit ky: window row readable for kernel row ky
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = r0 + 32 * i, q = brow - 1 + r;  // window row r holds pixel q (flat N*H*W index)
    const bool ok = r < WROWS && q >= 0 && q < p.M;
    const int qc = ok ? q : 0;
    const int img = qc / hw, y = (qc - img * hw) / p.W;
    w_mask[i] = ok ? ((y >= 1 ? 1 : 0) | 2 | (y <= p.H - 2 ? 4 : 0)) : 0;
    w_ptr[i] = p.a + (long long)qc * p.lda + lchunk * 8;
  }
  bool b_ok[PB];
  const u16* b_ptr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int gn = bcol + r0 + 32 * i;
    b_ok[i] = gn < p.N;
    b_ptr[i] = p.w + (long long)(b_ok[i] ? gn : 0) * p.ldw + lchunk * 8;
  }
  const int k_lim = p.Cin - lchunk * 8;  // this lane's chunk lies inside Cin iff (chunk base channel) < k_lim

  // ---- fragment read offsets: window row = tile row + kx; off-row taps read the zero row ----
  int aoff[MT][3];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int i = wr * TM + mi * 16 + l15, pix = brow + i;
    const bool pv = pix < p.M;
    const int pc = pv ? pix : 0;
    const int x = (pc % hw) % p.W;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const bool valid = pv && x + kx - 1 >= 0 && x + kx - 1 < p.W;
      aoff[mi][kx] = valid ? lds_off(i + kx, lq) : WROWS * BK + lq * 8;
    }
  }
  int boff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) boff[j] = lds_off(wc * TN + j * 16 + l15, lq);

  const int kt_begin = blockIdx.z * p.kt_per_split;          // multiples of 3 (host)
  const int kt_end = min(p.nk, kt_begin + p.kt_per_split);
  int s_cb = kt_begin / 9, s_tap = kt_begin - s_cb * 9, s_kt = kt_begin;  // next weight tile to stage
  int w_cb = s_cb, w_ky = s_tap / 3;                                        // next window to stage

#define WIN_GLDS(SRC, DST) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
#define WIN_STAGE_W(WB)                                                                                   \
  {                                                                                                       \
    const int c0 = w_cb * BK;                                                                             \
    const bool cok = c0 < k_lim;                                                                          \
    const long long koff = (long long)(w_ky - 1) * p.W * p.lda + c0;                                      \
    _Pragma("unroll") for (int i = 0; i < PW; ++i) {                                                      \
      if (wave_row0 + 32 * i < WROWS) {                                                                   \
        const u16* src = (((w_mask[i] >> w_ky) & 1) && cok) ? w_ptr[i] + koff : zp;                       \
        WIN_GLDS(src, (WB) + (wave_row0 + 32 * i) * BK);                                                  \
      }                                                                                                   \
    }                                                                                                     \
    if (++w_ky == 3) { w_ky = 0; ++w_cb; }                                                                \
  }
#define WIN_STAGE_B(SLOT)                                                                                 \
  if (s_kt < kt_end) {                                                                                    \
    const int c0 = s_cb * BK;                                                                             \
    const bool cok = c0 < k_lim;                                                                          \
    const int wk = s_tap * p.Cin + c0;                                                                    \
    u16* dB = bring + (SLOT) * BSLOT + wave_row0 * BK;                                                    \
    _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                      \
      const u16* src = (b_ok[i] && cok) ? b_ptr[i] + wk : zp;                                             \
      WIN_GLDS(src, dB + i * 32 * BK);                                                                    \
    }                                                                                                     \
    ++s_kt;                                                                                               \
    if (++s_tap == 9) { s_tap = 0; ++s_cb; }                                                              \
  }
#define WIN_WAIT()                                                \
  {                                                               \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                                 \
  }
#define WIN_READ(AF, WF, WB, KX, KK, SLOT)                                                                          \
  {                                                                                                                 \
    const u16* cB = bring + (SLOT) * BSLOT;                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AF[i] = *(const hx8*)&(WB)[aoff[i][KX] ^ ((KK) ? 32 : 0)];       \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const hx8*)&cB[boff[j] ^ ((KK) ? 32 : 0)];             \
  }
#define WIN_MFMA(AF, WF)                                                                                   \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                         \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(WF[j], AF[i], acc[i][j], 0, 0, 0); \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  hx8 af0[MT], wf0[NT], af1[MT], wf1[NT];
  EpiRegs<TM, TN, false> er;
  const bool fast = p.fast_epi;
  if (fast && p.splits == 1) epi_prefetch<TM, TN, false>(p, er, 0, brow + wr * TM, bcol + wc * TN, lane);

  WIN_STAGE_W(smem);
  WIN_STAGE_B(0);
  WIN_STAGE_B(1);
  WIN_WAIT();
  WIN_READ(af0, wf0, smem, 0, 0, 0);
  // One (chunk, ky) group = three K-tiles (kx = 0, 1, 2) read from window WCUR, weight slots S0, S1, S0; the next
  // group's window is staged into WNEXT during the kx = 0 step.  The loop alternates two instantiations (windows swapped,
  // slots S0/S1 swapped) so that every LDS address is a per-lane base plus a compile-time constant.
#define WIN_GROUP(WCUR, WNEXT, S0, S1)                                                         \
  {                                                                                            \
    /* kx = 0 */                                                                               \
    WIN_READ(af1, wf1, WCUR, 0, 1, S0);                                                        \
    WIN_MFMA(af0, wf0);                                                                        \
    WIN_WAIT();                                                                                \
    WIN_STAGE_B(S0);                                                                           \
    if (kt + 3 < kt_end) WIN_STAGE_W(WNEXT);                                                   \
    WIN_READ(af0, wf0, WCUR, 1, 0, S1);                                                        \
    WIN_MFMA(af1, wf1);                                                                        \
    /* kx = 1 */                                                                               \
    WIN_READ(af1, wf1, WCUR, 1, 1, S1);                                                        \
    WIN_MFMA(af0, wf0);                                                                        \
    WIN_WAIT();                                                                                \
    WIN_STAGE_B(S1);                                                                           \
    WIN_READ(af0, wf0, WCUR, 2, 0, S0);                                                        \
    WIN_MFMA(af1, wf1);                                                                        \
    /* kx = 2 */                                                                               \
    WIN_READ(af1, wf1, WCUR, 2, 1, S0);                                                        \
    WIN_MFMA(af0, wf0);                                                                        \
    WIN_WAIT();                                                                                \
    WIN_STAGE_B(S0);                                                                           \
    WIN_READ(af0, wf0, WNEXT, 0, 0, S1); /* first tile of the next group; unused after the last one */ \
    WIN_MFMA(af1, wf1);                                                                        \
  }
  for (int kt = kt_begin; kt < kt_end; kt += 3) {
    WIN_GROUP(smem, smem + WBUF, 0, 1);
    kt += 3;
    if (kt >= kt_end) break;
    WIN_GROUP(smem + WBUF, smem, 1, 0);
  }
#undef WIN_GROUP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef WIN_GLDS
#undef WIN_STAGE_W
#undef WIN_STAGE_B
#undef WIN_WAIT
#undef WIN_READ
#undef WIN_MFMA
  igemm_finish<TM, TN, false, false, CS>(p, acc, er, fast, smem, 0, brow + wr * TM, bcol + wc * TN, (lds_cfp)smem, 0, 0, 0, (brow + wr * TM) / 32,
                              (bcol + wc * TN) / 32);
}

// ------------------------------------------------------------------------------------------ shared-window 3x3 conv, round 3
// conv_win_kernel with the two things its counters asked for (profiles/r02_conv_pmc_*: 3.2 vector + 2.6 scalar instructions
// per MFMA, 30 % of the LDS cycles bank conflicts):
//  - staging through BUFFER descriptors (buffer_load_dwordx4 ... offen lds): a lane's byte offset inside the block's window /
//    weight panel is computed ONCE, the K position (kernel row, channel chunk, tap) is a scalar offset, rows that must read
//    zeros (vertical padding, rows past M or N) carry an offset beyond num_records and land as zeros.  No 64-bit pointer
//    arithmetic, no zero-page select in the loop: one v_cndmask per window DMA, nothing per weight DMA.
//  - an LDS swizzle that stays conflict-free when the window is read at row offsets 0 / 1 / 2: a ds_read_b128 lane group
//    holds rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c ^ 1 (MI355X_MICROARCH LDS table); XOR-ing chunk bits 1..2
//    with bits 1..2 of the row never maps a c-lane onto a (c ^ 1)-lane, and rows of equal parity inside either set differ in
//    (row >> 1) & 3 whatever the shift.  (The generic swizzle (row >> 1) & 7 touches bit 0 and collides at shifts 1 and 2.)
// Same K order, same arithmetic, same epilogue as conv_win_kernel and igemm_kernel: same bits (tested).
// Needs Cin % 64 == 0 and panels below 2 GB (host); everything else keeps conv_win_kernel.
__device__ __forceinline__ int lds_off_w(int row, int chunk) { return row * BK + ((chunk ^ (((row >> 1) & 3) << 1)) << 3); }
// (Tried and dropped, profiles/r03_conv_win2_ab.txt: scheduling fences that keep a step's fragment reads in front of the
// previous step's MFMAs, and s_setprio around the MFMA batches - 178 VGPRs instead of 154, 1.5 ... 4 % slower.)
// GNA (north-star item (b), the bounded case; sdmi_epilogue_ext.gn_in): the input is the RAW tensor and a per-(sample, channel)
// scale / shift table; every staged window is turned into SiLU(x * scale + shift) in place before the MFMAs read it - by the
// thread that staged the 16-byte piece (it waits for its own DMA, transforms, and the K loop's barrier publishes the result;
// no extra barrier), padding rows (zero-filled by the descriptor) left alone.  Same expression as gn_apply_kernel
// (fma, silu_f, saturating pack): the LDS holds the same bits the unfused pair would have staged, so the result is
// bit-identical to sdmi_groupnorm_apply + this kernel.  One column block only (Cout <= BN, host): a second column block
// would transform the same windows again.
// UPS (round 3): the nearest-2x upsample convs (layers/ResBlock.py:17-19) on the same kernel.  The window holds pixels of the
// VIRTUAL upsampled image (2H x 2W); only the staging address changes - window row r for kernel row ky comes from source pixel
// ((y + ky - 1) >> 1, x >> 1), a per-lane offset per ky computed once (the ky shift is no longer one scalar: it depends on the
// parity of y), K position = the chunk's scalar offset.  Fragment reads, K order, epilogue: unchanged, so the same bits as the
// generic kernel's gather loader with a third of its activation bytes staged.
template <int BM, int BN, bool CS, bool GNA = false, bool UPS = false>
__global__ __launch_bounds__(256, BM == 256 ? 1 : (BN == 64 ? 3 : 2)) void conv_win2_kernel(const IgemmParams p) {
  static_assert(!(GNA && UPS), "gn_in is not built for the upsample variant");
  constexpr int TM = BM / 2, TN = BN / 2, MT = TM / 16, NT = TN / 16;
  constexpr int WROWS = BM + 8;           // staged window rows: BM + 2 are read; whole 8-row DMA instructions
  constexpr int WBUF = (WROWS + 1) * BK;  // elements per window buffer, + one row of zeros
  constexpr int BSLOT = BN * BK;
  constexpr int PW = (WROWS + 31) / 32, PB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // [2][WBUF] windows | [2][BSLOT] weight ring
  u16* const bring = smem + 2 * WBUF;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1, l15 = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int brow = (p.nmajor ? bid % tiles_m : bid / tiles_n) * BM, bcol = (p.nmajor ? bid / tiles_m : bid % tiles_n) * BN;
  const int VH = UPS ? 2 * p.H : p.H, VW = UPS ? 2 * p.W : p.W;  // the image the 3x3 taps walk (upsampled: virtual)
  const int hw = VH * VW;

  if (t < 16) *(u32x4*)(smem + (t >> 3) * WBUF + WROWS * BK + (t & 7) * 8) = u32x4{0u, 0u, 0u, 0u};  // the zero rows

  // ---- staging bookkeeping (fixed over the K loop) ----
  const int r0 = t >> 3;
  const int lchunk = (t & 7) ^ (((r0 >> 1) & 3) << 1);  // logical 16-B chunk this lane fetches (swizzle on the source side)
  const int wave_row0 = __builtin_amdgcn_readfirstlane(wv) * 8;
  // window: descriptor base = pixel (brow - 1 - W), the first pixel a kernel row 0 read of window row 0 touches; window row r
  // for kernel row ky then sits at ((r + ky W) lda + chunk) elements.  The base may lie in front of the tensor: lanes whose
  // pixel does not exist are masked (offset DMA_OOB), never fetched.
  // (UPS: the descriptor starts at the tensor and the per-lane offsets are absolute: the host keeps the tensor below 2 GB)
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(UPS ? p.a : p.a + ((long long)brow - 1 - p.W) * p.lda), 0, DMA_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long long)bcol * p.ldw), 0, DMA_OOB, 0x00020000);
  unsigned w_voff[PW];
  int w_mask[PW];  // bit ky: window row readable for kernel row ky.  UPS: bit 4 / bits 5-6 = source rows to add for ky = 1 / 2
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = r0 + 32 * i, q = brow - 1 + r;  // window row r holds pixel q (flat N*H*W index)
    const bool ok = r < BM + 2 && q >= 0 && q < p.M;  // rows BM + 2 ... of the last 8-row DMA group are never read: not fetched
    const int qc = ok ? q : 0;
    const int img = qc / hw, y = (qc - img * hw) / VW;
    w_mask[i] = ok ? ((y >= 1 ? 1 : 0) | 2 | (y <= VH - 2 ? 4 : 0)) : 0;
    if constexpr (UPS) {
      const int x = qc - img * hw - y * VW;
      const int src0 = img * (p.H * p.W) + (x >> 1);  // + source row * W
      const int ym = y >= 1 ? y - 1 : 0, yp = y <= VH - 2 ? y + 1 : y;
      w_voff[i] = (unsigned)(((src0 + (ym >> 1) * p.W) * p.lda + lchunk * 8) * 2);  // kernel row 0; rows 1 / 2 lie 0 - 2 source rows below
      w_mask[i] |= (((y >> 1) - (ym >> 1)) << 4) | (((yp >> 1) - (ym >> 1)) << 5);
    } else {
      w_voff[i] = (unsigned)((r * p.lda + lchunk * 8) * 2);
    }
  }
  unsigned b_voff[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int rn = r0 + 32 * i;
    b_voff[i] = bcol + rn < p.N ? (unsigned)((rn * p.ldw + lchunk * 8) * 2) : DMA_OOB;
  }
  const unsigned lds0 = lds_addr_of(smem);

  // ---- fragment read offsets: window row = tile row + kx; off-row taps read the zero row ----
  int aoff[MT][3];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int i = wr * TM + mi * 16 + l15, pix = brow + i;
    const bool pv = pix < p.M;
    const int pc = pv ? pix : 0;
    const int x = (pc % hw) % VW;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const bool valid = pv && x + kx - 1 >= 0 && x + kx - 1 < VW;
      aoff[mi][kx] = valid ? lds_off_w(i + kx, lq) : WROWS * BK + lq * 8;
    }
  }
  int boff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) boff[j] = lds_off_w(wc * TN + j * 16 + l15, lq);

  const int kt_begin = blockIdx.z * p.kt_per_split;          // multiples of 3 (host)
  const int kt_end = min(p.nk, kt_begin + p.kt_per_split);
  int s_cb = kt_begin / 9, s_tap = kt_begin - s_cb * 9, s_kt = kt_begin;  // next weight tile to stage
  int w_cb = s_cb, w_ky = s_tap / 3;                                        // next window to stage
  const int row_bytes = p.W * p.lda * 2;                                    // one image row of the input
  // GNA: this thread's 8 channels of the chunk whose windows are being transformed, and the next window to transform
  // (the sample's whole (scale, shift) table sits in the LDS behind the weight ring - Cin * 8 bytes, copied once: 16 registers
  // per thread held across the K loop made the kernel spill)
  [[maybe_unused]] int g_ky = w_ky, g_cb = w_cb;
  [[maybe_unused]] float* const gtab = (float*)(bring + 2 * BSLOT);
#define WIN2_GN_LOAD(CB)
#define WIN2_GN_APPLY(WBI)                                                                                \
  if constexpr (GNA) {                                                                                    \
    const int kbit = 1 << g_ky;                                                                           \
    float gsc[8], gsh[8];                                                                                 \
    {                                                                                                     \
      const f32x4* sp = (const f32x4*)(gtab + (g_cb * BK + lchunk * 8) * 2);                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
        const f32x4 v = sp[e];                                                                            \
        gsc[2 * e] = v[0]; gsh[2 * e] = v[1]; gsc[2 * e + 1] = v[2]; gsh[2 * e + 1] = v[3];               \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < PW; ++i) {                                                      \
      if (wave_row0 + 32 * i < WROWS && (w_mask[i] & kbit)) {                                             \
        uint4* q = (uint4*)(smem + (WBI) * WBUF + (r0 + 32 * i) * BK + (t & 7) * 8);                      \
        float f[8];                                                                                       \
        unpack8(*q, f);                                                                                   \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) f[e] = silu_f(__builtin_fmaf(f[e], gsc[e], gsh[e])); \
        *q = pack8(f);                                                                                    \
      }                                                                                                   \
    }                                                                                                     \
    if (++g_ky == 3) { g_ky = 0; ++g_cb; }                                                                \
  }

#define WIN2_STAGE_W(WBI)                                                                                 \
  {                                                                                                       \
    const int soff = (UPS ? 0 : w_ky * row_bytes) + w_cb * (BK * 2);                                      \
    const int kbit = 1 << w_ky;                                                                           \
    [[maybe_unused]] const int ups_sh = w_ky == 1 ? 4 : 5, ups_and = w_ky == 0 ? 0 : (w_ky == 1 ? 1 : 3); \
    _Pragma("unroll") for (int i = 0; i < PW; ++i) {                                                      \
      if (wave_row0 + 32 * i < WROWS) {                                                                   \
        unsigned vsel = w_voff[i];                                                                        \
        if constexpr (UPS) vsel += (unsigned)(((w_mask[i] >> ups_sh) & ups_and) * row_bytes);             \
        const unsigned vo = (w_mask[i] & kbit) ? vsel : DMA_OOB;                                          \
        dma_buf16(rsrcA, vo, soff, lds0 + ((WBI) * WBUF + (wave_row0 + 32 * i) * BK) * 2);                \
      }                                                                                                   \
    }                                                                                                     \
    if (++w_ky == 3) { w_ky = 0; ++w_cb; }                                                                \
  }
#define WIN2_STAGE_B(SLOT)                                                                                \
  if (s_kt < kt_end) {                                                                                    \
    const int soff = (s_tap * p.Cin + s_cb * BK) * 2;                                                     \
    _Pragma("unroll") for (int i = 0; i < PB; ++i)                                                        \
      dma_buf16(rsrcB, b_voff[i], soff, lds0 + (2 * WBUF + (SLOT) * BSLOT + (wave_row0 + 32 * i) * BK) * 2); \
    ++s_kt;                                                                                               \
    if (++s_tap == 9) { s_tap = 0; ++s_cb; }                                                              \
  }
#define WIN2_WAIT()                                               \
  {                                                               \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                                 \
  }
#define WIN2_READ(AF, WF, WB, KX, KK, SLOT)                                                                         \
  {                                                                                                                 \
    const u16* cB = bring + (SLOT) * BSLOT;                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) AF[i] = *(const hx8*)&(WB)[aoff[i][KX] ^ ((KK) ? 32 : 0)];       \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const hx8*)&cB[boff[j] ^ ((KK) ? 32 : 0)];             \
  }
#define WIN2_MFMA(AF, WF)                                                                                  \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                         \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(WF[j], AF[i], acc[i][j], 0, 0, 0); \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  hx8 af0[MT], wf0[NT], af1[MT], wf1[NT];
  EpiRegs<TM, TN, false> er;
  const bool fast = p.fast_epi;
  if (fast && p.splits == 1) epi_prefetch<TM, TN, false>(p, er, 0, brow + wr * TM, bcol + wc * TN, lane);

  WIN2_GN_LOAD(w_cb);
  WIN2_STAGE_W(0);
  WIN2_STAGE_B(0);
  WIN2_STAGE_B(1);
  if constexpr (GNA) {
    // the table copy rides behind the first DMAs; its load is the youngest in the queue, so the wait it needs is vmcnt(0)
    const float* src = p.gn_ss + (long long)(brow / hw) * p.Cin * 2;
    for (int i = t; i < p.Cin / 2; i += 256) *(f32x4*)(gtab + 4 * i) = *(const f32x4*)(src + 4 * i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // and this thread's own window pieces have landed
    __syncthreads();
    WIN2_GN_APPLY(0);
  }
  WIN2_WAIT();
  WIN2_READ(af0, wf0, smem, 0, 0, 0);
  // group structure: see conv_win_kernel.  GNA: the next window is staged after the first barrier of a group, has landed at
  // the second (vmcnt(0)), is transformed behind it by the threads that staged it, and is first read after the third.
#define WIN2_GROUP(WCUR, WNEXT, WNI, S0, S1)                                                   \
  {                                                                                            \
    WIN2_READ(af1, wf1, WCUR, 0, 1, S0);                                                       \
    WIN2_MFMA(af0, wf0);                                                                       \
    WIN2_WAIT();                                                                               \
    WIN2_STAGE_B(S0);                                                                          \
    if (kt + 3 < kt_end) {                                                                     \
      if (w_ky == 0) WIN2_GN_LOAD(w_cb);                                                       \
      WIN2_STAGE_W(WNI);                                                                       \
    }                                                                                          \
    WIN2_READ(af0, wf0, WCUR, 1, 0, S1);                                                       \
    WIN2_MFMA(af1, wf1);                                                                       \
    WIN2_READ(af1, wf1, WCUR, 1, 1, S1);                                                       \
    WIN2_MFMA(af0, wf0);                                                                       \
    WIN2_WAIT();                                                                               \
    WIN2_STAGE_B(S1);                                                                          \
    if (kt + 3 < kt_end) WIN2_GN_APPLY(WNI);                                                   \
    WIN2_READ(af0, wf0, WCUR, 2, 0, S0);                                                       \
    WIN2_MFMA(af1, wf1);                                                                       \
    WIN2_READ(af1, wf1, WCUR, 2, 1, S0);                                                       \
    WIN2_MFMA(af0, wf0);                                                                       \
    WIN2_WAIT();                                                                               \
    WIN2_STAGE_B(S0);                                                                          \
    WIN2_READ(af0, wf0, WNEXT, 0, 0, S1);                                                      \
    WIN2_MFMA(af1, wf1);                                                                       \
  }
  for (int kt = kt_begin; kt < kt_end; kt += 3) {
    WIN2_GROUP(smem, smem + WBUF, 1, 0, 1);
    kt += 3;
    if (kt >= kt_end) break;
    WIN2_GROUP(smem + WBUF, smem, 0, 1, 0);
  }
#undef WIN2_GROUP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef WIN2_GN_LOAD
#undef WIN2_GN_APPLY
#undef WIN2_STAGE_W
#undef WIN2_STAGE_B
#undef WIN2_WAIT
#undef WIN2_READ
#undef WIN2_MFMA
  igemm_finish<TM, TN, false, false, CS>(p, acc, er, fast, smem, 0, brow + wr * TM, bcol + wc * TN, (lds_cfp)smem, 0, 0, 0, (brow + wr * TM) / 32,
                              (bcol + wc * TN) / 32);
}

// ------------------------------------------------------------------------------------------ host
// Split-K factor.  Depends only on the PER-SAMPLE problem (rows_per_sample, N, K), never on the batch, so one image
// produces the same bits whether it runs alone or inside a larger batch (the image-sharded multi-GPU path relies on
// it).  Under-filled grids (few output tiles, long K) are cut so that >= ~384 blocks exist per sample.
// tuning override (sdmi_tune_override): -1 = automatic
// (thread-local like the error state: a tuning or profiling call on one host thread never changes another thread's launches)
static thread_local int g_force_tile = -1, g_force_splits = -1, g_force_window = -1;
extern "C" void sdmi_tune_override(int tile, int splits) { g_force_tile = tile; g_force_splits = splits; }
extern "C" void sdmi_tune_conv_window(int on) { g_force_window = on; }

// Profiling aid (bench.py's roofline pass): an event the next split-K launch records on its stream BETWEEN the main
// kernel and the fixed-order reduce kernel, so that a caller bracketing the entry point with two events of its own can
// price the two kernels apart.  One-shot: cleared by the launch that uses it (and by passing NULL).
static thread_local hipEvent_t g_mid_event = nullptr;
extern "C" void sdmi_prof_mid_event(void* ev) { g_mid_event = (hipEvent_t)ev; }
static void record_mid_event(hipStream_t st) {
  if (g_mid_event) { (void)hipEventRecord(g_mid_event, st); g_mid_event = nullptr; }
}

static int splitk_factor(int rows_per_sample, int n, int nk, uint32_t flags, bool conv) {
  if (flags & (SDMI_F_GEGLU | SDMI_F_LNFOLD)) return 1;
  if (g_force_splits > 0) return g_force_splits > nk ? nk : g_force_splits;
  static int target = -1;
  if (target < 0) {
    const char* e = getenv("SDMI_SPLITK_TARGET");
    target = e ? atoi(e) : 512;
  }
  if (target <= 0) return 1;
  const int bn = ((n % 128 == 0) || (n > 512)) ? 128 : 64;
  const int bm = rows_per_sample >= 1024 ? 128 : 64;
  const long long tiles = (long long)((rows_per_sample + bm - 1) / bm) * ((n + bn - 1) / bn);
  const int opt_target = sdmi_opt(OPT_splitk_target), opt_max = sdmi_opt(OPT_splitk_max_tiles);
  if (tiles > (opt_max > 0 ? opt_max : 256)) return 1;
  long long s = (opt_target > 0 ? opt_target : target) / tiles;
  // at least 1024 of K per split (plain GEMMs: 1536), below that the fp32 round trip costs more than it buys (sweep of
  // the UNet's shapes, scratch/tune_sweep.py: K = 2560 GEMMs ran 25-35 % faster unsplit).  Round 2, with the two CFG
  // halves as concurrent batch-1 passes (scratch/ab_step.py, same-process A/B of the captured step): 512 blocks per
  // sample instead of 384 and splitting up to 256 tiles instead of 128 -1.3 %, GEMM splits from K = 3072 -0.5 %.
  const int opt_kt = sdmi_opt(conv ? OPT_splitk_min_kt_conv : OPT_splitk_min_kt_gemm);
  // (round 3, with conv_win2_kernel: at least 24 K-tiles per conv split instead of 16 - fewer, longer splits: captured step
  // 8.043 -> 7.965 and 8.005 -> 7.979 ms in two interleaved runs, 32 equal, 48 +4 %; conv_plan_r2 = 1 restores 16)
  const int min_kt = opt_kt > 0 ? opt_kt : (conv ? (sdmi_opt(OPT_conv_plan_r2) ? 16 : 24) : 24);
  if (s > nk / min_kt) s = nk / min_kt;
  if (s > 16) s = 16;
  if (conv && !sdmi_opt(OPT_conv_plan_r2)) {
    // round 3, batch-1 sweep with the round-3 kernels (profiles/r03_sweep_conv_b1.txt):
    // - a 48x48 conv with 320 input channels (45 K-tiles) runs 25 % faster unsplit on 128 x 64 tiles (180 of them) than as
    //   two splits of 128 x 128 tiles + a reduce kernel;
    // - the 12x12 convs with 2560 input channels: 12 splits instead of 16 (-14 %: a third less partial traffic)
    if (nk < 48 && rows_per_sample >= 2048) return 1;
    if (rows_per_sample <= 256 && s > 12) s = 12;
  }
  return s < 2 ? 1 : (int)s;
}

static int64_t splitk_ws_bytes(int m, int n, int nk, int batch, int rows_per_sample, uint32_t flags, bool conv) {
  const int s = splitk_factor(rows_per_sample, n, nk, flags, conv);
  if (s < 2) return 0;
  const int kps = (nk + s - 1) / s;
  const int splits = (nk + kps - 1) / kps;
  return (int64_t)splits * batch * m * ((n + 3) / 4 * 4) * (int64_t)sizeof(float);
}

// Tickets of the in-launch split-K reduction (SDMI_F_WS_TICKETS): one arrival counter per output tile, zero at load and
// re-armed by the reducing block of every launch, in this library-owned array.  A bank of counters belongs to ONE
// registered workspace (sdmi_ws_tickets_init), so launches that may overlap in time - which need different workspaces
// anyway - never share counters.  (Round 1 picked the bank from a hash of the workspace address: two 64 MB-aligned
// workspaces of two concurrent streams hashed to the same bank.)
constexpr int SK_TICKET_TILES = 4096, SK_TICKET_BANKS = 16;
__device__ unsigned g_sk_ticket[SK_TICKET_BANKS * SK_TICKET_TILES];
struct TicketBanks {
  std::mutex mu;
  const void* owner[64][SK_TICKET_BANKS] = {};  // [device][bank] -> registered workspace
  unsigned* base[64] = {};                       // device address of g_sk_ticket on that device
};
static TicketBanks g_banks;
// bank of a registered workspace on the current device, or nullptr; reg: claim a free bank if it has none yet
static unsigned* ticket_bank(const void* ws, bool reg, bool* full = nullptr) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  std::lock_guard<std::mutex> lock(g_banks.mu);
  if (!g_banks.base[dev]) {
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_sk_ticket)) != hipSuccess || !sym) return nullptr;
    g_banks.base[dev] = (unsigned*)sym;
  }
  int free_bank = -1;
  for (int b = 0; b < SK_TICKET_BANKS; ++b) {
    if (g_banks.owner[dev][b] == ws) return g_banks.base[dev] + b * SK_TICKET_TILES;
    if (!g_banks.owner[dev][b] && free_bank < 0) free_bank = b;
  }
  if (!reg) return nullptr;
  if (free_bank < 0) { if (full) *full = true; return nullptr; }
  g_banks.owner[dev][free_bank] = ws;
  return g_banks.base[dev] + free_bank * SK_TICKET_TILES;
}

// One tile variant: BM x BN block tile, WM x WN waves (wave tile BM/WM x BN/WN), NSTAGE ring slots in dynamic LDS.
struct TileVariant { int bm, bn, wm, wn, nstage; };
static const TileVariant kVariants[] = {
    {256, 128, 4, 2, 3},  // 0: 8 waves x (64 x 64), 144 KB LDS, 1 block / CU, 2 K-tiles in flight
    {128, 128, 2, 2, 2},  // 1: 4 waves x (64 x 64), 64 KB, 2 blocks / CU
    {128, 64, 2, 2, 2},   // 2: 48 KB, 3 blocks / CU
    {64, 128, 2, 2, 3},   // 3
    {64, 64, 2, 2, 3},    // 4
    // deeper rings for the latency-bound plain GEMMs (5 - 20 K-tiles; a K-tile costs one L2 round trip / tiles in flight)
    {64, 64, 2, 2, 5},    // 5: 80 KB, 2 blocks / CU, 4 K-tiles in flight
    {128, 64, 2, 2, 3},   // 6: 72 KB, 2 blocks / CU
    {64, 128, 2, 2, 4},   // 7: 96 KB, 1 block / CU
    // shared-window conv only (conv_win2_kernel<256, 128>): 4 waves x (128 x 64) - twelve fragment reads per 32 MFMAs instead
    // of eight per 16 (a 64 x 64 wave tile reads 128 B of LDS per clock and CU at full MFMA rate: the LDS's whole bandwidth),
    // 100 KB of LDS, one block per CU, one wave per SIMD (up to 512 VGPRs)
    {256, 128, 2, 2, 2},  // 8
};
constexpr int VI_WIN256 = 8;

template <int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, bool GEGLU, bool BUF = false>
static void launch_variant2(dim3 grid, hipStream_t st, const IgemmParams& p) {
  constexpr int ring = NSTAGE * (BM + BN) * BK * 2;
  constexpr int side = CONV ? 0 : BM * 8 + BN * 4;  // LNFOLD: (mean, rstd) per tile row + column sums, behind the ring
  constexpr int bytes = ring + side;
  auto kfn = igemm_kernel<BM, BN, WM, WN, CONV, NSTAGE, GEGLU, BUF>;
  static unsigned long long attr_done = 0;  // per device: > 64 KB of dynamic LDS has to be allowed once
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bytes > 65536 && !((attr_done >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    attr_done |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(kfn, grid, dim3(64 * WM * WN), p.ln_stats ? bytes : ring, st, p);
}
// GEMMs whose K is a whole number of K-tiles and whose block panels stay below 2 GB stage through buffer descriptors (the
// four tile shapes the UNet / VAE use; gemm_buf_off = 1 keeps the pointer-per-lane loader: A/B switch, same bits)
static bool gemm_buf_ok(const IgemmParams& p, int bm, int bn) {
  return !sdmi_opt(OPT_gemm_buf_off) && p.K % BK == 0 && (long long)bm * p.lda * 2 + 2ll * p.K < (1ll << 31) &&
         (long long)bn * p.ldw * 2 + 2ll * p.K < (1ll << 31);
}
template <int BM, int BN, int WM, int WN, bool CONV, int NSTAGE>
static void launch_variant(dim3 grid, hipStream_t st, const IgemmParams& p) {
  constexpr bool BUF_VARIANT = !CONV && WM * WN == 4 && ((BM == 128 && NSTAGE == 2) || (BM == 64 && NSTAGE == 3));
  if constexpr (!CONV && (BN / WN) % 32 == 0) {
    if (p.flags & SDMI_F_GEGLU) {
      if constexpr (BUF_VARIANT) if (gemm_buf_ok(p, BM, BN)) { launch_variant2<BM, BN, WM, WN, CONV, NSTAGE, true, true>(grid, st, p); return; }
      launch_variant2<BM, BN, WM, WN, CONV, NSTAGE, true>(grid, st, p);
      return;
    }
  }
  if constexpr (BUF_VARIANT) if (gemm_buf_ok(p, BM, BN)) { launch_variant2<BM, BN, WM, WN, CONV, NSTAGE, false, true>(grid, st, p); return; }
  launch_variant2<BM, BN, WM, WN, CONV, NSTAGE, false>(grid, st, p);
}

// A-stationary launch (igemm_astat_kernel): tiles_m x nsplit blocks, dynamic LDS = panel + ring + LayerNorm side region
template <int BN, bool GEGLU>
static void launch_astat2(int blocks, int nsplit, int bytes, hipStream_t st, const IgemmParams& p) {
  auto kfn = igemm_astat_kernel<BN, GEGLU>;
  static unsigned long long attr_done = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    attr_done |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), bytes, st, p, nsplit);
}
// persistent launch (igemm_persist_kernel): `slots` blocks walk the nb tiles
template <int BM, int BN, int WM, int WN, int NSTAGE, bool GEGLU>
static void launch_persist2(int slots, int batch, hipStream_t st, const IgemmParams& p) {
  constexpr int ring = NSTAGE * (BM + BN) * BK * 2;
  constexpr int bytes = ring + BM * 8 + BN * 4;
  auto kfn = igemm_persist_kernel<BM, BN, WM, WN, NSTAGE, GEGLU>;
  static unsigned long long attr_done = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bytes > 65536 && !((attr_done >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    attr_done |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(kfn, dim3(slots, batch, 1), dim3(64 * WM * WN), p.ln_stats ? bytes : ring, st, p);
}
template <int BM, int BN, int WM, int WN, int NSTAGE>
static void launch_persist(int slots, int batch, hipStream_t st, const IgemmParams& p) {
  if constexpr ((BN / WN) % 32 == 0) {
    if (p.flags & SDMI_F_GEGLU) { launch_persist2<BM, BN, WM, WN, NSTAGE, true>(slots, batch, st, p); return; }
  }
  launch_persist2<BM, BN, WM, WN, NSTAGE, false>(slots, batch, st, p);
}
// blocks of a tile variant one CU holds (LDS; the 4-wave tiles run at 2 waves per SIMD or more)
static int variant_blocks_per_cu(const TileVariant& tv, bool ln) {
  const int bytes = tv.nstage * (tv.bm + tv.bn) * BK * 2 + (ln ? tv.bm * 8 + tv.bn * 4 : 0);
  int n = (160 * 1024) / bytes;
  if (tv.wm * tv.wn == 8) n = n > 1 ? 1 : n;
  return n > 2 ? 2 : (n < 1 ? 1 : n);
}

// Tile choice.  The main loop is bound by the L2 -> LDS path (about 70 GB/s per CU), so the biggest tile that still
// gives every CU a block wins; SDMI_IGEMM_TILE=<index into kVariants> forces one (experiments).
// what the shared-window conv kernels need (stride 1, pad 1, same size, no upsampling; v2: whole 64-channel chunks, 32-bit
// panel offsets)
static bool conv_win2_ok(const IgemmParams& p, int bm) {
  if (p.ups)  // absolute per-lane offsets from the start of the tensor (conv_win2_kernel UPS)
    return !sdmi_opt(OPT_conv_win_v1) && !sdmi_opt(OPT_conv_win_ups_off) && bm == 128 && p.Cin % 64 == 0 && (long long)p.N * p.ldw * 2 < (1ll << 31) &&
           (long long)(p.M / 4) * p.lda * 2 + 2 * p.Cin < (1ll << 31);
  return !sdmi_opt(OPT_conv_win_v1) && p.Cin % 64 == 0 && (long long)p.N * p.ldw * 2 < (1ll << 31) &&
         ((long long)(bm + 16) + 2 * p.W) * p.lda * 2 + 2 * p.Cin < (1ll << 31);
}
static bool conv_window_shape(const IgemmParams& p, int batch) {
  if (p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || batch != 1) return false;
  if (p.ups) return p.Ho == 2 * p.H && p.Wo == 2 * p.W && conv_win2_ok(p, 128);  // only the round-3 kernel has the upsample variant
  return p.Ho == p.H && p.Wo == p.W;
}
static int choose_variant(const IgemmParams& p, int batch, bool conv) {
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("SDMI_IGEMM_TILE");
    forced = e ? atoi(e) : -1;
  }
  if (g_force_tile >= 0 && g_force_tile < (int)(sizeof(kVariants) / sizeof(kVariants[0]))) return g_force_tile;
  if (forced >= 0 && forced < (int)(sizeof(kVariants) / sizeof(kVariants[0]))) {
    return forced;
  }
  if (!conv && !(p.flags & SDMI_F_GEGLU)) {
    // plain GEMMs (K = 320 ... 5120: 5-80 K-tiles, latency-bound): three 128x64 blocks per CU hide the DMA wait better
    // than two 128x128 ones; grids too small for that run the smallest tile, except long-K wide-N ones (weights are the
    // traffic there).  From the sweep of the UNet's shapes, scratch/tune_sweep.py.
    const long long b2 = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * batch * p.splits;
    const int t2_min = sdmi_opt(OPT_gemm_t2_min_blocks);
    // (round 2, two concurrent batch-1 passes: 360 instead of 600 - the 9216 x 320 GEMMs of one CFG half - step -1.1 %)
    if (b2 >= (t2_min > 0 ? t2_min : 360)) return 2;
    // (64 x 128 only with enough rows: at M = 576 / 144, one CFG half of the 24x24 / 12x12 levels, 64 x 64 ran 1.1 - 1.6x
    // faster on the K >= 2048 GEMMs: scratch/gemm_sweep_b1.py)
    if (p.K >= 2048 && p.N >= 1024 && (p.M >= 1024 || !sdmi_opt(OPT_gemm_small_m_tile))) return 3;
    return 4;
  }
  if (!conv && p.M <= 256) return 3;  // GEGLU on a 12x12 level: 64-row tiles (144 rows in 128-row tiles compute 256)
  const bool bn128 = (p.N % 128 == 0) || (p.N > 512);
  const int bn = bn128 ? 128 : 64;
  const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn) * batch * p.splits;
  // 128-row tiles (and with them the shared-window conv kernel) once they give half the CUs a block: the sampler runs
  // the two CFG halves as concurrent batch-1 passes, so the other half of the chip is not idle (256 -> 128: +3.3 %
  // images/s at 768x768; SDMI_BM128_MIN overrides)
  // (round 3, with the round-3 kernels and split rules: 192 instead of 128 - captured step 7.979 -> 7.927 and 8.135 -> 8.086 ms
  // in two interleaved runs, 160 ... 224 equal, 256 +1.5 %; conv_plan_r2 = 1 restores 128)
  static const int bm128_env = [] { const char* e = getenv("SDMI_BM128_MIN"); return e ? atoi(e) : 0; }();
  const int opt_bm = sdmi_opt(OPT_bm128_min);
  const int bm128_min = opt_bm > 0 ? opt_bm : (bm128_env > 0 ? bm128_env : (sdmi_opt(OPT_conv_plan_r2) ? 128 : 192));
  const bool bm128 = p.M > 64 && tiles128 >= bm128_min;
  // 256 x 128 window tiles (4 waves x 128 x 64: 1.5 fragment reads per 4 MFMAs instead of 2): opt-in (conv_win_256 = 1).
  // The same loop at one block per CU and one wave per SIMD is 13-38 % SLOWER on the VAE's levels and 17-24 % at batch 4
  // (profiles/r03_conv_win256_ab.txt; same bits): what a larger wave tile saves in LDS reads it loses to the waits a lone
  // wave per SIMD cannot hide - round 2's 8-wave 256 x 128 GEMM variant lost the same way.
  if (conv && bn128 && p.N % 128 == 0 && sdmi_opt(OPT_conv_win_256) > 0 && conv_window_shape(p, batch) && conv_win2_ok(p, 256)) return VI_WIN256;
  if (bm128 && bn128) return 1;
  if (bm128) return 2;
  // (round 3) unsplit short-K convs whose 128 x 128 grid is too small: 128 x 64 tiles once there are 160 of them
  if (conv && bn128 && p.splits == 1 && !sdmi_opt(OPT_conv_plan_r2) && p.M > 64 &&
      (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * batch >= 160) return 2;
  if (bn128) return 3;
  return 4;
}

// Split-K factor and tile variant of a problem (no launch): shared by the launch path and the sdmi_*_stats_parts queries.
template <bool CONV>
static int plan_igemm(IgemmParams& p, int batch, bool have_ws, int64_t ws_bytes) {
  p.splits = 1;
  p.kt_per_split = p.nk;
  p.npad = (p.N + 3) / 4 * 4;
  const int s = splitk_factor(p.rows_per_sample, p.N, p.nk, p.flags, CONV);
  if (s >= 2 && have_ws && ws_bytes >= splitk_ws_bytes(p.M, p.N, p.nk, batch, p.rows_per_sample, p.flags, CONV)) {
    p.kt_per_split = (p.nk + s - 1) / s;
    if (CONV) p.kt_per_split = (p.kt_per_split + 2) / 3 * 3;  // whole (chunk, ky) groups: the shared-window kernel needs it
    p.splits = (p.nk + p.kt_per_split - 1) / p.kt_per_split;
  }
  return choose_variant(p, batch, CONV);
}
// row parts (colstats) / column parts (rowstats) a planned launch writes; 0 = it cannot emit them
static void stats_parts_of(const IgemmParams& p, int batch, int vi, int* row_parts, int* col_parts) {
  // parts are 32 rows / 32 columns whatever the tile variant: what a consumer sums does not depend on the variant, and so
  // not on the batch (the variant choice looks at the whole grid)
  const TileVariant& tv = kVariants[vi];
  const int tm = tv.bm / tv.wm, tn = tv.bn / tv.wn;
  const bool plain = p.splits == 1 && batch == 1 && p.N % 8 == 0 &&
                     !(p.flags & (SDMI_F_GEGLU | SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_U8_IMAGE | SDMI_F_OUT_F32)) && tn >= 32 && tm >= 32;
  // a split-K launch leaves the column statistics from its reduce kernel (igemm_splitk_reduce_stats_kernel)
  const bool split_cs = p.splits > 1 && batch == 1 && p.N % 4 == 0 && p.M % 32 == 0 && !sdmi_opt(OPT_split_colstats_off) &&
                        !(p.flags & (SDMI_F_GEGLU | SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_U8_IMAGE | SDMI_F_OUT_F32));
  if (row_parts) *row_parts = (((plain && p.M % tm == 0) || split_cs) && p.rows_per_sample % 32 == 0) ? p.M / 32 : 0;
  if (col_parts) *col_parts = (plain && p.N % 32 == 0) ? p.N / 32 : 0;
}

// sdmi_epilogue_ext.gn_in: what the planned launch must be (128 x 128 window tiles of the round-3 kernel, one column block, row
// tiles inside one sample)
static bool gn_in_ok(const IgemmParams& p, int vi, int batch) {
  static int win_env = -1;
  if (win_env < 0) { const char* e = getenv("SDMI_CONV_WIN"); win_env = (e && e[0] == '0') ? 0 : 1; }
  return vi == 1 && !p.ups && p.Cin <= 1024 && p.N <= 128 && p.splits == 1 && batch == 1 && p.rows_per_sample % 128 == 0 && conv_window_shape(p, batch) &&
         conv_win2_ok(p, 128) && (g_force_window >= 0 ? g_force_window : win_env) != 0;
}
// conv_win2_kernel launch; > 64 KB of dynamic LDS allowed once per device
template <int BM, int BN, bool CS, bool GNA = false, bool UPS = false>
static void launch_win2_cs(dim3 grid, int bytes, hipStream_t st, const IgemmParams& p) {
  auto kfn = conv_win2_kernel<BM, BN, CS, GNA, UPS>;
  static unsigned long long attr_done = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bytes > 65536 && !((attr_done >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    attr_done |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), bytes, st, p);
}
template <int BM, int BN>
static void launch_win2(dim3 grid, int bytes, hipStream_t st, const IgemmParams& p) {
  if constexpr (BM == 128) {
    if (p.ups) {
      if (p.colstats) launch_win2_cs<BM, BN, true, false, true>(grid, bytes, st, p);
      else launch_win2_cs<BM, BN, false, false, true>(grid, bytes, st, p);
      return;
    }
  }
  if (p.colstats) launch_win2_cs<BM, BN, true>(grid, bytes, st, p);
  else launch_win2_cs<BM, BN, false>(grid, bytes, st, p);
}
// GroupNorm + SiLU of the input applied to the staged windows (128 x 128 tiles, one column block)
static void launch_win2_gn(dim3 grid, int bytes, hipStream_t st, const IgemmParams& p) {
  bytes += 8192;  // the (scale, shift) table of one sample: Cin * 8 bytes, Cin <= 1024 (gn_in_ok); 75 KB: still two blocks per CU
  if (p.colstats) launch_win2_cs<128, 128, true, true>(grid, bytes, st, p);
  else launch_win2_cs<128, 128, false, true>(grid, bytes, st, p);
}

// the second pass of a split-K launch: plain reduce, reduce + column statistics, or reduce + the GroupNorm that follows
static bool gn_out_shape_ok(const IgemmParams& p, int batch, int groups) {
  if (groups <= 0 || p.N % groups != 0) return false;
  const int cpg = p.N / groups;
  const long long dwords = (long long)p.rows_per_sample * (cpg / 2);
  // a (sample, group) slice must fit one block's registers next to the fp32 sums of its partials: 12 channel pairs per thread
  // (24 pairs - the largest slices sdmi_groupnorm's fused kernel takes - spill here; such results keep the two kernels)
  const bool small = dwords <= 12 * 1024;
  return p.splits > 1 && batch == 1 && small && cpg % 2 == 0 && cpg <= 160 && p.N % 4 == 0 && p.rows_per_sample > 0 &&
         p.M % p.rows_per_sample == 0 && p.rows_per_sample < (1 << 19) && (long long)(p.M / p.rows_per_sample) * groups <= 65535 &&
         !(p.flags & (SDMI_F_GEGLU | SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_U8_IMAGE | SDMI_F_OUT_F32 | SDMI_F_LNFOLD));
}
static int launch_splitk_second_pass(const IgemmParams& p, int batch, hipStream_t st, float* split_colstats, const sdmi_gn_out* gn) {
  record_mid_event(st);
  if (gn) {
    GnOutParams go;
    go.y = (u16*)gn->y; go.ldy = gn->ldy; go.gamma = gn->gamma; go.beta = gn->beta; go.eps = gn->eps; go.groups = gn->groups;
    go.silu = (gn->flags & SDMI_F_SILU) ? 1 : 0; go.keep_raw = gn->keep_raw ? 1 : 0;
    const int cpg = p.N / gn->groups;
    const int dpt = (int)(((long long)p.rows_per_sample * (cpg / 2) + 1023) / 1024);
    const dim3 grid((unsigned)((p.M / p.rows_per_sample) * gn->groups));
    if (dpt <= 6) hipLaunchKernelGGL(igemm_splitk_reduce_gn_kernel<6>, grid, dim3(1024), 0, st, p, go);
    else hipLaunchKernelGGL(igemm_splitk_reduce_gn_kernel<12>, grid, dim3(1024), 0, st, p, go);
    return sdmi_check_launch("split-K reduce + GroupNorm");
  }
  long long items = (long long)batch * p.M * (p.npad / 4);
  long long rb = (items + 255) / 256;
  if (rb > 4096) rb = 4096;
  if (split_colstats) hipLaunchKernelGGL(igemm_splitk_reduce_stats_kernel, dim3((unsigned)((p.M / 32) * ((p.N + 31) / 32))), dim3(256), 0, st, p, split_colstats);
  else hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p, batch);
  return sdmi_check_launch("split-K reduce");
}

template <bool CONV>
static int launch_igemm(IgemmParams& p, int batch, void* ws, int64_t ws_bytes, hipStream_t st, const sdmi_epilogue_ext* ext = nullptr) {
  p.ws = nullptr;
  const int vi = plan_igemm<CONV>(p, batch, ws != nullptr, ws_bytes);
  if (p.splits > 1) p.ws = (float*)ws;
  const TileVariant& tv = kVariants[vi];
  const int bm = tv.bm, bn = tv.bn;
  const int nb = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
  p.colstats = nullptr; p.rowstats = nullptr; p.ln_stats = nullptr; p.ln_u = nullptr; p.ln_parts = 0; p.col_parts = 0; p.ln_eps = 0.f;
  p.gn_ss = nullptr;
  float* split_colstats = nullptr;
  if (ext && ext->gn_in) {
    // the conv applies GroupNorm + SiLU to its input itself: only where ONE block column covers every output channel (else
    // each column block would transform the same staged windows again) on the round-3 shared-window kernel, and never
    // silently on another path
    SDMI_REQUIRE(CONV, "gemm: gn_in is a conv3x3 mode");
    SDMI_REQUIRE(p.flags & SDMI_F_SILU, "conv3x3: gn_in applies GroupNorm + SiLU; pass SDMI_F_SILU");
    SDMI_REQUIRE(((uintptr_t)ext->gn_in & 15) == 0, "conv3x3: gn_in must be 16-byte aligned");
    SDMI_REQUIRE(gn_in_ok(p, vi, batch),
                 "conv3x3: gn_in needs a stride-1 pad-1 conv with Cout <= 128, Cin %% 64 == 0, H*W %% 128 == 0 on 128 x 128 window tiles "
                 "(cin=%d cout=%d hw=%d variant=%d splits=%d)", p.Cin, p.N, p.rows_per_sample, vi, p.splits);
    p.gn_ss = ext->gn_in;
  } else {
    SDMI_REQUIRE(!CONV || !(p.flags & SDMI_F_SILU), "conv3x3: SDMI_F_SILU without sdmi_epilogue_ext.gn_in");
  }
  const sdmi_gn_out* gn_out = ext ? ext->gn_out : nullptr;
  if (gn_out) {
    // the split-K reduce also writes the GroupNorm that follows: only for launches that split K, on results small enough for
    // the one-block-per-(sample, group) kernel; anything else is an error (sdmi_*_gn_out_ok tells beforehand)
    SDMI_REQUIRE(gn_out->y && gn_out->gamma && gn_out->beta && gn_out->ldy % 2 == 0 && gn_out->ldy >= p.N && ((uintptr_t)gn_out->y & 3) == 0 && gn_out->eps > 0.f,
                 "%s: gn_out needs y, gamma, beta, an even ldy >= N and eps > 0", CONV ? "conv3x3" : "gemm");
    SDMI_REQUIRE(gn_out_shape_ok(p, batch, gn_out->groups) && !(p.flags & SDMI_F_WS_TICKETS) && !ext->colstats && !ext->rowstats && p.ldc % 2 == 0,
                 "%s: gn_out needs a split-K launch whose (sample, group) slices fit one block (M=%d N=%d rows/sample=%d splits=%d groups=%d), "
                 "without colstats / rowstats / SDMI_F_WS_TICKETS", CONV ? "conv3x3" : "gemm", p.M, p.N, p.rows_per_sample, p.splits, gn_out->groups);
  }
  if (ext) {
    int rparts = 0, cparts = 0;
    stats_parts_of(p, batch, vi, &rparts, &cparts);
    SDMI_REQUIRE(!ext->colstats || rparts > 0, "%s: this problem cannot emit colstats (ask sdmi_*_stats_parts first)", CONV ? "conv3x3" : "gemm");
    SDMI_REQUIRE(!ext->rowstats || cparts > 0, "%s: this problem cannot emit rowstats (ask sdmi_*_stats_parts first)", CONV ? "conv3x3" : "gemm");
    SDMI_REQUIRE((((uintptr_t)ext->colstats | (uintptr_t)ext->rowstats) & 15) == 0, "epilogue ext: statistics buffers must be 16-byte aligned");
    p.colstats = ext->colstats; p.rowstats = ext->rowstats; p.col_parts = cparts;
    if (p.splits > 1) { split_colstats = p.colstats; p.colstats = nullptr; }  // written by the reduce kernel, not by the main one
    if (p.flags & SDMI_F_LNFOLD) {
      SDMI_REQUIRE(!CONV && ext->ln_rowstats && ext->ln_u && ext->ln_parts > 0 && ext->ln_eps > 0.f, "gemm: SDMI_F_LNFOLD needs ln_rowstats, ln_u, ln_parts, ln_eps");
      SDMI_REQUIRE(batch == 1 && p.splits == 1 && !(p.flags & (SDMI_F_ROWBIAS | SDMI_F_OUT_NCHW_F32)), "gemm: SDMI_F_LNFOLD with batch / split-K / rowbias / NCHW output");
      SDMI_REQUIRE((((uintptr_t)ext->ln_rowstats) & 7) == 0 && (((uintptr_t)ext->ln_u) & 3) == 0, "gemm: LNFOLD alignment");
      p.ln_stats = ext->ln_rowstats; p.ln_u = ext->ln_u; p.ln_parts = ext->ln_parts; p.ln_eps = ext->ln_eps;
    }
  }
  SDMI_REQUIRE(!(p.flags & SDMI_F_LNFOLD) || p.ln_stats, "gemm: SDMI_F_LNFOLD without an epilogue ext");
  {
    const uint32_t fl = p.flags;
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    bool fast = !(fl & (SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_U8_IMAGE)) && p.N % 8 == 0 && p.ldc % 8 == 0 && al16(p.out) && p.sc % 8 == 0;
    if (fl & SDMI_F_GEGLU) fast = fast && bn / tv.wn >= 64 && p.N % 32 == 0;
    if (fl & SDMI_F_BIAS) fast = fast && al16(p.bias);
    if (fl & SDMI_F_ROWBIAS) fast = fast && al16(p.rowbias) && p.ldrb % 4 == 0 && p.rows_per_sample >= bm;
    if (fl & SDMI_F_RESIDUAL) fast = fast && al16(p.res) && p.ldr % 8 == 0;
    p.fast_epi = fast ? 1 : 0;
    SDMI_REQUIRE(fast || !(p.colstats || p.rowstats || p.ln_stats), "%s: statistics / LNFOLD need the 16-byte epilogue (alignment of out / bias / residual, N %% 8)", CONV ? "conv3x3" : "gemm");
  }
  // the XCD remap gives every XCD (private L2) a contiguous run of tile ids: order the ids so that the run shares the
  // LARGER operand's panels - weight panels when N > M (the 24x24 / 12x12 levels: 29-59 MB of weights against 1-3 MB of
  // activations), activation panels otherwise.  SDMI_NMAJOR=0/1 forces it (A/B switch).
  {
    static int force = -2;
    if (force == -2) { const char* e = getenv("SDMI_NMAJOR"); force = e ? atoi(e) : -1; }
    // bytes of the two operands: weights N x K against activations M x K (GEMM) or M x Cin (conv: K = 9 Cin, the window
    // of a row tile is fetched once per tile)
    const long long wpanel = (long long)p.N * p.K, apanel = (long long)p.M * (CONV && !sdmi_opt(OPT_conv_nmajor_by_rows) ? p.K / 9 : p.K);
    p.nmajor = force >= 0 ? force : (wpanel > apanel ? 1 : 0);
  }
  // SDMI_F_WS_TICKETS (opt-in): the block that draws a tile's last ticket sums the partials in split order and runs the
  // epilogue itself (same bits as the reduce kernel).  Measured slower both rounds (round 1, one batch-2 pass: 10.9 against
  // 10.3 ms per step; round 2, two concurrent batch-1 passes: 8.82 against 8.28 ms; single convs +4 ... 15 us): the reducing
  // block's serial read of every partial tile costs more than the 7-13 us launch it replaces.
  p.ticket = nullptr;
  if (p.splits > 1 && !split_colstats && (p.flags & SDMI_F_WS_TICKETS) && (long long)nb * batch <= SK_TICKET_TILES)
    p.ticket = ticket_bank(ws, false);  // an unregistered workspace (or none free at registration): the reduce kernel
  dim3 grid(nb, batch, p.splits);
  if constexpr (CONV) {
    // stride-1 pad-1 convs on 128-row tiles: the shared-window kernel (SDMI_CONV_WIN=0 keeps the generic one: A/B switch)
    static int win_env = -1;
    if (win_env < 0) { const char* e = getenv("SDMI_CONV_WIN"); win_env = (e && e[0] == '0') ? 0 : 1; }
    const int win_ok = g_force_window >= 0 ? g_force_window : win_env;
    if (vi == VI_WIN256 && !(win_ok && conv_window_shape(p, batch) && conv_win2_ok(p, 256) && p.N % 128 == 0)) {
      sdmi_set_error("conv3x3: tile variant %d runs stride-1 pad-1 convs with Cin %% 64 == 0 and Cout %% 128 == 0 only", vi);
      return SDMI_EINVAL;
    }
    if (win_ok && (vi == 1 || vi == 2 || vi == VI_WIN256) && conv_window_shape(p, batch)) {
      constexpr int wbytes128 = (2 * (128 + 8 + 1) * BK + 2 * 128 * BK) * 2, wbytes64 = (2 * (128 + 8 + 1) * BK + 2 * 64 * BK) * 2;
      constexpr int wbytes256 = (2 * (256 + 8 + 1) * BK + 2 * 128 * BK) * 2;
      // round-3 kernel (buffer-descriptor staging, shift-proof swizzle) wherever its preconditions hold; conv_win_v1 = 1
      // keeps the round-2 kernel (A/B switch; same bits)
      const bool v2 = conv_win2_ok(p, bm);
      if (vi == VI_WIN256) {
        launch_win2<256, 128>(grid, wbytes256, st, p);
      } else if (vi == 1) {
        static unsigned long long attr_done = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((attr_done >> (dev & 63)) & 1ull)) {
          (void)hipFuncSetAttribute((const void*)conv_win_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wbytes128);
          (void)hipFuncSetAttribute((const void*)conv_win_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wbytes128);
          attr_done |= 1ull << (dev & 63);
        }
        if (p.gn_ss) {
          launch_win2_gn(grid, wbytes128, st, p);
        } else if (v2) {
          launch_win2<128, 128>(grid, wbytes128, st, p);
        } else {
          if (p.colstats) hipLaunchKernelGGL((conv_win_kernel<128, 128, true>), grid, dim3(256), wbytes128, st, p);
          else hipLaunchKernelGGL((conv_win_kernel<128, 128, false>), grid, dim3(256), wbytes128, st, p);
        }
      } else if (v2) {
        launch_win2<128, 64>(grid, wbytes64, st, p);
      } else {
        if (p.colstats) hipLaunchKernelGGL((conv_win_kernel<128, 64, true>), grid, dim3(256), wbytes64, st, p);
        else hipLaunchKernelGGL((conv_win_kernel<128, 64, false>), grid, dim3(256), wbytes64, st, p);
      }
      sdmi_note_kernel("%s<%d, %d, %s%s>%s", v2 ? "conv_win2_kernel" : "conv_win_kernel", bm, bn, p.colstats ? "true" : "false", p.gn_ss ? ", true" : (p.ups ? ", false, true" : ""),
                       (p.splits > 1 && !p.ticket) ? (gn_out ? " + igemm_splitk_reduce_gn_kernel" : split_colstats ? " + igemm_splitk_reduce_stats_kernel" : " + igemm_splitk_reduce_kernel") : "");
      int rcw = sdmi_check_launch("conv3x3 (shared window)");
      if (rcw || p.splits < 2 || p.ticket) return rcw;
      return launch_splitk_second_pass(p, batch, st, split_colstats, gn_out);
    }
  }
  SDMI_REQUIRE(!p.gn_ss, "conv3x3: gn_in is only available on the shared-window kernel (SDMI_CONV_WIN=0 or a forced variant is set)");
  // A-stationary blocks for short-K GEMMs with many column tiles per row tile: opt-in (gemm_astat = 1).  Same bits, half the
  // staged bytes, and 14-35 % SLOWER on every K = 320 GEMM tried (profiles/r03_gemm_astat_ab.txt: 9216 x 2560 x 320 GEGLU
  // 40.1 against 29.5 us): at one block per CU and one wave per SIMD a tile's K loop (five barrier-separated K-tiles) and
  // its epilogue run one after the other, ~5.7 us per tile, where two or three co-resident tile-per-block blocks overlap
  // them - these GEMMs are bound by that dependent chain per tile, not by the bytes through the L2 -> LDS path.
  if constexpr (!CONV) {
    const int aopt = sdmi_opt(OPT_gemm_astat);
    const bool geglu = (p.flags & SDMI_F_GEGLU) != 0;
    if (aopt > 0 && (vi == 1 || vi == 2) && batch == 1 && p.splits == 1 && p.fast_epi && p.K % 64 == 0 && p.nk <= 6 && p.N % bn == 0 &&
        (!geglu || vi == 1) && (long long)p.N * p.ldw * 2 < (1ll << 31) && (long long)128 * p.lda * 2 < (1ll << 31)) {
      static int cus[64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!cus[dev & 63]) {
        int n = 0;
        cus[dev & 63] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
      }
      const int tiles_m = (p.M + 127) / 128, tiles_n = p.N / bn;
      int nsplit = cus[dev & 63] / tiles_m;
      nsplit = nsplit < 1 ? 1 : (nsplit > tiles_n ? tiles_n : nsplit);
      {
        const int bytes = (p.nk * 128 * BK + 3 * bn * BK) * 2 + (2 * 128 + bn) * 4;
        if (vi == 1) {
          if (geglu) launch_astat2<128, true>(tiles_m * nsplit, nsplit, bytes, st, p);
          else launch_astat2<128, false>(tiles_m * nsplit, nsplit, bytes, st, p);
        } else {
          launch_astat2<64, false>(tiles_m * nsplit, nsplit, bytes, st, p);
        }
        sdmi_note_kernel("igemm_astat_kernel<%d, %s>", bn, geglu ? "true" : "false");
        return sdmi_check_launch("gemm (A-stationary)");
      }
    }
  }
  // Persistent blocks, the next tile's first K-tiles fetched under the current tile's tail and epilogue: an opt-in
  // (gemm_persist: 0 off, 1 for launches of more tiles than resident blocks, 2 whenever the kernel can run the problem,
  // 3 as 1 but GEGLU GEMMs only).
  // Bit-identical to igemm_kernel and measured equal or slower on every GEMM of the UNet (profiles/r02_persist_gemm_ab.txt).
  if constexpr (!CONV) {
    const int popt = sdmi_opt(OPT_gemm_persist);
    if (popt != 0 && p.splits == 1 && vi >= 1 && vi <= 4 && p.nk >= tv.nstage && !(p.flags & SDMI_F_GEGLU && (tv.bn / tv.wn) % 32 != 0)) {
      static int cus[64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!cus[dev & 63]) {
        int n = 0;
        cus[dev & 63] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
      }
      const int slots = cus[dev & 63] * variant_blocks_per_cu(tv, p.ln_stats != nullptr);
      if (popt == 2 || ((long long)nb * batch > slots && (popt != 3 || (p.flags & SDMI_F_GEGLU)))) {
        const int gx = nb < slots ? nb : (batch > 1 ? (slots / batch > 0 ? slots / batch : 1) : slots);
        if (vi == 1) launch_persist<128, 128, 2, 2, 2>(gx, batch, st, p);
        else if (vi == 2) launch_persist<128, 64, 2, 2, 2>(gx, batch, st, p);
        else if (vi == 3) launch_persist<64, 128, 2, 2, 3>(gx, batch, st, p);
        else launch_persist<64, 64, 2, 2, 3>(gx, batch, st, p);
        sdmi_note_kernel("igemm_persist_kernel<%d, %d, %d, %d, %d, %s>", bm, bn, tv.wm, tv.wn, tv.nstage, (p.flags & SDMI_F_GEGLU) ? "true" : "false");
        return sdmi_check_launch("gemm");
      }
    }
  }
  switch (vi) {
    case 0: launch_variant<256, 128, 4, 2, CONV, 3>(grid, st, p); break;
    case 1: launch_variant<128, 128, 2, 2, CONV, 2>(grid, st, p); break;
    case 2: launch_variant<128, 64, 2, 2, CONV, 2>(grid, st, p); break;
    case 3: launch_variant<64, 128, 2, 2, CONV, 3>(grid, st, p); break;
    case 4: launch_variant<64, 64, 2, 2, CONV, 3>(grid, st, p); break;
    default:
      if constexpr (!CONV) {  // GEMM-only variants
        if (vi == VI_WIN256) { sdmi_set_error("gemm: tile variant %d is conv-only", vi); return SDMI_EINVAL; }
        if (vi == 5) launch_variant<64, 64, 2, 2, false, 5>(grid, st, p);
        else if (vi == 6) launch_variant<128, 64, 2, 2, false, 3>(grid, st, p);
        else launch_variant<64, 128, 2, 2, false, 4>(grid, st, p);
      } else {
        sdmi_set_error("conv3x3: tile variant %d is GEMM-only", vi);
        return SDMI_EINVAL;
      }
      break;
  }
  sdmi_note_kernel("igemm_kernel<%d, %d, %d, %d, %s, %d, %s>%s", bm, bn, tv.wm, tv.wn, CONV ? "true" : "false", tv.nstage,
                   (p.flags & SDMI_F_GEGLU) ? "true" : "false", (p.splits > 1 && !p.ticket) ? (gn_out ? " + igemm_splitk_reduce_gn_kernel" : split_colstats ? " + igemm_splitk_reduce_stats_kernel" : " + igemm_splitk_reduce_kernel") : "");
  int rc = sdmi_check_launch(CONV ? "conv3x3" : "gemm");
  if (rc || p.splits < 2 || p.ticket) return rc;
  return launch_splitk_second_pass(p, batch, st, split_colstats, gn_out);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int gemm_params(IgemmParams& p, const void* a, int lda, int64_t sa, const void* w, int ldw, int64_t sw, void* out, int ldc,
                       int64_t sc, int m, int n, int k, int batch, const float* bias, const float* rowbias, int ldrb,
                       int rows_per_sample, const void* residual, int ldr, float alpha, uint32_t flags) {
  p = IgemmParams{};
  p.a = (const u16*)a; p.w = (const u16*)w; p.out = out; p.bias = bias; p.rowbias = rowbias; p.res = (const u16*)residual;
  p.sa = sa; p.sw = sw; p.sc = sc; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
  p.M = m; p.N = n; p.K = k; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : m; p.alpha = alpha; p.flags = flags;
  p.nk = (k + BK - 1) / BK;
  p.kt_per_tap = 1;
  return 0;
}

extern "C" int sdmi_gemm(const void* a, int lda, int64_t sa, const void* w, int ldw, int64_t sw, void* out, int ldc,
                         int64_t sc, int m, int n, int k, int batch, const float* bias, const float* rowbias,
                         int ldrb, int rows_per_sample, const void* residual, int ldr, float alpha, uint32_t flags, void* ws,
                         int64_t ws_bytes, void* stream) {
  return sdmi_gemm_ex(a, lda, sa, w, ldw, sw, out, ldc, sc, m, n, k, batch, bias, rowbias, ldrb, rows_per_sample, residual, ldr,
                      alpha, flags, ws, ws_bytes, stream, nullptr);
}

extern "C" int sdmi_gemm_ex(const void* a, int lda, int64_t sa, const void* w, int ldw, int64_t sw, void* out, int ldc,
                            int64_t sc, int m, int n, int k, int batch, const float* bias, const float* rowbias,
                            int ldrb, int rows_per_sample, const void* residual, int ldr, float alpha, uint32_t flags, void* ws,
                            int64_t ws_bytes, void* stream, const sdmi_epilogue_ext* ext) {
  SDMI_REQUIRE(a && w && out, "gemm: null pointer");
  SDMI_REQUIRE(m > 0 && n > 0 && k > 0 && batch > 0, "gemm: bad shape m=%d n=%d k=%d batch=%d", m, n, k, batch);
  SDMI_REQUIRE(k % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && lda >= k && ldw >= k, "gemm: k/lda/ldw must be multiples of 8 (k=%d lda=%d ldw=%d)", k, lda, ldw);
  SDMI_REQUIRE(aligned16(a) && aligned16(w) && aligned16(out), "gemm: pointers must be 16-byte aligned");
  SDMI_REQUIRE(sa % 8 == 0 && sw % 8 == 0 && sc % 4 == 0, "gemm: batch strides must keep 16-byte alignment");
  SDMI_REQUIRE(!(flags & SDMI_F_OUT_U8_IMAGE), "gemm: SDMI_F_OUT_U8_IMAGE is a conv3x3 output mode");
  SDMI_REQUIRE(!(flags & SDMI_F_BIAS) || bias, "gemm: SDMI_F_BIAS without bias");
  SDMI_REQUIRE(!(flags & SDMI_F_ROWBIAS) || (rowbias && rows_per_sample > 0), "gemm: SDMI_F_ROWBIAS without rowbias/rows_per_sample");
  SDMI_REQUIRE(!(flags & SDMI_F_RESIDUAL) || (residual && ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0), "gemm: bad residual");
  SDMI_REQUIRE(!(flags & SDMI_F_OUT_NCHW_F32) || (rows_per_sample > 0 && batch == 1), "gemm: NCHW output needs rows_per_sample, batch 1");
  if (flags & SDMI_F_GEGLU) {
    SDMI_REQUIRE(n % 32 == 0 && !(flags & (SDMI_F_RESIDUAL | SDMI_F_ROWBIAS | SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_F32)), "gemm: GEGLU needs n%%32==0 and no other epilogue");
    SDMI_REQUIRE(ldc % 4 == 0 && ldc >= n / 2, "gemm: GEGLU ldc");
  } else if (!(flags & SDMI_F_OUT_NCHW_F32)) {
    SDMI_REQUIRE(ldc % 4 == 0 && ldc >= n, "gemm: ldc must be a multiple of 4 and >= n (ldc=%d n=%d)", ldc, n);
  }
  IgemmParams p;
  gemm_params(p, a, lda, sa, w, ldw, sw, out, ldc, sc, m, n, k, batch, bias, rowbias, ldrb, rows_per_sample, residual, ldr, alpha, flags);
  return launch_igemm<false>(p, batch, ws, ws_bytes, (hipStream_t)stream, ext);
}

extern "C" int sdmi_gemm_stats_parts(int m, int n, int k, int batch, int rows_per_sample, uint32_t flags, int* row_parts, int* col_parts) {
  if (row_parts) *row_parts = 0;
  if (col_parts) *col_parts = 0;
  if (m <= 0 || n <= 0 || k <= 0 || batch <= 0) return SDMI_EINVAL;
  IgemmParams p;
  gemm_params(p, nullptr, k, 0, nullptr, k, 0, nullptr, n, 0, m, n, k, batch, nullptr, nullptr, 0, rows_per_sample, nullptr, 0, 1.0f, flags);
  const int vi = plan_igemm<false>(p, batch, true, (int64_t)1 << 62);
  stats_parts_of(p, batch, vi, row_parts, col_parts);
  return SDMI_OK;
}

static void conv_params(IgemmParams& p, const void* x, int ldx, int n, int h, int w, int cin, const void* w_packed, int cout,
                        int hout, int wout, int stride, int pad_t, int pad_l, const float* bias, const float* rowbias, int ldrb,
                        const void* residual, int ldr, void* out, int ldc, uint32_t flags) {
  p = IgemmParams{};
  p.a = (const u16*)x; p.w = (const u16*)w_packed; p.out = out; p.bias = bias; p.rowbias = rowbias; p.res = (const u16*)residual;
  p.lda = ldx; p.ldw = 9 * cin; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
  p.M = n * hout * wout; p.N = cout; p.K = 9 * cin; p.rows_per_sample = hout * wout; p.alpha = 1.0f; p.flags = flags;
  p.H = h; p.W = w; p.Cin = cin; p.Ho = hout; p.Wo = wout; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.ups = (flags & SDMI_F_UPSAMPLE2X) ? 1 : 0;
  p.kt_per_tap = (cin + BK - 1) / BK;
  p.nk = 9 * p.kt_per_tap;
}

extern "C" int sdmi_conv3x3(const void* x, int ldx, int n, int h, int w, int cin, const void* w_packed, int cout, int hout,
                            int wout, int stride, int pad_t, int pad_l, const float* bias, const float* rowbias, int ldrb,
                            const void* residual, int ldr, void* out, int ldc, uint32_t flags, void* ws, int64_t ws_bytes,
                            void* stream) {
  return sdmi_conv3x3_ex(x, ldx, n, h, w, cin, w_packed, cout, hout, wout, stride, pad_t, pad_l, bias, rowbias, ldrb, residual, ldr,
                         out, ldc, flags, ws, ws_bytes, stream, nullptr);
}

extern "C" int sdmi_conv3x3_stats_parts(int n, int h, int w, int cin, int cout, int hout, int wout, int stride, int pad_t, int pad_l,
                                        uint32_t flags, int* row_parts) {
  if (row_parts) *row_parts = 0;
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || hout <= 0 || wout <= 0) return SDMI_EINVAL;
  IgemmParams p;
  conv_params(p, nullptr, cin, n, h, w, cin, nullptr, cout, hout, wout, stride, pad_t, pad_l, nullptr, nullptr, 0, nullptr, 0, nullptr, cout, flags);
  const int vi = plan_igemm<true>(p, 1, true, (int64_t)1 << 62);
  stats_parts_of(p, 1, vi, row_parts, nullptr);
  return SDMI_OK;
}

extern "C" int sdmi_conv3x3_gn_in_ok(int n, int h, int w, int cin, int cout, int ldx) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ldx < cin) return 0;
  IgemmParams p;
  conv_params(p, nullptr, ldx, n, h, w, cin, nullptr, cout, h, w, 1, 1, 1, nullptr, nullptr, 0, nullptr, 0, nullptr, cout, SDMI_F_SILU);
  const int vi = plan_igemm<true>(p, 1, true, (int64_t)1 << 62);
  return gn_in_ok(p, vi, 1) ? 1 : 0;
}

extern "C" int sdmi_conv3x3_gn_out_ok(int n, int h, int w, int cin, int cout, int hout, int wout, int stride, int pad_t, int pad_l, uint32_t flags,
                                      int groups) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || hout <= 0 || wout <= 0) return 0;
  IgemmParams p;
  conv_params(p, nullptr, cin, n, h, w, cin, nullptr, cout, hout, wout, stride, pad_t, pad_l, nullptr, nullptr, 0, nullptr, 0, nullptr, cout, flags);
  (void)plan_igemm<true>(p, 1, true, (int64_t)1 << 62);
  return gn_out_shape_ok(p, 1, groups) ? 1 : 0;
}

extern "C" int sdmi_conv3x3_ex(const void* x, int ldx, int n, int h, int w, int cin, const void* w_packed, int cout, int hout,
                               int wout, int stride, int pad_t, int pad_l, const float* bias, const float* rowbias, int ldrb,
                               const void* residual, int ldr, void* out, int ldc, uint32_t flags, void* ws, int64_t ws_bytes,
                               void* stream, const sdmi_epilogue_ext* ext) {
  SDMI_REQUIRE(x && w_packed && (out || (ext && ext->gn_out && !ext->gn_out->keep_raw)), "conv3x3: null pointer");
  SDMI_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && hout > 0 && wout > 0, "conv3x3: bad shape");
  SDMI_REQUIRE(cin % 8 == 0 && ldx % 8 == 0 && ldx >= cin, "conv3x3: cin and ldx must be multiples of 8 (cin=%d ldx=%d)", cin, ldx);
  SDMI_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
  SDMI_REQUIRE(aligned16(x) && aligned16(w_packed) && aligned16(out), "conv3x3: pointers must be 16-byte aligned");
  const int ups = (flags & SDMI_F_UPSAMPLE2X) ? 1 : 0;
  // every tap of every output pixel must fall inside the zero-padded (upsampled) input by at most the pad
  SDMI_REQUIRE((hout - 1) * stride + 2 - pad_t <= (h << ups) && (wout - 1) * stride + 2 - pad_l <= (w << ups) && pad_t >= 0 && pad_l >= 0 && pad_t <= 1 && pad_l <= 1,
               "conv3x3: geometry out of range (h=%d w=%d hout=%d wout=%d stride=%d pad=%d,%d ups=%d)", h, w, hout, wout, stride, pad_t, pad_l, ups);
  SDMI_REQUIRE(!(flags & SDMI_F_BIAS) || bias, "conv3x3: SDMI_F_BIAS without bias");
  SDMI_REQUIRE(!(flags & SDMI_F_ROWBIAS) || rowbias, "conv3x3: SDMI_F_ROWBIAS without rowbias");
  SDMI_REQUIRE(!(flags & SDMI_F_RESIDUAL) || (residual && ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0), "conv3x3: bad residual");
  SDMI_REQUIRE(!(flags & (SDMI_F_GEGLU | SDMI_F_OUT_F32)), "conv3x3: unsupported flag");
  SDMI_REQUIRE(!(flags & SDMI_F_OUT_U8_IMAGE) || !(flags & SDMI_F_OUT_NCHW_F32), "conv3x3: two output layouts requested");
  if (!(flags & (SDMI_F_OUT_NCHW_F32 | SDMI_F_OUT_U8_IMAGE))) SDMI_REQUIRE(ldc % 4 == 0 && ldc >= cout, "conv3x3: ldc");
  SDMI_REQUIRE(!(flags & SDMI_F_LNFOLD), "conv3x3: SDMI_F_LNFOLD is a gemm mode");
  IgemmParams p;
  conv_params(p, x, ldx, n, h, w, cin, w_packed, cout, hout, wout, stride, pad_t, pad_l, bias, rowbias, ldrb, residual, ldr, out, ldc, flags);
  return launch_igemm<true>(p, 1, ws, ws_bytes, (hipStream_t)stream, ext);
}

extern "C" int sdmi_ws_tickets_init(void* ws, void* stream) {
  (void)stream;
  SDMI_REQUIRE(ws, "ws_tickets_init: null workspace");
  bool full = false;
  if (!ticket_bank(ws, true, &full)) {
    sdmi_set_error(full ? "ws_tickets_init: all %d counter banks of this device are taken (release one with sdmi_ws_tickets_release)"
                        : "ws_tickets_init: hipGetSymbolAddress(g_sk_ticket) failed", SK_TICKET_BANKS);
    return full ? SDMI_EINVAL : SDMI_ELAUNCH;
  }
  return 0;
}

extern "C" int sdmi_ws_tickets_release(void* ws) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  std::lock_guard<std::mutex> lock(g_banks.mu);
  for (int b = 0; b < SK_TICKET_BANKS; ++b)
    if (g_banks.owner[dev][b] == ws) { g_banks.owner[dev][b] = nullptr; return 0; }
  sdmi_set_error("ws_tickets_release: workspace is not registered on this device");
  return SDMI_EINVAL;
}

extern "C" int64_t sdmi_gemm_ws_bytes(int m, int n, int k, int batch, int rows_per_sample, uint32_t flags) {
  if (m <= 0 || n <= 0 || k <= 0 || batch <= 0) return 0;
  return splitk_ws_bytes(m, n, (k + BK - 1) / BK, batch, rows_per_sample > 0 ? rows_per_sample : m, flags, false);
}

extern "C" int64_t sdmi_conv3x3_ws_bytes(int n, int cin, int cout, int hout, int wout) {
  if (n <= 0 || cin <= 0 || cout <= 0 || hout <= 0 || wout <= 0) return 0;
  return splitk_ws_bytes(n * hout * wout, cout, 9 * ((cin + BK - 1) / BK), 1, hout * wout, 0, true);
}
