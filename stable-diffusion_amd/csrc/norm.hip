// norm.hip - HBM-bound normalisation kernels on NHWC bf16 activations (16-byte vector loads/stores).
//   GroupNorm(32)(+SiLU)  nn.GroupNorm + nn.SiLU  (layers/ResBlock.py:65-66,:72-73; Unet.py:104-106;
//                         AutoencoderKL.py:193-194; layers/SpatialTransformer.py:134,:212)
//   LayerNorm             nn.LayerNorm            (layers/SpatialTransformer.py:177-184)
//   row softmax           the softmax inside the VAE's single-head attention (:151)
// All reductions run in a fixed order (no float atomics): results are bit-reproducible run to run and
// GPU to GPU, which the image-sharded multi-GPU path relies on.
#include <stdlib.h>
#include "common.h"

// ------------------------------------------------------------------------------------ GroupNorm
__device__ __attribute__((aligned(16))) unsigned g_norm_zero16[4] = {0u, 0u, 0u, 0u};

// Lane mapping shared by the stats and apply kernels: a wave covers RPP = 64/LPR rows per pass, LPR = lanes per row
// (64 when C/8 >= 64, else the next power of two >= C/8), lane -> (row sub-index lane/LPR, 16-byte chunk lane%LPR
// [+64*s]).  A lane always owns the same channels, so per-channel state (sums, scale/shift) lives in registers.

// fold the per-block partials of sample n (fp64, fixed order) into per-channel scale/shift; one 256-thread block
__device__ __forceinline__ void gn_fold(const float* partial, int nblk, int groups, int c, int hw,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        float* __restrict__ scale_shift, int n, float* s_mean, float* s_rstd) {
  const int t = threadIdx.x;
  const int g = t >> 3, sub = t & 7;  // 8 threads per group, 32 groups per pass
  for (int g0 = 0; g0 < groups; g0 += 32) {
    const int gg = g0 + g;
    double s0 = 0.0, s1 = 0.0;
    if (gg < groups)
      for (int b = sub; b < nblk; b += 8) {
        const float2 pp = *(const float2*)(partial + (((long long)n * nblk + b) * groups + gg) * 2);
        s0 += (double)pp.x;
        s1 += (double)pp.y;
      }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (sub == 0 && gg < groups) {
      const double cnt = (double)hw * (double)(c / groups);
      const double mean = s0 / cnt;
      double var = s1 / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[gg & 63] = (float)mean;
      s_rstd[gg & 63] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int cpg = c / groups;
  for (int ci = t; ci < c; ci += 256) {
    const int gg = ci / cpg;
    const float sc = gamma[ci] * s_rstd[gg & 63];
    float2 o;
    o.x = sc;
    o.y = beta[ci] - s_mean[gg & 63] * sc;
    *(float2*)(scale_shift + ((long long)n * c + ci) * 2) = o;
  }
}


// pass 1a: per-block partial (sum, sumsq) of every group.  grid (nblk, N).  Per-channel sums in registers, rows
// sub-indices folded by shuffles, the 4 waves combined through LDS in wave order, thread g folds group g: fixed
// order everywhere (bit-reproducible).
template <int NS>
__global__ __launch_bounds__(256) void gn_partial_kernel(const u16* __restrict__ x, int hw, int c, int ldx,
                                                         int rows_per_block, int cpg, int lpr_shift,
                                                         float* partial, unsigned* ticket,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ scale_shift) {
  __shared__ float s_sum[2560], s_sq[2560];
  __shared__ int s_last;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int blk = blockIdx.x, n = blockIdx.y, nblk = gridDim.x;
  const int cc = c >> 3;
  const int lpr = 1 << lpr_shift, rpp = 64 >> lpr_shift, rsub = lane >> lpr_shift, chl = lane & (lpr - 1);
  const int row_beg = blk * rows_per_block;
  const int row_end = min(hw, row_beg + rows_per_block);
  float a[NS][8], b[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[s][e] = b[s][e] = 0.f;
  const u16* base = x + (long long)n * hw * ldx;
  const u16* const zp = (const u16*)g_norm_zero16;
  // 4 passes per wave in flight per iteration; rows past the block / chunks past C read a zero page (adds 0)
  for (int row = row_beg + wv * rpp + rsub; row < row_end; row += 16 * rpp) {
    u32x4 v[4][NS];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = row + 4 * rpp * j;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ch = chl + 64 * s;
        v[j][s] = *(const u32x4*)((rr < row_end && ch < cc) ? base + (long long)rr * ldx + ch * 8 : zp);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float f[8];
        unpack8(make_uint4(v[j][s][0], v[j][s][1], v[j][s][2], v[j][s][3]), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[s][e] += f[e]; b[s][e] += f[e] * f[e]; }
      }
  }
  // fold the row sub-indices (lanes chl, chl+LPR, ...) : butterfly over the high lane bits
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[s][e] += __shfl_xor(a[s][e], o); b[s][e] += __shfl_xor(b[s][e], o); }
  }
  for (int w = 0; w < 4; ++w) {
    if (wv == w && rsub == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ch = chl + 64 * s;
        if (ch < cc) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ci = ch * 8 + e;
            if (w == 0) { s_sum[ci] = a[s][e]; s_sq[ci] = b[s][e]; }
            else { s_sum[ci] += a[s][e]; s_sq[ci] += b[s][e]; }
          }
        }
      }
    }
    __syncthreads();
  }
  const int groups = c / cpg;
  if (t < groups) {
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < cpg; ++i) { s0 += s_sum[t * cpg + i]; s1 += s_sq[t * cpg + i]; }
    float* o = partial + (((long long)n * nblk + blk) * groups + t) * 2;
    if (ticket) {  // write-through (agent-scope) stores: visible to the folding block without an L2 write-back fence
      __hip_atomic_store(o, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      o[0] = s0;
      o[1] = s1;
    }
  }
  if (!ticket) return;  // gn_finalize_kernel follows as its own launch
  // Last-arriver fold: the block that draws the last ticket of its sample turns the partials into scale/shift (what
  // gn_finalize_kernel does as a separate launch) and re-arms the ticket for the next call.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    const unsigned prev = __hip_atomic_fetch_add(ticket + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == (unsigned)nblk - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  gn_fold(partial, nblk, groups, c, hw, gamma, beta, eps, scale_shift, n, s_sum, s_sq);
}

// pass 1b: fold the partials (fp64), emit per-(sample,channel) scale/shift.  grid N, 256 threads.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nblk, int groups, int c,
                                                          int hw, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale_shift) {
  __shared__ float s_mean[64], s_rstd[64];
  gn_fold(partial, nblk, groups, c, hw, gamma, beta, eps, scale_shift, blockIdx.x, s_mean, s_rstd);
}
// pass 2: y = act(x*scale + shift).  grid (nblk, N); same lane mapping; the lane's scale/shift stay in registers
// for all its rows, two passes per wave in flight.
template <int NS>
__global__ __launch_bounds__(256) void gn_apply_kernel(const u16* __restrict__ x, int ldx, const float* __restrict__ ss,
                                                       u16* __restrict__ y, int ldy, int hw, int c, int rows_per_block,
                                                       int lpr_shift, int silu) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n = blockIdx.y;
  const int cc = c >> 3;
  const int lpr = 1 << lpr_shift, rpp = 64 >> lpr_shift, rsub = lane >> lpr_shift, chl = lane & (lpr - 1);
  const int row_beg = blockIdx.x * rows_per_block;
  const int row_end = min(hw, row_beg + rows_per_block);
  float sc[NS][8], sh[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int ch = chl + 64 * s;
    if (ch < cc) {
      const float4* sp = (const float4*)(ss + ((long long)n * c + ch * 8) * 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 v = sp[e];
        sc[s][2 * e] = v.x; sh[s][2 * e] = v.y; sc[s][2 * e + 1] = v.z; sh[s][2 * e + 1] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[s][e] = 0.f; sh[s][e] = 0.f; }
    }
  }
  const u16* xb = x + (long long)n * hw * ldx;
  u16* yb = y + (long long)n * hw * ldy;
  const u16* const zp = (const u16*)g_norm_zero16;
  for (int row = row_beg + wv * rpp + rsub; row < row_end; row += 8 * rpp) {
    u32x4 v[2][NS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = row + 4 * rpp * j;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ch = chl + 64 * s;
        v[j][s] = *(const u32x4*)((rr < row_end && ch < cc) ? xb + (long long)rr * ldx + ch * 8 : zp);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = row + 4 * rpp * j;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ch = chl + 64 * s;
        float f[8];
        unpack8(make_uint4(v[j][s][0], v[j][s][1], v[j][s][2], v[j][s][3]), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float o = f[e] * sc[s][e] + sh[s][e];
          f[e] = silu ? silu_f(o) : o;
        }
        if (rr < row_end && ch < cc) *(uint4*)(yb + (long long)rr * ldy + ch * 8) = pack8(f);
      }
    }
  }
}

// Tickets of the last-arriver fold: zero at load, re-armed by the folding block.  64 banks of 256 samples, bank chosen
// from the workspace address, so GroupNorm calls that overlap in time on one device (different streams) must use
// different workspaces - calls on one stream, or otherwise ordered, can share anything.
constexpr int GN_TICKET_SAMPLES = 256, GN_TICKET_BANKS = 64;
__device__ unsigned g_gn_ticket[GN_TICKET_BANKS * GN_TICKET_SAMPLES];

static int gn_nblk(int hw) {
  // 64 rows per block, at most 256 blocks per sample: the fold is paid per arrival (one ticket atomic per block on one
  // counter, and the folding block's serial pass over the partials): 288 arrivals per sample cost more than they bought
  int nb = (hw + 63) / 64;
  return nb < 1 ? 1 : (nb > 256 ? 256 : nb);
}
static int gn_lpr_shift(int c) {
  const int cc = c / 8;
  int sh = 0;
  while ((1 << sh) < cc && sh < 6) ++sh;
  return sh;
}

extern "C" int64_t sdmi_groupnorm_ws_bytes(int n, int hw, int c) {
  (void)c;
  return (int64_t)n * gn_nblk(hw) * 64 * 2 * sizeof(float);
}

extern "C" int sdmi_groupnorm_stats(const void* x, int n, int hw, int c, int ldx, int groups, const float* gamma,
                                    const float* beta, float eps, float* scale_shift, void* ws, void* stream) {
  SDMI_REQUIRE(x && gamma && beta && scale_shift && ws, "groupnorm_stats: null pointer");
  SDMI_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldx >= c, "groupnorm_stats: c and ldx must be multiples of 8 (c=%d ldx=%d)", c, ldx);
  SDMI_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "groupnorm_stats: groups=%d does not divide c=%d", groups, c);
  SDMI_REQUIRE(c <= 2560 && n <= 65535, "groupnorm_stats: c=%d > 2560 or n too large", c);
  SDMI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)scale_shift & 15) == 0 && ((uintptr_t)ws & 7) == 0, "groupnorm_stats: alignment");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = gn_nblk(hw);
  const int rpb = (hw + nblk - 1) / nblk;
  const int ns = (c / 8 + 63) / 64;
  const int lsh = gn_lpr_shift(c);
  dim3 grid(nblk, n), block(256);
  const u16* xp = (const u16*)x;
  float* part = (float*)ws;
  unsigned* ticket = nullptr;
  // SDMI_GN_FOLD=1: last-arriver fold inside the partial kernel instead of the finalize launch.  Off: measured equal or
  // slightly slower in graph replay (1.09 vs 1.07 ms of GroupNorm per UNet step) - a dependent launch costs ~1.7 us here,
  // the ticket fan-in plus the acquire cost as much.
  static int fold_in_kernel = -1;
  if (fold_in_kernel < 0) { const char* e = getenv("SDMI_GN_FOLD"); fold_in_kernel = (e && e[0] == '1') ? 1 : 0; }
  if (fold_in_kernel && n <= GN_TICKET_SAMPLES) {
    static void* sym_of_device[64] = {};  // every device holds its own copy of the array at its own address
    int dev = 0;
    (void)hipGetDevice(&dev);
    void*& sym = sym_of_device[dev & 63];
    if (!sym && (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_gn_ticket)) != hipSuccess || !sym)) {
      sym = nullptr;
      sdmi_set_error("groupnorm_stats: hipGetSymbolAddress(g_gn_ticket) failed");
      return SDMI_ELAUNCH;
    }
    ticket = (unsigned*)sym + (((uintptr_t)ws >> 8) % GN_TICKET_BANKS) * GN_TICKET_SAMPLES;
  }
#define GN_PARTIAL(NS) hipLaunchKernelGGL(gn_partial_kernel<NS>, grid, block, 0, st, xp, hw, c, ldx, rpb, c / groups, lsh, part, ticket, gamma, beta, eps, scale_shift)
  switch (ns) {
    case 1: GN_PARTIAL(1); break;
    case 2: GN_PARTIAL(2); break;
    case 3: GN_PARTIAL(3); break;
    case 4: GN_PARTIAL(4); break;
    default: GN_PARTIAL(5); break;
  }
#undef GN_PARTIAL
  int rc = sdmi_check_launch("groupnorm partial");
  if (rc || ticket) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n), block, 0, st, part, nblk, groups, c, hw, gamma, beta, eps, scale_shift);
  return sdmi_check_launch("groupnorm finalize");
}

extern "C" int sdmi_groupnorm_apply(const void* x, int ldx, const float* scale_shift, void* y, int ldy, int n, int hw,
                                    int c, uint32_t flags, void* stream) {
  SDMI_REQUIRE(x && scale_shift && y, "groupnorm_apply: null pointer");
  SDMI_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c, "groupnorm_apply: c/ldx/ldy must be multiples of 8");
  SDMI_REQUIRE(c <= 2560 && n <= 65535, "groupnorm_apply: c=%d > 2560 or n too large", c);
  SDMI_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)scale_shift) & 15) == 0, "groupnorm_apply: alignment");
  hipStream_t st = (hipStream_t)stream;
  const int lsh = gn_lpr_shift(c);
  const int rpp = 64 >> lsh;
  // 2 passes per wave, both in flight (8*rpp rows per block): many small blocks keep enough bytes in flight per CU;
  // very large tensors get longer blocks (<= 32768 per sample; round 3: the cap was 8192, and the 768x768 levels of the VAE ran
  // 96-row blocks at 4.9 TB/s where 32-row blocks reach 5.4 - profiles/r03_gn_apply_bw.txt, option gn_apply_rows)
  int rpb = 8 * rpp;
  if ((hw + rpb - 1) / rpb > 32768) rpb = ((hw + 32767) / 32768 + 8 * rpp - 1) / (8 * rpp) * (8 * rpp);
  if (sdmi_opt(OPT_gn_apply_rows) > 0) rpb = sdmi_opt(OPT_gn_apply_rows) * 8 * rpp;
  dim3 grid((hw + rpb - 1) / rpb, n), block(256);
  const int ns = (c / 8 + 63) / 64;
  const int silu = (flags & SDMI_F_SILU) ? 1 : 0;
  const u16* xp = (const u16*)x;
  u16* yp = (u16*)y;
  switch (ns) {
    case 1: hipLaunchKernelGGL(gn_apply_kernel<1>, grid, block, 0, st, xp, ldx, scale_shift, yp, ldy, hw, c, rpb, lsh, silu); break;
    case 2: hipLaunchKernelGGL(gn_apply_kernel<2>, grid, block, 0, st, xp, ldx, scale_shift, yp, ldy, hw, c, rpb, lsh, silu); break;
    case 3: hipLaunchKernelGGL(gn_apply_kernel<3>, grid, block, 0, st, xp, ldx, scale_shift, yp, ldy, hw, c, rpb, lsh, silu); break;
    case 4: hipLaunchKernelGGL(gn_apply_kernel<4>, grid, block, 0, st, xp, ldx, scale_shift, yp, ldy, hw, c, rpb, lsh, silu); break;
    default: hipLaunchKernelGGL(gn_apply_kernel<5>, grid, block, 0, st, xp, ldx, scale_shift, yp, ldy, hw, c, rpb, lsh, silu); break;
  }
  return sdmi_check_launch("groupnorm apply");
}


// ------------------------------------------------------------------------------------ fused GroupNorm (small tensors)
// One 1024-thread block owns one (sample, group): the group's elements (hw x cpg) sit in registers between the
// statistics and the normalisation, so the tensor is read once and written once by ONE launch (the three-kernel path
// reads it twice and pays three dependent launches: 13 us on a 12x12 level, 22 us on 96x96x320, against 4-9 us here).
// Two-pass variance from registers (exact mean first), fixed reduction order.  4-byte accesses: a group's slice of a
// row is cpg*2 = 20...160 bytes at a 4-byte-aligned offset.
template <int DPT>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const u16* __restrict__ x, int ldx, u16* __restrict__ y, int ldy,
                                                        int hw, int cpg, int groups, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int silu) {
  __shared__ float s_red[16];
  __shared__ float s_sc[160], s_sh[160];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // blocks b and b+8 share an XCD: give an XCD neighbouring groups (they share the 128-byte lines of a row)
  int g, n;
  {
    const int bid = blockIdx.x, nsamp = gridDim.x / groups;
    if ((groups & 7) == 0) {
      const int gpx = groups >> 3, xcd = bid & 7, k = bid >> 3;
      g = xcd * gpx + k % gpx;
      n = k / gpx;
    } else {
      g = bid % groups;
      n = bid / groups;
    }
    (void)nsamp;
  }
  const int dpr = cpg >> 1;  // dwords (channel pairs) per row of this group
  const int total = hw * dpr;
  const int qstep = 1024 / dpr, rstep = 1024 - qstep * dpr;
  const u16* xb = x + (long long)n * hw * ldx + g * cpg;
  u16* yb = y + (long long)n * hw * ldy + g * cpg;
  const unsigned* const zp = (const unsigned*)g_norm_zero16;
  unsigned v[DPT];
  {
    int row = t / dpr, d = t - row * dpr;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const bool ok = k * 1024 + t < total;
      v[k] = *(ok ? (const unsigned*)(xb + (long long)row * ldx + 2 * d) : zp);
      d += rstep; row += qstep;
      if (d >= dpr) { d -= dpr; ++row; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < DPT; ++k) { float a, b; unpack2(v[k], a, b); s += a + b; }   // masked slots hold zeros
  s = wave_sum(s);
  if (lane == 0) s_red[wv] = s;
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
    ss += (k * 1024 + t < total) ? __builtin_fmaf(db, db, da * da) : 0.f;   // explicit: igemm_splitk_reduce_gn_kernel repeats it bit for bit
  }
  ss = wave_sum(ss);
  if (lane == 0) s_red[wv] = ss;
  __syncthreads();
  float vtot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) vtot += s_red[w];
  const float rstd = 1.0f / sqrtf(vtot / cnt + eps);
  if (t < cpg) {
    const float sc = gamma[g * cpg + t] * rstd;
    s_sc[t] = sc;
    s_sh[t] = __builtin_fmaf(-mean, sc, beta[g * cpg + t]);
  }
  __syncthreads();
  {
    // (row, d) are re-derived from an opaque copy of t: otherwise the compiler keeps all DPT load addresses alive
    // for the stores (3 extra registers per element: spills from DPT = 24 up)
    int t2 = t;
    asm volatile("" : "+v"(t2));
    int row = t2 / dpr, d = t2 - row * dpr;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      float a, b; unpack2(v[k], a, b);
      float oa = __builtin_fmaf(a, s_sc[2 * d], s_sh[2 * d]), ob = __builtin_fmaf(b, s_sc[2 * d + 1], s_sh[2 * d + 1]);
      if (silu) { oa = silu_f(oa); ob = silu_f(ob); }
      if (k * 1024 + t < total) *(unsigned*)(yb + (long long)row * ldy + 2 * d) = pack2h(oa, ob);
      d += rstep; row += qstep;
      if (d >= dpr) { d -= dpr; ++row; }
    }
  }
}

// GroupNorm(+SiLU) in one call: the fused kernel when a (sample, group) fits a block's registers, else stats + apply.
extern "C" int sdmi_groupnorm(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups, const float* gamma,
                              const float* beta, float eps, uint32_t flags, float* scale_shift, void* ws, void* stream) {
  SDMI_REQUIRE(x && y && gamma && beta, "groupnorm: null pointer");
  SDMI_REQUIRE(n > 0 && hw > 0 && c > 0 && groups > 0 && c % groups == 0, "groupnorm: bad shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  const int cpg = c / groups;
  static int fused_ok = -1;  // SDMI_GN_FUSED=0 disables the fused kernel (A/B switch)
  if (fused_ok < 0) { const char* e = getenv("SDMI_GN_FUSED"); fused_ok = (e && e[0] == '0') ? 0 : 1; }
  const long long dwords = (long long)hw * (cpg / 2);
  // measured (graph replay, 2 samples): fused 5.9 vs 13.3 us on 12x12x1280, 9.4 vs 14.4 on 24x24x1280, 16.2 vs 20.8 on
  // 24x24x2560; but 16.9 vs 15.0 on 48x48x640 and 45 vs 21 on 96x96x320 (only N*32 blocks, short row segments)
  const bool small = dwords <= 12 * 1024 || (dwords <= 24 * 1024 && cpg >= 40);
  if (fused_ok && small && cpg % 2 == 0 && cpg <= 160 && ldx % 2 == 0 && ldy % 2 == 0 && (long long)n * groups <= 65535 &&
      (((uintptr_t)x | (uintptr_t)y) & 3) == 0) {
    hipStream_t st = (hipStream_t)stream;
    const int dpt = (int)((dwords + 1023) / 1024);
    dim3 grid(n * groups), block(1024);
    const int silu = (flags & SDMI_F_SILU) ? 1 : 0;
    const u16* xp = (const u16*)x;
    u16* yp = (u16*)y;
#define GN_FUSED(D) hipLaunchKernelGGL(gn_fused_kernel<D>, grid, block, 0, st, xp, ldx, yp, ldy, hw, cpg, groups, gamma, beta, eps, silu)
    if (dpt <= 6) GN_FUSED(6);
    else if (dpt <= 12) GN_FUSED(12);
    else GN_FUSED(24);
#undef GN_FUSED
    return sdmi_check_launch("groupnorm fused");
  }
  SDMI_REQUIRE(scale_shift && ws, "groupnorm: this shape needs the scale_shift and workspace buffers");
  int rc = sdmi_groupnorm_stats(x, n, hw, c, ldx, groups, gamma, beta, eps, scale_shift, ws, stream);
  if (rc) return rc;
  return sdmi_groupnorm_apply(x, ldx, scale_shift, y, ldy, n, hw, c, flags, stream);
}

// ------------------------------------------------------------------------------------ GroupNorm from producer statistics
// The conv / GEMM that produced x already left per-channel (sum, sumsq) of its fp32 results per wave tile of rows
// (sdmi_epilogue_ext.colstats, csrc/igemm.hip), so x is not read again for its statistics: one small kernel folds them
// per (sample, group) into scale/shift (fp64, fixed order), then the apply pass runs.  Up to two channel segments: the
// input of an up-path ResBlock is the concat of two producers' tensors (Unet.py:132), and a group may straddle them.
struct GnSegs {
  const float* stats[2];
  int parts[2], ld[2], col0[2], c0[2], c[2];
  int nseg;
};
constexpr int GNF_THREADS = 1024, GNF_WAVES = GNF_THREADS / 64, GNF_CHUNK = 4 * GNF_THREADS, GNF_MAX_CHUNKS = 64;
// chunks: gridDim.z > 1 splits a (sample, group)'s partials over blocks (the 768x768 levels of the VAE have 18432 row parts);
// each block then leaves its fp64 (sum, sumsq) in `chunk_out` and gn_finalize_chunks_kernel adds them in chunk order.
__global__ __launch_bounds__(GNF_THREADS) void gn_finalize_cols_kernel(GnSegs sg, int groups, int c, int hw, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float eps,
                                                                       float* __restrict__ scale_shift, double* __restrict__ chunk_out) {
  __shared__ double red[2][GNF_WAVES];
  __shared__ float s_mr[2];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int g = blockIdx.x, n = blockIdx.y;
  const int cpg = c / groups;
  const int maxparts = sg.nseg > 1 ? max(sg.parts[0], sg.parts[1]) : sg.parts[0];
  const int total = cpg * maxparts;
  const int per_z = ((total + gridDim.z - 1) / gridDim.z + GNF_CHUNK - 1) / GNF_CHUNK * GNF_CHUNK;
  const int lo = blockIdx.z * per_z, hi = min(total, lo + per_z);
  const float* const zp = (const float*)g_norm_zero16;
  double s0 = 0.0, s1 = 0.0;
  // four loads in flight per thread and iteration (a (sample, group) of 96x96x320 is 2880 partials: one iteration)
  for (int base = lo; base < hi; base += GNF_CHUNK) {
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = base + j * GNF_THREADS + t;
      const int part = idx / cpg, ch = g * cpg + (idx - part * cpg);
      const int sgi = (sg.nseg > 1 && ch >= sg.c0[1]) ? 1 : 0;
      const bool ok = idx < hi && part < sg.parts[sgi];
      v[j] = *(const float2*)(ok ? sg.stats[sgi] + (((long long)n * sg.parts[sgi] + part) * sg.ld[sgi] + sg.col0[sgi] + (ch - sg.c0[sgi])) * 2 : zp);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0 += (double)v[j].x; s1 += (double)v[j].y; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  if (lane == 0) { red[0][wv] = s0; red[1][wv] = s1; }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < GNF_WAVES; ++w) { a += red[0][w]; b += red[1][w]; }
    if (gridDim.z > 1) {
      double* o = chunk_out + (((long long)n * groups + g) * gridDim.z + blockIdx.z) * 2;
      o[0] = a;
      o[1] = b;
    }
    const double cnt = (double)hw * (double)cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mr[0] = (float)mean;
    s_mr[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  if (gridDim.z > 1) return;
  __syncthreads();
  if (t < cpg) {
    const int ci = g * cpg + t;
    const float sc = gamma[ci] * s_mr[1];
    *(float2*)(scale_shift + ((long long)n * c + ci) * 2) = make_float2(sc, beta[ci] - s_mr[0] * sc);
  }
}
// second level of a chunked fold: block (group, sample) adds the chunks in order and writes its channels' scale/shift
__global__ __launch_bounds__(256) void gn_finalize_chunks_kernel(const double* __restrict__ chunks, int nchunks, int groups, int c, int hw,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, float* __restrict__ scale_shift) {
  const int t = threadIdx.x, g = blockIdx.x, n = blockIdx.y;
  const int cpg = c / groups;
  const double* in = chunks + ((long long)n * groups + g) * nchunks * 2;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nchunks; ++k) { a += in[2 * k]; b += in[2 * k + 1]; }   // every thread, same order: no exchange needed
  const double cnt = (double)hw * (double)cpg;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
  if (t < cpg) {
    const int ci = g * cpg + t;
    const float sc = gamma[ci] * fr;
    *(float2*)(scale_shift + ((long long)n * c + ci) * 2) = make_float2(sc, beta[ci] - fm * sc);
  }
}
extern "C" int64_t sdmi_groupnorm_colstats_ss_bytes(int n, int c, int groups) {
  // scale/shift table + room for the chunked fold of very long partial lists (fp64 pairs)
  return ((int64_t)n * c * 2 * (int64_t)sizeof(float) + 15) / 16 * 16 + (int64_t)n * groups * GNF_MAX_CHUNKS * 2 * (int64_t)sizeof(double);
}

extern "C" int sdmi_groupnorm_colstats_finalize(int n, int hw, int c, int groups, const float* gamma, const float* beta, float eps, int nseg,
                                                const float* const* seg_stats, const int* seg_parts, const int* seg_ld,
                                                const int* seg_col0, const int* seg_c0, const int* seg_c, float* scale_shift,
                                                void* stream) {
  SDMI_REQUIRE(gamma && beta && scale_shift && seg_stats && seg_parts && seg_ld && seg_col0 && seg_c0 && seg_c,
               "groupnorm_from_colstats: null pointer");
  SDMI_REQUIRE(((uintptr_t)scale_shift & 15) == 0, "groupnorm_from_colstats: scale_shift must be 16-byte aligned");
  SDMI_REQUIRE(n > 0 && hw > 0 && c > 0 && groups > 0 && c % groups == 0 && n <= 65535, "groupnorm_from_colstats: bad shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  SDMI_REQUIRE(c / groups <= 256, "groupnorm_from_colstats: more than 256 channels per group");
  SDMI_REQUIRE(nseg == 1 || nseg == 2, "groupnorm_from_colstats: 1 or 2 segments");
  GnSegs sg = {};
  sg.nseg = nseg;
  int next = 0;
  for (int i = 0; i < nseg; ++i) {
    SDMI_REQUIRE(seg_stats[i] && ((uintptr_t)seg_stats[i] & 7) == 0 && seg_parts[i] > 0 && seg_c[i] > 0 && seg_col0[i] >= 0 &&
                     seg_ld[i] >= seg_col0[i] + seg_c[i],
                 "groupnorm_from_colstats: bad segment %d", i);
    SDMI_REQUIRE(seg_c0[i] == next, "groupnorm_from_colstats: segments must tile the channels in order (segment %d starts at %d, expected %d)", i, seg_c0[i], next);
    next += seg_c[i];
    sg.stats[i] = seg_stats[i]; sg.parts[i] = seg_parts[i]; sg.ld[i] = seg_ld[i]; sg.col0[i] = seg_col0[i]; sg.c0[i] = seg_c0[i]; sg.c[i] = seg_c[i];
  }
  SDMI_REQUIRE(next == c, "groupnorm_from_colstats: segments cover %d of %d channels", next, c);
  int maxparts = sg.parts[0];
  if (nseg > 1 && sg.parts[1] > maxparts) maxparts = sg.parts[1];
  const long long total = (long long)(c / groups) * maxparts;
  SDMI_REQUIRE(total < (1ll << 31), "groupnorm_from_colstats: too many partials");
  int nchunks = (int)((total + 4 * GNF_CHUNK - 1) / (4 * GNF_CHUNK));   // a block folds up to four iterations' worth alone
  if (nchunks > GNF_MAX_CHUNKS) nchunks = GNF_MAX_CHUNKS;
  if (nchunks < 1) nchunks = 1;
  double* chunk_ws = (double*)((char*)scale_shift + ((int64_t)n * c * 2 * (int64_t)sizeof(float) + 15) / 16 * 16);
  hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(groups, n, nchunks), dim3(GNF_THREADS), 0, (hipStream_t)stream, sg, groups, c, hw, gamma,
                     beta, eps, scale_shift, chunk_ws);
  int rc = sdmi_check_launch("groupnorm finalize (colstats)");
  if (rc) return rc;
  if (nchunks > 1) {
    hipLaunchKernelGGL(gn_finalize_chunks_kernel, dim3(groups, n), dim3(256), 0, (hipStream_t)stream, chunk_ws, nchunks, groups, c, hw, gamma,
                       beta, eps, scale_shift);
    rc = sdmi_check_launch("groupnorm finalize (chunks)");
    if (rc) return rc;
  }
  return SDMI_OK;
}

extern "C" int sdmi_groupnorm_from_colstats(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups,
                                            const float* gamma, const float* beta, float eps, uint32_t flags, int nseg,
                                            const float* const* seg_stats, const int* seg_parts, const int* seg_ld,
                                            const int* seg_col0, const int* seg_c0, const int* seg_c, float* scale_shift,
                                            void* stream) {
  SDMI_REQUIRE(x && y, "groupnorm_from_colstats: null pointer");
  const int rc = sdmi_groupnorm_colstats_finalize(n, hw, c, groups, gamma, beta, eps, nseg, seg_stats, seg_parts, seg_ld, seg_col0, seg_c0,
                                                  seg_c, scale_shift, stream);
  if (rc) return rc;
  return sdmi_groupnorm_apply(x, ldx, scale_shift, y, ldy, n, hw, c, flags, stream);
}

// ------------------------------------------------------------------------------------ GroupNorm without allocations
// sdmi_groupnorm for callers that cannot allocate (HIP graph capture, tight serving loops): the per-(sample, channel)
// scale/shift table and the per-block partial sums, which sdmi_groupnorm takes as caller-allocated buffers, live in a
// file-scope device array.  Every GPU the library runs on holds its own copy at its own address, resolved once per
// device by hipGetSymbolAddress.  All calls on one GPU share the array: calls that overlap in time on one GPU race.
// Nothing has to start at zero (every word that is read was written earlier in the same call).
constexpr size_t GN_STATIC_BYTES = 1u << 20;
static __device__ __align__(256) unsigned char scratch[GN_STATIC_BYTES];

static unsigned char* g_gn_static_ptr[64] = {};  // per device: the array's address on that GPU
static int gn_static_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    sdmi_set_error("groupnorm_static: hipGetDevice failed or device index %d >= 64", dev);
    return -1;
  }
  return dev;
}
// Resolve the array's address on the CURRENT device (once per device).  hipGetSymbolAddress is not a stream operation and
// may load the code object: it must not run while a stream of this thread is capturing, so callers that capture call
// this (or one eager sdmi_groupnorm_static) first.
extern "C" int sdmi_groupnorm_static_init(void) {
  const int dev = gn_static_device();
  if (dev < 0) return SDMI_ELAUNCH;
  if (!g_gn_static_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      sdmi_set_error("groupnorm_static: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
      return SDMI_ELAUNCH;
    }
    g_gn_static_ptr[dev] = (unsigned char*)ptr;
  }
  return SDMI_OK;
}

extern "C" int64_t sdmi_groupnorm_static_max_bytes(void) { return (int64_t)GN_STATIC_BYTES; }

extern "C" int sdmi_groupnorm_static(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups,
                                     const float* gamma, const float* beta, float eps, uint32_t flags, void* stream) {
  SDMI_REQUIRE(n > 0 && hw > 0 && c > 0, "groupnorm_static: bad shape n=%d hw=%d c=%d", n, hw, c);
  const int64_t ss_bytes = ((int64_t)n * c * 2 * (int64_t)sizeof(float) + 255) / 256 * 256;
  const int64_t need = ss_bytes + sdmi_groupnorm_ws_bytes(n, hw, c);
  SDMI_REQUIRE(need <= (int64_t)GN_STATIC_BYTES,
               "groupnorm_static: n=%d hw=%d c=%d needs %lld bytes of scratch, the static buffer holds %lld: use sdmi_groupnorm "
               "with caller-allocated buffers", n, hw, c, (long long)need, (long long)GN_STATIC_BYTES);
  // the array of the CURRENT device: `stream` must belong to it (a launch on another device's stream fails in HIP)
  const int dev = gn_static_device();
  if (dev < 0) return SDMI_ELAUNCH;
  if (!g_gn_static_ptr[dev]) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      sdmi_set_error("groupnorm_static: first call on device %d is under stream capture; call sdmi_groupnorm_static_init() "
                     "(or run one call eagerly) before capturing", dev);
      return SDMI_EINVAL;
    }
    const int rc = sdmi_groupnorm_static_init();
    if (rc) return rc;
  }
  unsigned char* ptr = g_gn_static_ptr[dev];
  return sdmi_groupnorm(x, ldx, y, ldy, n, hw, c, groups, gamma, beta, eps, flags, (float*)ptr, ptr + ss_bytes, stream);
}

// ------------------------------------------------------------------------------------ LayerNorm
// one wave per row, the row lives in registers (two-pass mean / variance like torch).
template <int NS>
__global__ __launch_bounds__(256) void layernorm_kernel(const u16* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, u16* __restrict__ y,
                                                        int ldy, int rows, int c) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cc = c >> 3;
  float f[NS][8];
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int ch = lane + 64 * s;
    if (ch < cc) {
      unpack8(*(const uint4*)(x + (long long)row * ldx + ch * 8), f[s]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += f[s][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[s][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)c;
  float sq = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (lane + 64 * s < cc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[s][e] - mean; sq += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int ch = lane + 64 * s;
    if (ch < cc) {
      const float4* gp = (const float4*)(gamma + ch * 8);
      const float4* bp = (const float4*)(beta + ch * 8);
      const float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
      float o[8];
      o[0] = (f[s][0] - mean) * rstd * g0.x + b0.x; o[1] = (f[s][1] - mean) * rstd * g0.y + b0.y;
      o[2] = (f[s][2] - mean) * rstd * g0.z + b0.z; o[3] = (f[s][3] - mean) * rstd * g0.w + b0.w;
      o[4] = (f[s][4] - mean) * rstd * g1.x + b1.x; o[5] = (f[s][5] - mean) * rstd * g1.y + b1.y;
      o[6] = (f[s][6] - mean) * rstd * g1.z + b1.z; o[7] = (f[s][7] - mean) * rstd * g1.w + b1.w;
      *(uint4*)(y + (long long)row * ldy + ch * 8) = pack8(o);
    }
  }
}

extern "C" int sdmi_layernorm(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy,
                              int rows, int c, void* stream) {
  SDMI_REQUIRE(x && gamma && beta && y, "layernorm: null pointer");
  SDMI_REQUIRE(rows > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c, "layernorm: c/ldx/ldy must be multiples of 8");
  SDMI_REQUIRE(c <= 2560, "layernorm: c=%d > 2560", c);
  SDMI_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm: alignment");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
  const int ns = (c / 8 + 63) / 64;
  const u16* xp = (const u16*)x;
  u16* yp = (u16*)y;
  switch (ns) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, xp, ldx, gamma, beta, eps, yp, ldy, rows, c); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, xp, ldx, gamma, beta, eps, yp, ldy, rows, c); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, xp, ldx, gamma, beta, eps, yp, ldy, rows, c); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, xp, ldx, gamma, beta, eps, yp, ldy, rows, c); break;
    default: hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, st, xp, ldx, gamma, beta, eps, yp, ldy, rows, c); break;
  }
  return sdmi_check_launch("layernorm");
}

