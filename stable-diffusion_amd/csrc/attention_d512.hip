// attention_d512.hip - flash-style single-head attention with head_dim 512 (the VAE mid block,
// layers/SpatialTransformer.py:132-161): o = softmax(q k^T * scale) v without an N x N score buffer.
//
// Structure (gfx950): block = 4 waves x 32 query rows, one block per CU (512 VGPRs per wave, 128 KB of LDS).  A 32 x 512
// accumulator does not fit one wave, so the head dimension of O is split four ways: wave w owns O[:, 128w : 128w + 128].
// Every wave computes the full 32 x 32 score block of a key tile itself (32 MFMAs over d = 512 with its 32 x 512 Q fragment
// in 128 registers; the four waves run the same instructions on the same data, so their softmax state is identical bit
// for bit and nothing is exchanged), then 8 MFMAs of P V for its own 128 columns: 40 MFMAs per key tile and wave against
// 16 without the recomputation - the price of no cross-wave reduction of S and no barrier inside a tile.  K / V tiles of
// 32 keys (32 KB each) are staged by LDS-DMA, double buffered.  Operand roles as in attention.hip (S^T = K Q^T, query on
// the lane; V^T through ds_read_b64_tr_b16).
#include "common.h"

struct Attn512Params {
  const u16* q; const u16* k; const u16* v; u16* o;
  long long sq, sk, sv, so;
  int ldq, ldk, ldv, ldo;
  int n, nqt;
  float scale_log2;
};

// tile images [32 keys][512]: 16-byte chunks XOR-swizzled so that a fragment read's lanes spread over the banks (rows are
// 1 KB apart: unswizzled, every key of a chunk column would sit in the same banks)
__device__ __forceinline__ int k5_off(int key, int chunk) { return key * 512 + ((chunk ^ (key & 15)) << 3); }
__device__ __forceinline__ int v5_off(int key, int chunk) { return key * 512 + ((chunk ^ ((key & 3) << 2)) << 3); }

__device__ __forceinline__ void attn5_dma16(const void* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
}

__device__ __forceinline__ hx8 attn5_vfrag(const u16* sV, int row0, int c, int tp) {
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (short4_t __attribute__((address_space(3)))*)(sV + v5_off(row0, c) + 4 * (tp & 1)));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (short4_t __attribute__((address_space(3)))*)(sV + v5_off(row0 + 8, c) + 4 * (tp & 1)));
  union { struct { short4_t a, b; } h; hx8 v; } u;
  u.h.a = lo;
  u.h.b = hi;
  return u.v;
}

template <bool TAIL>
__device__ __forceinline__ void attn5_tile(const u16* sK, const u16* sV, const hx8 (&qf)[32], f32x16 (&oacc)[4], float& mrun,
                                           float& lrun, int key0, int n, int wv, int r, int hh, int dblk, int tq, int tp) {
  // one wave per SIMD: nobody else hides an LDS latency, so the K fragments are requested a group of 8 ahead of the MFMAs
  // that use them, and the 32-step sum runs as two interleaved accumulator chains
  f32x16 s, s1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s[i] = 0.f; s1[i] = 0.f; }
  hx8 kf[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kf[0][j] = *(const hx8*)&sK[k5_off(r, 2 * j + hh)];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[(c + 1) & 1][j] = *(const hx8*)&sK[k5_off(r, 2 * (8 * (c + 1) + j) + hh)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      s = MFMA32(kf[c & 1][j], qf[8 * c + j], s, 0, 0, 0);
      s1 = MFMA32(kf[c & 1][j + 1], qf[8 * c + j + 1], s1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // the first 16 keys' V fragments: in flight during the softmax
  hx8 vf0[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) vf0[db] = attn5_vfrag(sV, 4 * hh + tq, 16 * wv + 4 * db + 2 * dblk + (tp >> 1), tp);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] += s1[i];
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= n) s[i] = -1e30f;
  }
  float mx = s[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mnew = fmaxf(mrun, mx);
  // after the first tiles the maximum rarely moves: rescale O only when some row's did (wave-uniform branch; the rows
  // whose maximum stayed multiply by exp2(0) = 1, exactly what skipping does)
  if (__builtin_amdgcn_ballot_w64(mnew > mrun) != 0) {
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    lrun *= alpha;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    hx8 pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = __builtin_amdgcn_exp2f(s[8 * s2 + j] - mnew);
      lrun += e;
      pb[j] = (h16)e;
    }
    hx8 vf[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vf[db] = s2 == 0 ? vf0[db] : attn5_vfrag(sV, 16 + 4 * hh + tq, 16 * wv + 4 * db + 2 * dblk + (tp >> 1), tp);
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = MFMA32(vf[db], pb, oacc[db], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 1) void attn512_kernel(const Attn512Params p) {
  extern __shared__ __attribute__((aligned(16))) u16 smem5[];  // [2 buffers][K 32 x 512 | V 32 x 512]
  const int t = threadIdx.x, lane = t & 63, r = lane & 31, hh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qt = blockIdx.x % p.nqt, bz = blockIdx.x / p.nqt;
  const u16* Q = p.q + (long long)bz * p.sq;
  const u16* K = p.k + (long long)bz * p.sk;
  const u16* V = p.v + (long long)bz * p.sv;
  u16* O = p.o + (long long)bz * p.so;
  const int qrow = qt * 32 + r;
  const int qrc = qrow < p.n ? qrow : p.n - 1;
  hx8 qf[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    const hx8 raw = *(const hx8*)(Q + (long long)qrc * p.ldq + 16 * ks + 8 * hh);
    if (p.scale_log2 == 1.0f) qf[ks] = raw;  // q already carries scale * log2(e)
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (h16)((float)raw[j] * p.scale_log2);
  }
  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  float mrun = -1e30f, lrun = 0.f;
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) u16*)smem5;
  const int ntiles = (p.n + 31) / 32;
  // staging: wave w moves key rows 8w .. 8w+7 of the tile, one DMA instruction (64 lanes x 16 B = one 1 KB row) each for K
  // and V; the swizzle is applied on the source side (lane l fetches the chunk that belongs at position l)
  unsigned offk[8], offv[8];  // per-lane byte offsets inside a full tile; the tile base advances by 32 rows per tile
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 8 * wv + j;
    offk[j] = 2u * (unsigned)(row * p.ldk + ((lane ^ (row & 15)) << 3));
    offv[j] = 2u * (unsigned)(row * p.ldv + ((lane ^ ((row & 3) << 2)) << 3));
  }
  auto stage = [&](int tile, int buf) {
    const unsigned dst = lds0 + buf * 65536 + 8 * wv * 1024;
    if ((tile + 1) * 32 <= p.n) {
      const char* kt = (const char*)K + (long long)tile * 64 * p.ldk;
      const char* vt = (const char*)V + (long long)tile * 64 * p.ldv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        attn5_dma16(kt + offk[j], dst + j * 1024);
        attn5_dma16(vt + offv[j], dst + 32768 + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 8 * wv + j;
        const int key = min(tile * 32 + row, p.n - 1);  // rows past the end repeat the last key (masked in the scores)
        attn5_dma16(K + (long long)key * p.ldk + ((lane ^ (row & 15)) << 3), dst + j * 1024);
        attn5_dma16(V + (long long)key * p.ldv + ((lane ^ ((row & 3) << 2)) << 3), dst + 32768 + j * 1024);
      }
    }
  };
  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("" ::: "memory");
  __syncthreads();
  const int tfull = p.n >> 5;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
    const u16* cK = smem5 + buf * 32768;
    const u16* cV = cK + 16384;
    if (tile < tfull) attn5_tile<false>(cK, cV, qf, oacc, mrun, lrun, tile * 32, p.n, wv, r, hh, dblk, tq, tp);
    else attn5_tile<true>(cK, cV, qf, oacc, mrun, lrun, tile * 32, p.n, wv, r, hh, dblk, tq, tp);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
  }
  const float l = lrun + __shfl_xor(lrun, 32);
  const float inv = 1.0f / l;
  if (qrow < p.n) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 128 * wv + 32 * db + 8 * g + 4 * hh;
        *(uint2*)(O + (long long)qrow * p.ldo + d) = make_uint2(pack2h(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                                                                pack2h(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv));
      }
  }
}

// ---------------------------------------------------------------------------------------------- 8 waves, 64 query rows
// The 4-wave kernel above is bound by its K / V stream: 32 query rows per block re-read all of K and V (18.9 MB at 9216
// tokens) 288 times = 5.4 GB through the L2 -> LDS path per image.  Here a block is 8 waves = 2 groups of 32 query rows x 4
// slices of the head dimension, so one staged tile serves 64 rows (2.7 GB).  With two waves per SIMD the 32 x 512 Q
// fragment no longer fits (128 registers), so the reduction over d of the scores is split as well: wave (g, w) holds
// Q[rows of g, 128w : 128w + 128] (32 registers) and computes the PARTIAL score block over its slice of d (8 MFMAs instead of
// 32), the four partials of a group are exchanged through the LDS (4 KB each) and every wave adds them in the same order
// (identical sums, so the softmax state of the four waves stays identical bit for bit without further exchange); then
// 8 MFMAs of P V for its own 128 columns of O, as before.  16 MFMAs per tile and wave - no recomputation - for one more
// barrier per tile.  LDS: 2 x 64 KB tile buffers + 32 KB of partials = the CU's 160 KB.
template <bool TAIL>
__device__ __forceinline__ void attn5w8_tile(const u16* sK, const u16* sV, float* xs, const hx8 (&qf)[8], f32x16 (&oacc)[4], float& mrun,
                                             float& lrun, int key0, int n, int g, int w, int lane, int r, int hh, int dblk, int tq, int tp) {
  f32x16 s, s1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s[i] = 0.f; s1[i] = 0.f; }
  hx8 kf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kf[j] = *(const hx8*)&sK[k5_off(r, 16 * w + 2 * j + hh)];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    s = MFMA32(kf[j], qf[j], s, 0, 0, 0);
    s1 = MFMA32(kf[j + 1], qf[j + 1], s1, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  float* const mine = xs + (g * 4 + w) * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *(f32x4*)&mine[(j * 64 + lane) * 4] = f32x4{s[4 * j] + s1[4 * j], s[4 * j + 1] + s1[4 * j + 1], s[4 * j + 2] + s1[4 * j + 2], s[4 * j + 3] + s1[4 * j + 3]};
  __syncthreads();
  {
    const float* const grp = xs + g * 4096;
    f32x4 pr[4][4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[ww][j] = *(const f32x4*)&grp[ww * 1024 + (j * 64 + lane) * 4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[4 * j + e] = ((pr[0][j][e] + pr[1][j][e]) + pr[2][j][e]) + pr[3][j][e];
  }
  __builtin_amdgcn_sched_barrier(0);
  // the first 16 keys' V fragments: in flight during the softmax
  hx8 vf0[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) vf0[db] = attn5_vfrag(sV, 4 * hh + tq, 16 * w + 4 * db + 2 * dblk + (tp >> 1), tp);
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= n) s[i] = -1e30f;
  }
  float mx = s[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mnew = fmaxf(mrun, mx);
  if (__builtin_amdgcn_ballot_w64(mnew > mrun) != 0) {
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    lrun *= alpha;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    hx8 pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = __builtin_amdgcn_exp2f(s[8 * s2 + j] - mnew);
      lrun += e;
      pb[j] = (h16)e;
    }
    hx8 vf[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vf[db] = s2 == 0 ? vf0[db] : attn5_vfrag(sV, 16 + 4 * hh + tq, 16 * w + 4 * db + 2 * dblk + (tp >> 1), tp);
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = MFMA32(vf[db], pb, oacc[db], 0, 0, 0);
  }
}

__global__ __launch_bounds__(512, 1) void attn512w8_kernel(const Attn512Params p) {
  extern __shared__ __attribute__((aligned(16))) u16 smem5[];  // [2 buffers][K 32 x 512 | V 32 x 512] | partial scores [2][4][1024] f32
  const int t = threadIdx.x, lane = t & 63, r = lane & 31, hh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6), g = wv >> 2, w = wv & 3;
  const int qt = blockIdx.x % p.nqt, bz = blockIdx.x / p.nqt;
  const u16* Q = p.q + (long long)bz * p.sq;
  const u16* K = p.k + (long long)bz * p.sk;
  const u16* V = p.v + (long long)bz * p.sv;
  u16* O = p.o + (long long)bz * p.so;
  float* const xs = (float*)(smem5 + 65536);
  const int qrow = qt * 64 + 32 * g + r;
  const int qrc = qrow < p.n ? qrow : p.n - 1;
  hx8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const hx8 raw = *(const hx8*)(Q + (long long)qrc * p.ldq + 128 * w + 16 * ks + 8 * hh);
    if (p.scale_log2 == 1.0f) qf[ks] = raw;  // q already carries scale * log2(e)
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (h16)((float)raw[j] * p.scale_log2);
  }
  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  float mrun = -1e30f, lrun = 0.f;
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) u16*)smem5;
  const int ntiles = (p.n + 31) / 32;
  // staging: wave wv moves key rows 4wv .. 4wv+3 of the tile (one 1 KB row per DMA instruction, K and V)
  unsigned offk[4], offv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 4 * wv + j;
    offk[j] = 2u * (unsigned)(row * p.ldk + ((lane ^ (row & 15)) << 3));
    offv[j] = 2u * (unsigned)(row * p.ldv + ((lane ^ ((row & 3) << 2)) << 3));
  }
  auto stage = [&](int tile, int buf) {
    const unsigned dst = lds0 + buf * 65536 + 4 * wv * 1024;
    if ((tile + 1) * 32 <= p.n) {
      const char* kt = (const char*)K + (long long)tile * 64 * p.ldk;
      const char* vt = (const char*)V + (long long)tile * 64 * p.ldv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        attn5_dma16(kt + offk[j], dst + j * 1024);
        attn5_dma16(vt + offv[j], dst + 32768 + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * wv + j;
        const int key = min(tile * 32 + row, p.n - 1);  // rows past the end repeat the last key (masked in the scores)
        attn5_dma16(K + (long long)key * p.ldk + ((lane ^ (row & 15)) << 3), dst + j * 1024);
        attn5_dma16(V + (long long)key * p.ldv + ((lane ^ ((row & 3) << 2)) << 3), dst + 32768 + j * 1024);
      }
    }
  };
  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("" ::: "memory");
  __syncthreads();
  const int tfull = p.n >> 5;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
    const u16* cK = smem5 + buf * 32768;
    const u16* cV = cK + 16384;
    if (tile < tfull) attn5w8_tile<false>(cK, cV, xs, qf, oacc, mrun, lrun, tile * 32, p.n, g, w, lane, r, hh, dblk, tq, tp);
    else attn5w8_tile<true>(cK, cV, xs, qf, oacc, mrun, lrun, tile * 32, p.n, g, w, lane, r, hh, dblk, tq, tp);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
  }
  const float l = lrun + __shfl_xor(lrun, 32);
  const float inv = 1.0f / l;
  if (qrow < p.n) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = 128 * w + 32 * db + 8 * q4 + 4 * hh;
        *(uint2*)(O + (long long)qrow * p.ldo + d) = make_uint2(pack2h(oacc[db][4 * q4] * inv, oacc[db][4 * q4 + 1] * inv),
                                                                pack2h(oacc[db][4 * q4 + 2] * inv, oacc[db][4 * q4 + 3] * inv));
      }
  }
}

static int attention_d512_launch(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v, int ldv,
                                 int64_t sv, void* o, int ldo, int64_t so, int batch, int n, float scale, bool log2q, void* stream) {
  SDMI_REQUIRE(q && k && v && o, "attention_d512: null pointer");
  SDMI_REQUIRE(batch > 0 && n > 0, "attention_d512: bad shape");
  SDMI_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && ldq >= 512 && ldk >= 512 && ldv >= 512 && ldo >= 512,
               "attention_d512: row strides must be multiples of 8 and >= 512");
  SDMI_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0, "attention_d512: pointer alignment");
  SDMI_REQUIRE(sq % 8 == 0 && sk % 8 == 0 && sv % 8 == 0 && so % 4 == 0, "attention_d512: batch strides must keep alignment");
  SDMI_REQUIRE(ldk < (1 << 22) && ldv < (1 << 22), "attention_d512: k/v row stride too large");  // 32-bit in-tile byte offsets
  SDMI_REQUIRE(log2q || scale > 0.f, "attention_d512: scale must be positive");
  SDMI_REQUIRE((long long)((n + 31) / 32) * batch < (1ll << 31), "attention_d512: grid too large");
  Attn512Params p;
  p.q = (const u16*)q; p.k = (const u16*)k; p.v = (const u16*)v; p.o = (u16*)o;
  p.sq = sq; p.sk = sk; p.sv = sv; p.so = so; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.n = n; p.scale_log2 = log2q ? 1.0f : scale * 1.44269504088896340736f;
  constexpr int lds_bytes = 2 * 65536, lds_bytes_w8 = 2 * 65536 + 2 * 4 * 4096;
  static unsigned long long attr_done = 0;
  static bool w8_ok[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute((const void*)attn512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    w8_ok[dev & 63] = hipFuncSetAttribute((const void*)attn512w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_w8) == hipSuccess;
    (void)hipGetLastError();
    attr_done |= 1ull << (dev & 63);
  }
  // attn_d512_variant: 0 = automatic (the 8-wave kernel), 1 = the 4-wave kernel, 2 = the 8-wave kernel (A/B switch)
  const int variant = sdmi_opt(OPT_attn_d512_variant);
  SDMI_REQUIRE(variant != 2 || w8_ok[dev & 63], "attention_d512: the 8-wave kernel needs %d bytes of LDS per block", lds_bytes_w8);
  if (variant != 1 && w8_ok[dev & 63]) {
    p.nqt = (n + 63) / 64;
    hipLaunchKernelGGL(attn512w8_kernel, dim3(p.nqt * batch), dim3(512), lds_bytes_w8, (hipStream_t)stream, p);
    sdmi_note_kernel("attn512w8_kernel");
  } else {
    p.nqt = (n + 31) / 32;
    hipLaunchKernelGGL(attn512_kernel, dim3(p.nqt * batch), dim3(256), lds_bytes, (hipStream_t)stream, p);
    sdmi_note_kernel("attn512_kernel");
  }
  return sdmi_check_launch("attention_d512");
}

extern "C" int sdmi_attention_d512(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v, int ldv,
                                   int64_t sv, void* o, int ldo, int64_t so, int batch, int n, float scale, void* stream) {
  return attention_d512_launch(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, batch, n, scale, false, stream);
}

extern "C" int sdmi_attention_d512_log2q(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                                         int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int n, void* stream) {
  return attention_d512_launch(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, batch, n, 1.0f, true, stream);
}
