// attention.hip - fused flash-style attention forward, head_dim 64, bf16 in/out, fp32 softmax state.
//
// Replaces xformers.ops.memory_efficient_attention(q,k,v) and the head split/merge copies around it
// (layers/SpatialTransformer.py:76-95): o = softmax(q k^T * scale) v, no mask, no dropout.  q/k/v/o are
// read and written in place as (B, tokens, heads*64) views, so nothing is permuted or made contiguous.
//
// Structure (gfx950): block = 4 waves, each wave owns 32 query rows; KV tiles of 64 keys are staged
// global -> registers -> LDS (the loads of tile t+1 are in flight while tile t is computed).
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16 with K rows as the A operand (ds_read_b128, XOR-swizzled rows) and
//                 the wave's Q fragment (registers, loaded once) as B: the accumulator then has the QUERY on
//                 the lane and the keys in registers, so the online softmax (row max / sum) is lane-local
//                 plus one cross-half exchange, and P (bf16) is directly the B operand of the next product.
//   O^T += V^T P^T  V is stored row-major [key][d] and read through ds_read_b64_tr_b16 (hardware
//                 transpose), the k order inside each 16-key step following the accumulator's row order.
#include "common.h"

struct AttnParams {
  const u16* q; const u16* k; const u16* v; u16* o;
  long long sq, sk, sv, so;
  int ldq, ldk, ldv, ldo;
  int nq, nkv, nqt, heads;
  float scale_log2;
};

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }

// One KV tile for one wave: S^T = K Q^T, online softmax with the query on the lane, O^T += V^T P^T.
template <bool TAIL>
__device__ __forceinline__ void attn_tile(const u16* sK, const u16* sV, const hx8 (&qf)[4], f32x16 (&oacc)[2], float& mrun,
                                          float& lrun, float scale_log2, int kv0, int nkv, int r, int hh, int dblk, int tq,
                                          int tp) {
  f32x16 s[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { s[0][i] = 0.f; s[1][i] = 0.f; }
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const hx8 a = *(const hx8*)&sK[k_off(32 * kt2 + r, 2 * ks + hh)];
      s[kt2] = MFMA32(a, qf[ks], s[kt2], 0, 0, 0);
    }
  // raw-score max (scale > 0, so max commutes with the scaling); padded keys of the last tile are excluded.  TAIL is a
  // template parameter and the partial tile is peeled out of the loop: as a run-time (even wave-uniform) condition hipcc
  // turned the mask into ~100 always-executed compare/select instructions per tile, a third of the loop's VALU work.
  if constexpr (TAIL) {
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kv0 + 32 * kt2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= nkv) s[kt2][i] = -1e30f;
      }
  }
  float mx = -1e30f;
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt2][i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mnew = fmaxf(mrun, mx * scale_log2);
  // the running maximum settles after a few tiles: when no lane's maximum moved (wave-uniform test) alpha is exactly 1 for
  // every lane and the rescale of O and l (17 VALU instructions per tile) is skipped
  const bool moved = __builtin_amdgcn_ballot_w64(mnew > mrun) != 0;
  float rs = 0.f;
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt2][i], scale_log2, -mnew));
      s[kt2][i] = e;
      rs += e;
    }
  if (moved) {
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    lrun *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
  }
  lrun += rs;
  mrun = mnew;
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      hx8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (h16)s[kt2][8 * s2 + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int c = 4 * dt + 2 * dblk + (tp >> 1);
        const int row0 = 32 * kt2 + 16 * s2 + 4 * hh + tq;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4_t __attribute__((address_space(3)))*)(sV + v_off(row0, c) + 4 * (tp & 1)));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (short4_t __attribute__((address_space(3)))*)(sV + v_off(row0 + 8, c) + 4 * (tp & 1)));
        union { struct { short4_t a, b; } h; hx8 v; } u;
        u.h.a = lo;
        u.h.b = hi;
        oacc[dt] = MFMA32(u.v, pb, oacc[dt], 0, 0, 0);
      }
    }
}

// grid: 1-D, nqt * heads * batch blocks; the XCD remap gives each XCD a contiguous run of (batch, head, q-tile) so
// the K/V of the (one or two) heads an XCD is working on stay in its 4 MB L2 instead of being re-streamed by 8 XCDs.
__global__ __launch_bounds__(256, 3) void attn64_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) u16 sK[2][64 * 64];  // double-buffered: one barrier per KV tile
  __shared__ __attribute__((aligned(16))) u16 sV[2][64 * 64];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, hh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % p.nqt, bh = bid / p.nqt;
  const int head = bh % p.heads, bz = bh / p.heads;
  const u16* Q = p.q + (long long)bz * p.sq + head * 64;
  const u16* K = p.k + (long long)bz * p.sk + head * 64;
  const u16* V = p.v + (long long)bz * p.sv + head * 64;
  u16* O = p.o + (long long)bz * p.so + head * 64;

  const int qrow = qt * 128 + wv * 32 + r;
  const int qrc = qrow < p.nq ? qrow : p.nq - 1;
  hx8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const hx8*)(Q + (long long)qrc * p.ldq + 16 * ks + 8 * hh);

  // staging: 64 rows x 8 chunks of 16 B per tile and operand; thread -> (row = t>>3 (+32), chunk = t&7).  Full tiles
  // advance four per-thread pointers by a constant; the last, partial tile clamps its key rows to nkv-1 (those columns
  // of S are masked to -inf, so the duplicated rows weigh nothing): no per-load predicates in the steady state.
  const int chunk = t & 7, r0 = t >> 3;
  u32x4 rk0, rk1, rv0, rv1;
  const u16 *pk0 = K + (long long)r0 * p.ldk + chunk * 8, *pk1 = pk0 + (long long)32 * p.ldk;
  const u16 *pv0 = V + (long long)r0 * p.ldv + chunk * 8, *pv1 = pv0 + (long long)32 * p.ldv;
  const long long stepk = (long long)64 * p.ldk, stepv = (long long)64 * p.ldv;
#define ATTN_LOAD_TILE(KV0)                                                                   \
  if ((KV0) + 64 <= p.nkv) {                                                                  \
    rk0 = *(const u32x4*)pk0; rv0 = *(const u32x4*)pv0;                                       \
    rk1 = *(const u32x4*)pk1; rv1 = *(const u32x4*)pv1;                                       \
    pk0 += stepk; pk1 += stepk; pv0 += stepv; pv1 += stepv;                                   \
  } else {                                                                                    \
    const int key0 = min((KV0) + r0, p.nkv - 1), key1 = min((KV0) + r0 + 32, p.nkv - 1);      \
    rk0 = *(const u32x4*)(K + (long long)key0 * p.ldk + chunk * 8);                           \
    rv0 = *(const u32x4*)(V + (long long)key0 * p.ldv + chunk * 8);                           \
    rk1 = *(const u32x4*)(K + (long long)key1 * p.ldk + chunk * 8);                           \
    rv1 = *(const u32x4*)(V + (long long)key1 * p.ldv + chunk * 8);                           \
  }
#define ATTN_STORE_TILE(B)                                    \
  {                                                           \
    *(u32x4*)&sK[B][k_off(r0, chunk)] = rk0;                  \
    *(u32x4*)&sV[B][v_off(r0, chunk)] = rv0;                  \
    *(u32x4*)&sK[B][k_off(r0 + 32, chunk)] = rk1;             \
    *(u32x4*)&sV[B][v_off(r0 + 32, chunk)] = rv1;             \
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  float mrun = -1e30f, lrun = 0.f;

  // transposed-read lane constants: 16-lane group g -> (hh, dblk); inside the group lane i = 4*tq + tp
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;

  const int ntiles = (p.nkv + 63) / 64;
  ATTN_LOAD_TILE(0);
  ATTN_STORE_TILE(0);
  __syncthreads();
  const int nfull = p.nkv >> 6;  // full tiles run the mask-free instantiation
  for (int tile = 0; tile < nfull; ++tile) {
    const int kv0 = tile * 64, cur = tile & 1;
    const bool more = tile + 1 < ntiles;
    if (more) { ATTN_LOAD_TILE(kv0 + 64); }  // in flight over this tile's MFMAs
    attn_tile<false>(sK[cur], sV[cur], qf, oacc, mrun, lrun, p.scale_log2, kv0, p.nkv, r, hh, dblk, tq, tp);
    if (more) ATTN_STORE_TILE(cur ^ 1);  // the other buffer: last read in tile-1, every wave is past that barrier
    __syncthreads();
  }
  if (nfull < ntiles)  // the partial last tile (already staged by the iteration before it, or by the prologue)
    attn_tile<true>(sK[nfull & 1], sV[nfull & 1], qf, oacc, mrun, lrun, p.scale_log2, nfull * 64, p.nkv, r, hh, dblk, tq, tp);
#undef ATTN_LOAD_TILE
#undef ATTN_STORE_TILE

  const float ltot = lrun + __shfl_xor(lrun, 32);
  const float inv = 1.0f / ltot;
  if (qrow < p.nq) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * hh;
        const uint2 o2 = make_uint2(pack2h(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv),
                                    pack2h(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv));
        *(uint2*)(O + (long long)qrow * p.ldo + d) = o2;
      }
  }
}

extern "C" int sdmi_attention(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                              int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq, int nkv,
                              float scale, void* stream) {
  SDMI_REQUIRE(q && k && v && o, "attention: null pointer");
  SDMI_REQUIRE(batch > 0 && heads > 0 && nq > 0 && nkv > 0, "attention: bad shape");
  SDMI_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention: row strides must be multiples of 8");
  SDMI_REQUIRE(ldq >= heads * 64 && ldk >= heads * 64 && ldv >= heads * 64 && ldo >= heads * 64, "attention: row stride < heads*64");
  SDMI_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0, "attention: pointer alignment");
  SDMI_REQUIRE(sq % 8 == 0 && sk % 8 == 0 && sv % 8 == 0 && so % 4 == 0, "attention: batch strides must keep alignment");
  SDMI_REQUIRE(scale > 0.f, "attention: scale must be positive");
  SDMI_REQUIRE((long long)((nq + 127) / 128) * heads * batch < (1ll << 31), "attention: grid too large");
  AttnParams p;
  p.q = (const u16*)q; p.k = (const u16*)k; p.v = (const u16*)v; p.o = (u16*)o;
  p.sq = sq; p.sk = sk; p.sv = sv; p.so = so; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.nq = nq; p.nkv = nkv; p.scale_log2 = scale * 1.44269504088896340736f;
  p.nqt = (nq + 127) / 128; p.heads = heads;
  dim3 grid(p.nqt * heads * batch), block(256);
  hipLaunchKernelGGL(attn64_kernel, grid, block, 0, (hipStream_t)stream, p);
  return sdmi_check_launch("attention");
}
