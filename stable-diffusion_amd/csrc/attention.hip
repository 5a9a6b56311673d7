// attention.hip - fused flash-style attention forward, head_dim 64, 16-bit (fp16 / bf16 build) in/out, fp32 softmax state.
//
// Replaces xformers.ops.memory_efficient_attention(q,k,v) and the head split/merge copies around it
// (layers/SpatialTransformer.py:76-95): o = softmax(q k^T * scale) v, no mask, no dropout.  q/k/v/o are
// read and written in place as (B, tokens, heads*64) views, so nothing is permuted or made contiguous.
//
// Structure (gfx950): block = 4 waves, each wave owns 32 query rows; KV tiles of 64 keys are staged by LDS-DMA
// (global_load_lds_dwordx4) into double-buffered tile images, tile t+1 in flight while tile t is computed.
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
  // all eight K fragments are requested before the first MFMA needs one (the compiler otherwise reads each just in time
  // and every MFMA of the chain waits out an LDS latency); the two accumulator chains alternate
  hx8 kf[2][4];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[kt2][ks] = *(const hx8*)&sK[k_off(32 * kt2 + r, 2 * ks + hh)];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) s[kt2] = MFMA32(kf[kt2][ks], qf[ks], s[kt2], 0, 0, 0);
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
  // double-buffered tile images, one barrier per KV tile.  Four separate arrays, and the loop below is unrolled by two
  // so that every access names its array: the compiler then knows that a fragment read of one buffer cannot alias the
  // LDS-DMA in flight into the other (it otherwise waits for vmcnt(0) before the first read of every tile).
  __shared__ __attribute__((aligned(16))) u16 sK0[64 * 64], sK1[64 * 64], sV0[64 * 64], sV1[64 * 64];
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

  // staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land on 1 KB of consecutive LDS = 8 tile rows): wave w
  // moves rows [16w, 16w+16) of K and of V with two instructions each; the XOR swizzle of the LDS image is applied on
  // the SOURCE side (lane -> (row, LDS slot) -> the logical chunk that belongs in that slot).  Nothing passes through
  // registers and no ds_write is issued.  Full tiles advance four per-thread pointers by a constant; the last,
  // partial tile clamps its key rows to nkv-1 (those columns of S are masked to -inf, so the duplicated rows weigh
  // nothing): no per-load predicates in the steady state.
  const int srow0 = 16 * wv + (lane >> 3), srow1 = srow0 + 8, spos = lane & 7;
  const int ck0 = spos ^ ((srow0 >> 1) & 7), ck1 = spos ^ ((srow1 >> 1) & 7);
  const int cv0 = spos ^ (((srow0 >> 1) & 1) << 2), cv1 = spos ^ (((srow1 >> 1) & 1) << 2);
  // per-lane element offsets inside a tile (32-bit) + a wave-uniform tile base that advances by 64 rows: the loads
  // address as SGPR base + VGPR offset, no vector arithmetic per tile
  const unsigned ok0 = 2u * (srow0 * p.ldk + ck0 * 8), ok1 = 2u * (srow1 * p.ldk + ck1 * 8);  // bytes, < 2^31 (host)
  const unsigned ov0 = 2u * (srow0 * p.ldv + cv0 * 8), ov1 = 2u * (srow1 * p.ldv + cv1 * 8);
  const char *kt = (const char*)K, *vt = (const char*)V;
  const long long stepk = (long long)128 * p.ldk, stepv = (long long)128 * p.ldv;  // bytes per 64-row tile
  const int wbase = __builtin_amdgcn_readfirstlane(wv) * 1024;  // elements: this wave's first 1 KB block of a tile image
#define ATTN_GLDS(SRC, DST) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
#define ATTN_STAGE_TILE(KV0, SK, SV)                                                          \
  if ((KV0) + 64 <= p.nkv) {                                                                  \
    ATTN_GLDS(kt + ok0, &SK[wbase]); ATTN_GLDS(vt + ov0, &SV[wbase]);                         \
    ATTN_GLDS(kt + ok1, &SK[wbase + 512]); ATTN_GLDS(vt + ov1, &SV[wbase + 512]);             \
    kt += stepk; vt += stepv;                                                                 \
  } else {                                                                                    \
    const int key0 = min((KV0) + srow0, p.nkv - 1), key1 = min((KV0) + srow1, p.nkv - 1);     \
    ATTN_GLDS(K + (long long)key0 * p.ldk + ck0 * 8, &SK[wbase]);                             \
    ATTN_GLDS(V + (long long)key0 * p.ldv + cv0 * 8, &SV[wbase]);                             \
    ATTN_GLDS(K + (long long)key1 * p.ldk + ck1 * 8, &SK[wbase + 512]);                       \
    ATTN_GLDS(V + (long long)key1 * p.ldv + cv1 * 8, &SV[wbase + 512]);                       \
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  float mrun = -1e30f, lrun = 0.f;

  // transposed-read lane constants: 16-lane group g -> (hh, dblk); inside the group lane i = 4*tq + tp
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;

  const int ntiles = (p.nkv + 63) / 64;
  ATTN_STAGE_TILE(0, sK0, sV0);
  __syncthreads();  // waits for this wave's DMA (vmcnt) before the barrier: tile 0 has landed for everyone
  const int nfull = p.nkv >> 6;  // full tiles run the mask-free instantiation
  // one full tile read from (CK, CV); the other buffer was last read a tile ago and every wave is past that barrier:
  // tile+1 lands in it while this tile is computed
#define ATTN_FULL_TILE(TILE, CK, CV, NK, NV)                                                                       \
  {                                                                                                                \
    const int kv0 = (TILE) * 64;                                                                                   \
    if ((TILE) + 1 < ntiles) { ATTN_STAGE_TILE(kv0 + 64, NK, NV); }                                                \
    attn_tile<false>(CK, CV, qf, oacc, mrun, lrun, p.scale_log2, kv0, p.nkv, r, hh, dblk, tq, tp);                 \
    __syncthreads();                                                                                               \
  }
  int tile = 0;
  for (; tile + 2 <= nfull; tile += 2) {
    ATTN_FULL_TILE(tile, sK0, sV0, sK1, sV1);
    ATTN_FULL_TILE(tile + 1, sK1, sV1, sK0, sV0);
  }
  if (tile < nfull) {  // odd count of full tiles: one more from buffer 0, a partial tile (if any) follows in buffer 1
    ATTN_FULL_TILE(tile, sK0, sV0, sK1, sV1);
    if (nfull < ntiles) attn_tile<true>(sK1, sV1, qf, oacc, mrun, lrun, p.scale_log2, nfull * 64, p.nkv, r, hh, dblk, tq, tp);
  } else if (nfull < ntiles) {  // the partial last tile (staged by the iteration before it, or by the prologue)
    attn_tile<true>(sK0, sV0, qf, oacc, mrun, lrun, p.scale_log2, nfull * 64, p.nkv, r, hh, dblk, tq, tp);
  }
#undef ATTN_FULL_TILE
#undef ATTN_STAGE_TILE
#undef ATTN_GLDS

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
  SDMI_REQUIRE(ldk < (1 << 22) && ldv < (1 << 22), "attention: k/v row stride too large");  // 32-bit in-tile byte offsets
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
