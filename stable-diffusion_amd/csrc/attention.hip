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
  // fused LayerNorm + to_q projection (attn64_kernel<true>, sdmi_cross_attention_lnq): q is not read; the query rows are
  // computed from the raw token stream x, the gamma-folded to_q weights and the row statistics x's producer left
  const u16* x; long long sx; int ldx;      // (B, nq, C) raw tokens
  const u16* wq; int ldwq;                  // [heads * 64][C], K contiguous (sdmi_pack_linear_ln; scale * log2(e) folded in)
  const float* lnu; const float* lnv;       // [heads * 64]
  const float* rstats; int parts; float eps;  // [B * nq][parts][2] (sum, sumsq) of x's rows
  int C;
};

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }
// eight packed words of P (32 keys of one chain) and their row-sum helpers (see A3_E2)
union A3P { hx8 v; hx2 h[4]; };
__device__ __forceinline__ hx2 a3_ones() { return hx2{(h16)1.0f, (h16)1.0f}; }
#define A3_SUM4(P) (((P).h[0] + (P).h[1]) + ((P).h[2] + (P).h[3]))
#define A3_LADD(L, TA, TB) L = DOT2H((TA) + (TB), a3_ones(), L)

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
// QPROJ (round 3, sdmi_cross_attention_lnq): the cross-attention's query projection inside the kernel.  The reference runs
// LayerNorm -> to_q (layers/SpatialTransformer.py:184, :62) in front of every cross-attention: here a GEMM launch of ~10 us for
// a 128 x 64 x C product per block.  Each wave computes the Q^T tile of its own 32 rows with the same 32x32x16 MFMA it uses for
// S^T = K Q^T: A = a 32-row slab of the head's weights, B = the wave's token rows straight from global memory (a lane's B
// fragment is 8 consecutive channels of its row: one 16-byte load).  The weight rows of the two slabs are taken in the order
// that makes the MFMA result registers of lane (row r, half hh) exactly the B fragments qf[ks] of the score MFMAs
// (d = 16 ks + 8 hh + e): no lane exchange, no LDS.  The LayerNorm is folded as in the GEMM epilogue (csrc/igemm.hip):
// q = rstd (acc - mean u[d]) + v[d] with the row's (mean, rstd) from its producer's row statistics, rounded to 16 bits as the
// stored q would have been.  K / V tile 0 is requested before the projection and lands under it.
template <bool QPROJ>
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
  if constexpr (!QPROJ) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const hx8*)(Q + (long long)qrc * p.ldq + 16 * ks + 8 * hh);
  }

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
  extern __shared__ __attribute__((aligned(16))) u16 dynW[];  // QPROJ: the head's 64 weight rows, C / 8 + 1 slots of 16 B per row
  [[maybe_unused]] const int wslots = p.C / 8 + 1;             // (the odd slot count keeps a 16-lane fragment read on 16 distinct slots)
  if constexpr (QPROJ) {
    {
      const int total = 64 * wslots;
      const u16* wsrc = p.wq + (long long)head * 64 * p.ldwq;
      for (int base = 0; base < total; base += 256) {
        const int pc = min(base + t, total - 1);
        const int row = pc / wslots, piece = min(pc - row * wslots, wslots - 2);   // the pad slot re-fetches the last piece
        ATTN_GLDS(wsrc + (long long)row * p.ldwq + piece * 8, &dynW[(base + __builtin_amdgcn_readfirstlane(wv) * 64) * 8]);
      }
    }
    // ---- Q^T = Wq_h X^T: weight row of A-row m (slab 0; slab 1 = + 32) so that result register i of lane (r, hh) is
    //      q[r][16 (i >> 3) + 8 hh + (i & 7)] (+ 32 for slab 1)
    const int dm = 16 * (r >> 4) + 8 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1) + (r & 3);
    const u16* xr = p.x + (long long)bz * p.sx + (long long)qrc * p.ldx + 8 * hh;
    // the row's channels in chunks of 320 (20 fragments in flight at once: one memory latency per chunk)
    hx8 xf[20];
#pragma unroll
    for (int s4 = 0; s4 < 20; ++s4) xf[s4] = *(const hx8*)(xr + 16 * s4);
    f32x16 qa[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { qa[0][i] = 0.f; qa[1][i] = 0.f; }
    // ---- row statistics -> (mean, rstd): four interleaved partial sums, finished in double (as the GEMM's LN fold)
    const float2* sp = (const float2*)p.rstats + ((long long)bz * p.nq + qrc) * p.parts;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int pp = 0; pp < p.parts; pp += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2 v = pp + j < p.parts ? sp[pp + j] : make_float2(0.f, 0.f);
        a0[j] += v.x; a1[j] += v.y;
      }
    }
    const double inv = 1.0 / (double)p.C;
    const double mean = (double)((a0[0] + a0[1]) + (a0[2] + a0[3])) * inv;
    const double var = fmax((double)((a1[0] + a1[1]) + (a1[2] + a1[3])) * inv - mean * mean, 0.0);
    const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)p.eps));
    __syncthreads();  // the head's weights (DMA above) have landed for every wave
    const u16* wl0 = dynW + (dm * wslots + hh) * 8;
    const u16* wl1 = wl0 + 32 * wslots * 8;
    for (int cb = 0; cb < p.C; cb += 320) {
      if (cb > 0) {
#pragma unroll
        for (int s4 = 0; s4 < 20; ++s4) xf[s4] = *(const hx8*)(xr + cb + 16 * s4);
      }
#pragma unroll
      for (int s4 = 0; s4 < 20; ++s4) {
        const hx8 wa = *(const hx8*)(wl0 + (cb + 16 * s4) ), wb = *(const hx8*)(wl1 + (cb + 16 * s4));
        qa[0] = MFMA32(wa, xf[s4], qa[0], 0, 0, 0);
        qa[1] = MFMA32(wb, xf[s4], qa[1], 0, 0, 0);
      }
    }
    // ---- LayerNorm fold + rounding: qf[ks][e] = q[r][16 ks + 8 hh + e] = result register 8 (ks & 1) + e of slab ks >> 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float* up = p.lnu + head * 64 + 16 * ks + 8 * hh;
      const float* vp = p.lnv + head * 64 + 16 * ks + 8 * hh;
      const f32x4 u0 = *(const f32x4*)up, u1 = *(const f32x4*)(up + 4), v0 = *(const f32x4*)vp, v1 = *(const f32x4*)(vp + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float uu = e < 4 ? u0[e & 3] : u1[e & 3], vv = e < 4 ? v0[e & 3] : v1[e & 3];
        qf[ks][e] = f2h16((qa[ks >> 1][8 * (ks & 1) + e] - fmean * uu) * frstd + vv);
      }
    }
  }
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

// ------------------------------------------------------------------------------------------ 64 query rows per wave
// Second structure, for long self-attention (the 96x96 / 128x128 levels: 9216 / 16384 tokens).  The SQ counters of the
// kernel above (profiles/r02_attn_pmc.txt) show it VALU-issue bound: per 32 x 64 wave-tile 16 MFMAs (512 matrix-pipe cycles)
// against ~150 vector instructions at 4 - 8 issue cycles each, a third of the time neither pipe busy (dependency chain
// QK^T -> row max -> cross-half exchange through the LDS -> exp -> PV, once per tile and wave).  Here
//   * a wave owns 64 rows = two independent 32-row chains (half the K / V fragment reads per row);
//   * a block is 128 rows x 2 halves of the key range: waves 0/1 sweep the first half of the keys for rows 0-63 / 64-127,
//     waves 2/3 the second half, each pair with its own double-buffered K/V stream; the halves' (m, l, O) are merged
//     through the LDS at the end.  Work then comes in units of 64 rows x half the keys: 9216 tokens x 10 (batch, head)
//     pairs give 2880 units for 1024 SIMDs (94 % after rounding up to whole units, against 70 % for 64 rows x all keys);
//   * Q is pre-scaled by scale * log2(e) and the running maximum enters the scores as the initial accumulator, so exp2()
//     applies to the MFMA result as it stands;
//   * the maximum is only raised when a row's scores exceed it by more than 2^8 (checked per 32-key block on each lane's
//     own half of the keys, wave-uniform branch; the cross-half exchange happens only inside that rare branch): p <= 256
//     keeps fp16's relative precision, and the exps no longer wait for a max reduction.
constexpr float ATTN_DEFER_LOG2 = 8.0f;  // raise the running maximum only when a block's scores exceed it by more than this

// One 32-key block (half kb of the staged 64-key tile) for one wave: two 32-row chains.
template <bool FIRST, bool TAIL>
__device__ __forceinline__ void attn2_block(const u16* sK, const u16* sV, int kb, const hx8 (&qf)[2][4], f32x16 (&oacc)[2][2],
                                            f32x16 (&negm)[2], float (&mrun)[2], float (&lrun)[2], int key0, int nkv, int r, int hh,
                                            int dblk, int tq, int tp) {
  hx8 kf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const hx8*)&sK[k_off(32 * kb + r, 2 * ks + hh)];
  f32x16 s[2];
  // = q.k * scale * log2(e) - m: the first MFMA of a chain reads -m as C and writes another register tile (the builtin
  // ties C to D and copies the 16 registers first: 32 moves per block for the two chains)
  asm(MFMA32_MNEMONIC " %0, %2, %3, %5\n\t" MFMA32_MNEMONIC " %1, %2, %4, %6"
      : "=&v"(s[0]), "=&v"(s[1])
      : "v"(kf[0]), "v"(qf[0][0]), "v"(qf[1][0]), "v"(negm[0]), "v"(negm[1]));
#pragma unroll
  for (int ks = 1; ks < 4; ++ks)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) s[qb] = MFMA32(kf[ks], qf[qb][ks], s[qb], 0, 0, 0);
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (key >= nkv) { s[0][i] = -1e30f; s[1][i] = -1e30f; }
    }
  }
  float mx[2];  // maximum over this lane's 16 keys of the block (the other 16 sit on lane ^ 32)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float m0 = s[qb][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m0 = fmaxf(m0, s[qb][i]);
    mx[qb] = m0;
  }
  const bool raise = FIRST || __builtin_amdgcn_ballot_w64(fmaxf(mx[0], mx[1]) > ATTN_DEFER_LOG2) != 0;
  if (raise) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float mrow = fmaxf(mx[qb], __shfl_xor(mx[qb], 32));
      // FIRST: the maximum starts at 0 (a zero C tile); move it to the block's maximum, whatever its sign; O and l are zero
      const float delta = FIRST ? mrow : fmaxf(mrow, 0.f);
      mrun[qb] += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[qb][i] -= delta;
      if constexpr (!FIRST) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        lrun[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[qb][0][i] *= alpha; oacc[qb][1][i] *= alpha; }
      }
      // -(m + delta), element by element from the old tile (a fresh splat is a new value to the register allocator, which
      // then keeps two copies of the tile and moves 32 registers per block on the common path)
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[qb][i] -= delta;
    }
  }
  // exp -> fp16 P -> PV in two 16-key halves: the second half's exps issue behind the first half's MFMAs
  hx2 lpart[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    hx8 pb[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[qb][j] = (h16)__builtin_amdgcn_exp2f(s[qb][8 * s2 + j]);
      // row sums of the rounded P (the operations of attn3_block, in its order: same bits)
      A3P u;
      u.v = pb[qb];
      if (s2 == 0) lpart[qb] = A3_SUM4(u);
      else A3_LADD(lrun[qb], lpart[qb], A3_SUM4(u));
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int c = 4 * dt + 2 * dblk + (tp >> 1);
      const int row0 = 32 * kb + 16 * s2 + 4 * hh + tq;
      const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (short4_t __attribute__((address_space(3)))*)(sV + v_off(row0, c) + 4 * (tp & 1)));
      const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (short4_t __attribute__((address_space(3)))*)(sV + v_off(row0 + 8, c) + 4 * (tp & 1)));
      union { struct { short4_t a, b; } h; hx8 v; } u;
      u.h.a = lo;
      u.h.b = hi;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) oacc[qb][dt] = MFMA32(u.v, pb[qb], oacc[qb][dt], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------ software-pipelined block (round 3)
// attn2_block issues the two chains in lockstep: 8 QK^T MFMAs with nothing beside them, then a vector-dense softmax with 8
// PV MFMAs squeezed in (SQ counters, profiles/r02_attn_pmc_attn64x2_kernel.txt: matrix and vector pipes busy together for
// 26 % of the matrix time).  A v_mfma_f32_32x32x16 occupies the SIMD's issue for 8 of its 32 cycles
// (MI355X_MICROARCH.md, cycle constants): ~24 cycles of independent vector work fit beside every MFMA.  Here chain 1 LAGS
// chain 0 by half a block - its exp / P.V for block i - 1 run beside chain 0's QK^T for block i - so that every MFMA of
// the 16 per block has about one sixteenth of the block's softmax work (two exp2, a pack, a dot2) in its shadow:
//     MFMA   Q0.0 Q0.1 Q0.2 Q0.3 | V1a.0 V1a.1 | Q1.0 Q1.1 Q1.2 Q1.3 | V1b.0 V1b.1 | V0a.0 V0a.1 | V0b.0 V0b.1
//     VALU   E1a0 E1a1 E1a2 E1a3 | E1b01 E1b23 | max0 E0a0 E0a1 E0a2 | E0a3  E0b0  | E0b1  E0b23 | max1   -
// (Qc = S^T chain c, Vcx = O^T += V^T P^T of chain c for keys 0-15 (a) / 16-31 (b) of the block, E = two scores -> one
// packed word of P + its row-sum term, chain 1's E / V rows belong to the PREVIOUS block).  Chain 1's scores (s1) and the
// V fragments (va, vb: registers, time-shared by chain 1's old block and chain 0's new one) live across calls; nothing of
// an old tile is read from the LDS after its barrier.  Every chain sees the operations of attn2_block in the same order:
// the results are bit-identical (tested).  The order is pinned with sched_barrier fences: left alone the scheduler clusters
// the MFMAs again.
#define A3_FENCE() __builtin_amdgcn_sched_barrier(0)
// two scores -> one packed word of P.  Row sums (round 3, third form): the eight packed words of a 32-key block are added
// as fp16 pairs in a tree (v_pk_add_f16: seven instructions, each sum <= 8 x 256) and the two halves of the result enter
// the fp32 row sum through ONE v_dot2c - eight instructions per block and chain.  Round 2 took a v_dot2c per word (eight,
// but the dot instructions do not run beside another wave's MFMAs: scratch/mb/coexec2.hip, profiles/r03_coexec2.txt - 32
// of them next to 8 MFMAs take 245 ns against 138 | 76 alone, as do v_dot2_f32_f16, v_pk_mul_f32, v_pk_add_f32, while
// v_pk_add_f16 overlaps like any plain vector instruction); the first round-3 form took v_add_f32 on the unrounded
// exponentials (sixteen, -1.5 ... -4 %).  The sum is that of the ROUNDED P, what the P.V product sees.
#define A3_E2(S, K, W)                                             \
  {                                                                \
    hx2 w_;                                                        \
    w_[0] = (h16)__builtin_amdgcn_exp2f(S[2 * (K)]);               \
    w_[1] = (h16)__builtin_amdgcn_exp2f(S[2 * (K) + 1]);           \
    W = w_;                                                        \
  }
// row maximum of one chain's block and, rarely, the move of its running maximum (as in attn2_block, per chain)
template <bool FIRST>
__device__ __forceinline__ void a3_max(f32x16& s, f32x16 (&oacc)[2], f32x16& negm, float& mrun, float& lrun) {
  float m0 = s[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m0 = fmaxf(m0, s[i]);
  const bool raise = FIRST || __builtin_amdgcn_ballot_w64(m0 > ATTN_DEFER_LOG2) != 0;
  if (raise) {
    const float mrow = fmaxf(m0, __shfl_xor(m0, 32));
    const float delta = FIRST ? mrow : fmaxf(mrow, 0.f);
    mrun += delta;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] -= delta;
    if constexpr (!FIRST) {
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      lrun *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] -= delta;
  }
}
__device__ __forceinline__ hx8 a3_vfrag(const u16* sV, int kb, int s2, int dt, int hh, int dblk, int tq, int tp) {
  const int c = 4 * dt + 2 * dblk + (tp >> 1);
  const int row0 = 32 * kb + 16 * s2 + 4 * hh + tq;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(sV + v_off(row0, c) + 4 * (tp & 1)));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(sV + v_off(row0 + 8, c) + 4 * (tp & 1)));
  union { struct { short4_t a, b; } h; hx8 v; } u;
  u.h.a = lo;
  u.h.b = hi;
  return u.v;
}
template <bool TAIL>
__device__ __forceinline__ void a3_mask(f32x16& s, int key0, int nkv, int hh) {
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= nkv) s[i] = -1e30f;
  }
}

// FIRST: the wave's first block (no block of chain 1 pending; the running maxima start here)
template <bool FIRST, bool TAIL>
__device__ __forceinline__ void attn3_block(const u16* sK, const u16* sV, int kb, const hx8 (&qf)[2][4], f32x16 (&oacc)[2][2],
                                            f32x16 (&negm)[2], float (&mrun)[2], float (&lrun)[2], f32x16& s1, hx8 (&va)[2],
                                            hx8 (&vb)[2], int key0, int nkv, int r, int hh, int dblk, int tq, int tp) {
  hx8 kf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const hx8*)&sK[k_off(32 * kb + r, 2 * ks + hh)];
  f32x16 s0;
  A3P p1a, p1b, p0a, p0b;
  // ---- Q0 | E1a
  if constexpr (!FIRST) A3_E2(s1, 0, p1a.h[0]);
  A3_FENCE();
  asm(MFMA32_MNEMONIC " %0, %1, %2, %3" : "=&v"(s0) : "v"(kf[0]), "v"(qf[0][0]), "v"(negm[0]));
  if constexpr (!FIRST) A3_E2(s1, 1, p1a.h[1]);
  A3_FENCE();
  s0 = MFMA32(kf[1], qf[0][1], s0, 0, 0, 0);
  if constexpr (!FIRST) A3_E2(s1, 2, p1a.h[2]);
  A3_FENCE();
  s0 = MFMA32(kf[2], qf[0][2], s0, 0, 0, 0);
  if constexpr (!FIRST) A3_E2(s1, 3, p1a.h[3]);
  A3_FENCE();
  s0 = MFMA32(kf[3], qf[0][3], s0, 0, 0, 0);
  hx2 l1a;
  if constexpr (!FIRST) l1a = A3_SUM4(p1a);
  A3_FENCE();
  // ---- V1a | E1b
  if constexpr (!FIRST) {
    oacc[1][0] = MFMA32(va[0], p1a.v, oacc[1][0], 0, 0, 0);
    A3_E2(s1, 4, p1b.h[0]);
    A3_E2(s1, 5, p1b.h[1]);
    A3_FENCE();
    oacc[1][1] = MFMA32(va[1], p1a.v, oacc[1][1], 0, 0, 0);
    A3_E2(s1, 6, p1b.h[2]);
    A3_E2(s1, 7, p1b.h[3]);
    A3_LADD(lrun[1], l1a, A3_SUM4(p1b));
    A3_FENCE();
  }
  // va is free: this block's V fragments for keys 0-15
  va[0] = a3_vfrag(sV, kb, 0, 0, hh, dblk, tq, tp);
  va[1] = a3_vfrag(sV, kb, 0, 1, hh, dblk, tq, tp);
  // ---- Q1 | max0, E0a   (s1 is free: every E1 of the previous block has run)
  asm(MFMA32_MNEMONIC " %0, %1, %2, %3" : "=&v"(s1) : "v"(kf[0]), "v"(qf[1][0]), "v"(negm[1]));
  a3_mask<TAIL>(s0, key0, nkv, hh);
  a3_max<FIRST>(s0, oacc[0], negm[0], mrun[0], lrun[0]);
  A3_FENCE();
  s1 = MFMA32(kf[1], qf[1][1], s1, 0, 0, 0);
  A3_E2(s0, 0, p0a.h[0]);
  A3_FENCE();
  s1 = MFMA32(kf[2], qf[1][2], s1, 0, 0, 0);
  A3_E2(s0, 1, p0a.h[1]);
  A3_FENCE();
  s1 = MFMA32(kf[3], qf[1][3], s1, 0, 0, 0);
  A3_E2(s0, 2, p0a.h[2]);
  A3_FENCE();
  // ---- V1b | E0a3, E0b0
  if constexpr (!FIRST) oacc[1][0] = MFMA32(vb[0], p1b.v, oacc[1][0], 0, 0, 0);
  A3_E2(s0, 3, p0a.h[3]);
  A3_FENCE();
  if constexpr (!FIRST) oacc[1][1] = MFMA32(vb[1], p1b.v, oacc[1][1], 0, 0, 0);
  A3_E2(s0, 4, p0b.h[0]);
  const hx2 l0a = A3_SUM4(p0a);
  A3_FENCE();
  // vb is free: this block's V fragments for keys 16-31
  vb[0] = a3_vfrag(sV, kb, 1, 0, hh, dblk, tq, tp);
  vb[1] = a3_vfrag(sV, kb, 1, 1, hh, dblk, tq, tp);
  // ---- V0a | E0b
  oacc[0][0] = MFMA32(va[0], p0a.v, oacc[0][0], 0, 0, 0);
  A3_E2(s0, 5, p0b.h[1]);
  A3_FENCE();
  oacc[0][1] = MFMA32(va[1], p0a.v, oacc[0][1], 0, 0, 0);
  A3_E2(s0, 6, p0b.h[2]);
  A3_E2(s0, 7, p0b.h[3]);
  A3_FENCE();
  // ---- V0b | row sum 0, max1
  oacc[0][0] = MFMA32(vb[0], p0b.v, oacc[0][0], 0, 0, 0);
  A3_LADD(lrun[0], l0a, A3_SUM4(p0b));
  a3_mask<TAIL>(s1, key0, nkv, hh);
  a3_max<FIRST>(s1, oacc[1], negm[1], mrun[1], lrun[1]);
  A3_FENCE();
  oacc[0][1] = MFMA32(vb[1], p0b.v, oacc[0][1], 0, 0, 0);
  A3_FENCE();
}
// after the wave's last block: chain 1's pending exp / P.V
__device__ __forceinline__ void attn3_drain(f32x16 (&oacc)[2][2], float (&lrun)[2], const f32x16& s1, const hx8 (&va)[2], const hx8 (&vb)[2]) {
  A3P p1a, p1b;
  A3_E2(s1, 0, p1a.h[0]); A3_E2(s1, 1, p1a.h[1]); A3_E2(s1, 2, p1a.h[2]); A3_E2(s1, 3, p1a.h[3]);
  oacc[1][0] = MFMA32(va[0], p1a.v, oacc[1][0], 0, 0, 0);
  oacc[1][1] = MFMA32(va[1], p1a.v, oacc[1][1], 0, 0, 0);
  A3_E2(s1, 4, p1b.h[0]); A3_E2(s1, 5, p1b.h[1]); A3_E2(s1, 6, p1b.h[2]); A3_E2(s1, 7, p1b.h[3]);
  A3_LADD(lrun[1], A3_SUM4(p1a), A3_SUM4(p1b));
  oacc[1][0] = MFMA32(vb[0], p1b.v, oacc[1][0], 0, 0, 0);
  oacc[1][1] = MFMA32(vb[1], p1b.v, oacc[1][1], 0, 0, 0);
}

// K/V staging of this kernel is LDS-DMA issued from inline asm (M0 = LDS base of the wave's 1 KB piece, then
// global_load_lds_dwordx4): the compiler does not see an LDS write, so fragment reads through run-time selected buffer
// pointers are not held back behind "vmcnt(0)" for a DMA they cannot alias (with the builtin that needs separately
// named arrays and one copy of the loop per buffer order; two streams would need four copies).  The waits are explicit:
// s_waitcnt vmcnt(0) in front of the barrier that ends a tile.  No compiler-generated code in this kernel uses M0.
__device__ __forceinline__ void attn2_dma16(const void* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
}

// BUFS (round 3): K / V staging through buffer descriptors (32-bit offsets; host: nkv * ld * 2 < 2^31) - no 64-bit address
// arithmetic per DMA instruction, and the partial last tile needs no clamped second path: rows past nkv arrive as zeros
// (their scores are masked either way: same bits)
template <bool PIPE, bool BUFS>
__global__ __launch_bounds__(256, 2) void attn64x2_kernel(const AttnParams p) {
  // tile images [stream][buffer][K | V]; afterwards the second stream's O passes through its images to the first stream
  __shared__ __attribute__((aligned(16))) u16 smem[2 * 2 * 2 * 4096];
  __shared__ float sML[2][4 * 64];  // merge: per row group m[2], l[2] per lane
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, hh = lane >> 5;
  const int rg = __builtin_amdgcn_readfirstlane(wv & 1), kh = __builtin_amdgcn_readfirstlane(wv >> 1);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % p.nqt, bh = bid / p.nqt;
  const int head = bh % p.heads, bz = bh / p.heads;
  const u16* Q = p.q + (long long)bz * p.sq + head * 64;
  const u16* K = p.k + (long long)bz * p.sk + head * 64;
  const u16* V = p.v + (long long)bz * p.sv + head * 64;
  u16* O = p.o + (long long)bz * p.so + head * 64;

  // Q fragments, pre-scaled by scale * log2(e) (one more fp16 rounding of q: the scores then leave the MFMA in the log2 domain)
  hx8 qf[2][4];
  int qrow[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = qt * 128 + rg * 64 + qb * 32 + r;
    const int qrc = qrow[qb] < p.nq ? qrow[qb] : p.nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const hx8 raw = *(const hx8*)(Q + (long long)qrc * p.ldq + 16 * ks + 8 * hh);
      if (p.scale_log2 == 1.0f) qf[qb][ks] = raw;  // q already in the log2 domain (sdmi_attention_log2q)
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (h16)((float)raw[j] * p.scale_log2);
    }
  }
  // staging: the two waves of a stream each move 32 rows of K and of V per tile (4 + 4 DMA instructions of 8 rows)
  const int spos = lane & 7;
  u16* const sbase = smem + kh * 4 * 4096;  // this stream's [buffer][K | V]
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) u16*)sbase + rg * 4096;  // bytes
  f32x16 oacc[2][2], negm[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][0][i] = oacc[0][1][i] = oacc[1][0][i] = oacc[1][1][i] = 0.f; negm[0][i] = negm[1][i] = 0.f; }
  float mrun[2] = {0.f, 0.f}, lrun[2] = {0.f, 0.f};
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
  // PIPE: chain 1's scores of the block in flight and the V fragments it still needs (attn3_block)
  f32x16 ps1;
  hx8 pva[2], pvb[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) ps1[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { pva[0][j] = pva[1][j] = pvb[0][j] = pvb[1][j] = (h16)0.f; }
#define ATTN2_BLOCK(FIRST_, TAIL_, CK, CV, KB, K0)                                                                          \
  {                                                                                                                         \
    if constexpr (PIPE) attn3_block<FIRST_, TAIL_>(CK, CV, KB, qf, oacc, negm, mrun, lrun, ps1, pva, pvb, K0, p.nkv, r, hh, dblk, tq, tp); \
    else attn2_block<FIRST_, TAIL_>(CK, CV, KB, qf, oacc, negm, mrun, lrun, K0, p.nkv, r, hh, dblk, tq, tp);              \
  }

  const int ntiles = (p.nkv + 63) / 64;
  const int nbar = (ntiles + 1) / 2;  // tiles of the first stream (the longer one): every wave runs nbar + 1 barriers
  const int t0 = kh ? nbar : 0, t1 = kh ? ntiles : nbar;
  // per-lane byte offsets of the 4 + 4 pieces inside a tile (full tiles: wave-uniform tile base + these; the last,
  // partial tile clamps its rows to the last key instead)
  unsigned offk[4], offv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * rg + 8 * j + (lane >> 3);
    offk[j] = 2u * (row * p.ldk + (spos ^ ((row >> 1) & 7)) * 8);
    offv[j] = 2u * (row * p.ldv + (spos ^ (((row >> 1) & 1) << 2)) * 8);
  }
  const long long stepk = (long long)128 * p.ldk, stepv = (long long)128 * p.ldv;  // bytes per 64-key tile
  auto stage_full = [&](int tile, int buf) {
    const char* kt = (const char*)K + tile * stepk;
    const char* vt = (const char*)V + tile * stepv;
    const unsigned dst = lds0 + buf * 16384;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      attn2_dma16(kt + offk[j], dst + j * 1024);
      attn2_dma16(vt + offv[j], dst + 8192 + j * 1024);
    }
  };
  auto stage_tail = [&](int tile, int buf) {
    const unsigned dst = lds0 + buf * 16384;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 32 * rg + 8 * j + (lane >> 3);
      const int key = min(tile * 64 + row, p.nkv - 1);  // rows past the end repeat the last key (masked in the scores)
      attn2_dma16(K + (long long)key * p.ldk + (spos ^ ((row >> 1) & 7)) * 8, dst + j * 1024);
      attn2_dma16(V + (long long)key * p.ldv + (spos ^ (((row >> 1) & 1) << 2)) * 8, dst + 8192 + j * 1024);
    }
  };
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, BUFS ? (unsigned)((p.nkv - 1) * p.ldk * 2 + 128) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, BUFS ? (unsigned)((p.nkv - 1) * p.ldv * 2 + 128) : 0u, 0x00020000);
  auto stage = [&](int tile, int buf) {
    if constexpr (BUFS) {
      const unsigned dst = lds0 + buf * 16384;
      const unsigned sk_ = tile * (unsigned)stepk, sv_ = tile * (unsigned)stepv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dma_buf16(rsrcK, offk[j] + sk_, 0, dst + j * 1024);
        dma_buf16(rsrcV, offv[j] + sv_, 0, dst + 8192 + j * 1024);
      }
    } else {
      if ((tile + 1) * 64 <= p.nkv) stage_full(tile, buf);
      else stage_tail(tile, buf);
    }
  };
#define ATTN2_END_TILE()                                  \
  {                                                       \
    __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */     \
    asm volatile("" ::: "memory");                        \
    __syncthreads();                                      \
  }
  if (t1 > t0) stage(t0, 0);
  ATTN2_END_TILE();
  int done = 0;
  if (t1 > t0) {
    // first tile of this wave's range: its first block sets the running maximum
    {
      if (t0 + 1 < t1) stage(t0 + 1, 1);
      const u16* cK = sbase;
      const u16* cV = cK + 4096;
      const int k0 = t0 * 64;
      if (k0 + 64 <= p.nkv) {
        ATTN2_BLOCK(true, false, cK, cV, 0, k0);
        ATTN2_BLOCK(false, false, cK, cV, 1, k0 + 32);
      } else {
        ATTN2_BLOCK(true, true, cK, cV, 0, k0);
        if (k0 + 32 < p.nkv) ATTN2_BLOCK(false, true, cK, cV, 1, k0 + 32);
      }
      ATTN2_END_TILE();
      done = 1;
    }
    // full tiles
    const int tfull = min(t1, p.nkv >> 6);
    int tile = t0 + 1;
    for (; tile < tfull; ++tile, ++done) {
      const int buf = done & 1;
      if (tile + 1 < t1) stage(tile + 1, buf ^ 1);
      const u16* cK = sbase + buf * 8192;
      const u16* cV = cK + 4096;
      ATTN2_BLOCK(false, false, cK, cV, 0, tile * 64);
      ATTN2_BLOCK(false, false, cK, cV, 1, tile * 64 + 32);
      ATTN2_END_TILE();
    }
    if (tile < t1) {  // the partial last tile
      const u16* cK = sbase + (done & 1) * 8192;
      const u16* cV = cK + 4096;
      const int k0 = tile * 64;
      ATTN2_BLOCK(false, true, cK, cV, 0, k0);
      if (k0 + 32 < p.nkv) ATTN2_BLOCK(false, true, cK, cV, 1, k0 + 32);
      ATTN2_END_TILE();
      ++done;
    }
  }
  if constexpr (PIPE) if (t1 > t0) attn3_drain(oacc, lrun, ps1, pva, pvb);
  for (; done < nbar; ++done) ATTN2_END_TILE();
#undef ATTN2_END_TILE
#undef ATTN2_BLOCK

  // merge: the second stream's waves hand (O, m, l) to the first stream's wave of the same rows, lane to lane, through the
  // second stream's tile images (every read of them is behind the loop's last barrier)
  float* mgO = (float*)(smem + 4 * 4096) + rg * 4096;  // 64 accumulator registers x 64 lanes per row group (16 KB)
  float* mgS = sML[rg];
  if (kh == 1) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mgO[((qb * 2 + dt) * 16 + i) * 64 + lane] = oacc[qb][dt][i];
      mgS[qb * 64 + lane] = t1 > t0 ? mrun[qb] : -3.0e38f;  // a stream without tiles weighs nothing
      mgS[(2 + qb) * 64 + lane] = lrun[qb];
    }
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float m1 = mgS[qb * 64 + lane], l1 = mgS[(2 + qb) * 64 + lane];
    const float m = fmaxf(mrun[qb], m1);
    const float a0 = __builtin_amdgcn_exp2f(mrun[qb] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    const float lsum = lrun[qb] * a0 + l1 * a1;
    const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32));
    const float w0 = a0 * inv, w1 = a1 * inv;
    if (qrow[qb] < p.nq) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * dt + 8 * g + 4 * hh;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = oacc[qb][dt][4 * g + e] * w0 + mgO[((qb * 2 + dt) * 16 + 4 * g + e) * 64 + lane] * w1;
          *(uint2*)(O + (long long)qrow[qb] * p.ldo + d) = make_uint2(pack2h(o[0], o[1]), pack2h(o[2], o[3]));
        }
    }
  }
}

// ------------------------------------------------------------------------------------------ 8-wave two-group kernel (round 3)
// scratch/mb/coexec.hip on MI355X: two waves that share a SIMD and BOTH carry mixed MFMA + VALU streams cost the SUM of
// their MFMA and VALU times (16 MFMAs + 128 VALU: 369 ns, against 256 + 151 alone), however the streams are interleaved;
// an MFMA-only wave beside a VALU-only wave costs about the longer of the two (230 ns against 148 | 199).  attn64x2_kernel
// is the first case.  This kernel makes it the second.  Block = 8 waves = two groups of four (waves w and w + 4 share a
// SIMD: scratch/mb/hwid.hip); every wave alternates a MATRIX phase - O^T += V^T P^T of block i - 1, then S^T = K Q^T of
// block i: 16 MFMAs out of registers, no LDS read, no vector work - with a VECTOR phase - the fragment reads of V (block i)
// and K (block i + 1), mask / running maximum / exp2 / pack / row sums of block i: no MFMA - and group B runs one phase
// behind group A; an s_barrier of all eight waves ends every phase, so that on every SIMD one wave's matrix phase always
// faces the other's vector phase.
// Work split: 128 query rows x 4 quarters of the key range; wave w = (row group w & 1, quarter w >> 1): group A sweeps
// quarters 0 / 1, group B quarters 2 / 3.  A block = 32 keys; every quarter streams its blocks through a ring of four
// 8 KB images [K | V] (LDS-DMA, each of the quarter's two waves brings half): 128 KB of LDS, one block per CU.  The DMA of
// block i + 3 is issued at the end of matrix phase i (its image held block i - 1, whose last read was in vector phase
// i - 1) and must have landed at the end of matrix phase i + 2: five phases.  (m, l, O) of the four quarters meet in the
// LDS at the end.
// Measured on the way here (s_memrealtime stamps per phase, 2 x 5 x 9216 x 9216): with the fragment reads at the start of
// the matrix phase the 16 MFMAs took 380-410 ns to issue instead of 221 (LDS latency under load ~120 ns, and the phase
// boundary is where all eight waves read at once); with 64-key tiles the eight DMA instructions of a tile cost their
// wave ~170 ns of issue in one phase.
__global__ __launch_bounds__(512, 2) void attn64pp_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];  // [quarter][ring slot][K | V] images of 32 keys; merge area afterwards
  __shared__ float sML[3][2][2 * 64];                          // merge: m per (quarter 1..3, row group, chain) and lane
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 31, hh = lane >> 5;
  const int rg = wv & 1, kq = wv >> 1, grp = wv >> 2;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % p.nqt, bh = bid / p.nqt;
  const int head = bh % p.heads, bz = bh / p.heads;
  const u16* Q = p.q + (long long)bz * p.sq + head * 64;
  const u16* K = p.k + (long long)bz * p.sk + head * 64;
  const u16* V = p.v + (long long)bz * p.sv + head * 64;
  u16* O = p.o + (long long)bz * p.so + head * 64;

  hx8 qf[2][4];
  int qrow[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = qt * 128 + rg * 64 + qb * 32 + r;
    const int qrc = qrow[qb] < p.nq ? qrow[qb] : p.nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const hx8 raw = *(const hx8*)(Q + (long long)qrc * p.ldq + 16 * ks + 8 * hh);
      if (p.scale_log2 == 1.0f) qf[qb][ks] = raw;
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (h16)((float)raw[j] * p.scale_log2);
    }
  }
  const int spos = lane & 7;
  u16* const sbase = smem + kq * 4 * 4096;  // this quarter's ring: 4 slots x (K 2048 | V 2048) u16
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) u16*)sbase + rg * 2048;  // bytes: this wave's 16 rows
  // negm: a tile of -(running maximum), the C operand of every block's first S^T MFMA: the scores come out of the matrix
  // phase already relative to the maximum (as in attn2_block)
  f32x16 oacc[2][2], negm[2], sc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][0][i] = oacc[0][1][i] = oacc[1][0][i] = oacc[1][1][i] = 0.f; negm[0][i] = negm[1][i] = 0.f; sc[0][i] = sc[1][i] = 0.f; }
  float mrun[2] = {0.f, 0.f}, lrun[2] = {0.f, 0.f};
  const int dblk = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;

  // key range of this quarter in 32-key blocks [b0, b0 + nb)
  const int nblk = (p.nkv + 31) / 32;
  const int nper = (nblk + 3) / 4;
  const int b0 = min(nblk, kq * nper), nb = min(nblk, b0 + nper) - b0;
  const int nphase_all = 2 * nper + 2;  // barriers every wave runs: start + group offset + 2 per block

  // staging through buffer descriptors, 32-bit per-lane offsets (host: nkv * ld * 2 < 2^31).  The descriptors end with
  // key nkv - 1: the rows of a partial last block past it arrive as ZEROS (scratch/mb/oob_lds.hip: a lane whose offset is
  // out of range writes zeros through the LDS path too) - their scores are masked, and 0 x 0 keeps P.V finite.  Blocks
  // past the quarter's end are "staged" with a descriptor of no records (zeros into a free slot): every matrix phase
  // issues the same four DMA instructions, and vmcnt counts blocks.
  unsigned offk[2], offv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * rg + 8 * j + (lane >> 3);
    offk[j] = 2u * (row * p.ldk + (spos ^ ((row >> 1) & 7)) * 8);
    offv[j] = 2u * (row * p.ldv + (spos ^ (((row >> 1) & 1) << 2)) * 8);
  }
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (p.nkv - 1) * p.ldk * 2 + 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (p.nkv - 1) * p.ldv * 2 + 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcZ = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, 0, 0x00020000);
  const int stepk = 64 * p.ldk, stepv = 64 * p.ldv;  // bytes per 32-key block
  auto stage = [&](int i) {                          // block i of this quarter -> slot i & 3
    const unsigned dst = lds0 + (i & 3) * 8192;
    const bool live = i < nb;
    const unsigned sk_ = (b0 + i) * stepk, sv_ = (b0 + i) * stepv;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma_buf16(live ? rsrcK : rsrcZ, offk[j] + sk_, 0, dst + j * 1024);
      dma_buf16(live ? rsrcV : rsrcZ, offv[j] + sv_, 0, dst + 4096 + j * 1024);
    }
  };
  // every phase ends here: this wave's LDS reads are complete, then all eight waves meet.  MX: a matrix phase ends - (a) the
  // DMA of the block whose K the next vector phase reads has landed: all but the two youngest stages (8 instructions);
  // (b) the MFMAs are asm statements, which the compiler's hazard recognizer does not see into: the s_nops cover the wait
  // states between an MFMA's last pass and a vector read of its result (every vector read of sc / oacc is behind such a
  // barrier)
#define PP_BAR(MX)                                                                                    \
  {                                                                                                   \
    if (MX) __builtin_amdgcn_s_waitcnt(0x0078); /* vmcnt(8) lgkmcnt(0) */                             \
    else __builtin_amdgcn_s_waitcnt(0xC07F);    /* lgkmcnt(0) */                                      \
    if (MX) asm volatile("s_nop 15\n\ts_nop 3");                                                      \
    __builtin_amdgcn_sched_barrier(0); /* nothing drifts into the other kind of phase */              \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                     \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  hx8 kf[4], vf[2][2];
  A3P pb[2][2];  // P of the block in flight: [chain][keys 0-15 | 16-31]
#define PP_READ_K(I)                                                                                         \
  {                                                                                                          \
    const u16* cK = sbase + ((I) & 3) * 4096;                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const hx8*)&cK[k_off(r, 2 * ks + hh)];       \
  }
#define PP_READ_V(I)                                                                                         \
  {                                                                                                          \
    const u16* cV = sbase + ((I) & 3) * 4096 + 2048;                                                         \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                         \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) vf[s2][dt] = a3_vfrag(cV, 0, s2, dt, hh, dblk, tq, tp); \
  }
  // the MFMAs as asm statements with the accumulator tied in place: with the builtin the register allocator is free to put
  // an MFMA's result beside its C operand, and under this loop's pressure it does (a second copy of O^T, v_mov_b64 trains)
#define PP_MFMA(ACC, A, B) asm volatile(MFMA32_MNEMONIC " %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define PP_MFMA0(D, A, B, C_) asm volatile(MFMA32_MNEMONIC " %0, %1, %2, %3" : "=&v"(D) : "v"(A), "v"(B), "v"(C_))
#define PP_PV()                                                                                              \
  {                                                                                                          \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                         \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                       \
        _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) PP_MFMA(oacc[qb][dt], vf[s2][dt], pb[qb][s2].v);    \
  }
#define PP_QK()                                                                                              \
  {                                                                                                          \
    PP_MFMA0(sc[0], kf[0], qf[0][0], negm[0]);                                                               \
    PP_MFMA0(sc[1], kf[0], qf[1][0], negm[1]);                                                               \
    _Pragma("unroll") for (int ks = 1; ks < 4; ++ks)                                                         \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) PP_MFMA(sc[qb], kf[ks], qf[qb][ks]);                  \
  }
  // vector phase of block i.  Its results - and the fragments it has read for the next matrix phase - are "used" at its
  // end: without that LLVM sinks half of the exponentials below the barrier, into the matrix phase (they are needed only
  // there).  (With both a masked and an unmasked softmax body inside the loop the register allocator ran out - 256 VGPRs,
  // spills and 64 v_mov_b64 per block: the masked block stands outside the loop.)
#define PP_SOFTMAX(FIRST_, TAIL_, I)                                                                         \
  {                                                                                                          \
    PP_READ_V(I);                                                                                            \
    if ((I) + 1 < nb) PP_READ_K((I) + 1);                                                                    \
    const int key0 = (b0 + (I)) * 32;                                                                        \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                       \
      a3_mask<TAIL_>(sc[qb], key0, p.nkv, hh);                                                               \
      a3_max<FIRST_>(sc[qb], oacc[qb], negm[qb], mrun[qb], lrun[qb]);                                        \
      A3_E2(sc[qb], 0, pb[qb][0].h[0]); A3_E2(sc[qb], 1, pb[qb][0].h[1]);                                    \
      A3_E2(sc[qb], 2, pb[qb][0].h[2]); A3_E2(sc[qb], 3, pb[qb][0].h[3]);                                    \
      A3_E2(sc[qb], 4, pb[qb][1].h[0]); A3_E2(sc[qb], 5, pb[qb][1].h[1]);                                    \
      A3_E2(sc[qb], 6, pb[qb][1].h[2]); A3_E2(sc[qb], 7, pb[qb][1].h[3]);                                    \
      A3_LADD(lrun[qb], A3_SUM4(pb[qb][0]), A3_SUM4(pb[qb][1]));                                             \
    }                                                                                                        \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                         \
      asm volatile("" : "+v"(pb[qb][0].v), "+v"(pb[qb][1].v), "+v"(negm[qb]), "+v"(lrun[qb]), "+v"(oacc[qb][0]), "+v"(oacc[qb][1])); \
    asm volatile("" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]));                        \
    asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));                                   \
  }

  stage(0); stage(1); stage(2);
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
  PP_BAR(false);
  int bars = 1;
  if (grp) { PP_BAR(false); ++bars; }  // group B runs one phase behind
  if (nb > 0) {
    const bool last_partial = (p.nkv & 31) != 0 && b0 + nb == nblk;  // the quarter's last block holds keys past nkv
    // block 0: its K fragments are read here, S^T only
    PP_READ_K(0);
    PP_QK();
    stage(3);
    PP_BAR(true); ++bars;
    if (nb == 1 && last_partial) PP_SOFTMAX(true, true, 0) else PP_SOFTMAX(true, false, 0)
    PP_BAR(false); ++bars;
  // the 16 MFMAs of a matrix phase in an order that reuses no accumulator within four instructions (one wave alone does
  // not fill the pipe with dependent pairs: 17.3 ns per MFMA against 13.8), the four DMA instructions of block i + 3 spread
  // between them (issued together they wait for the address unit: ~100 ns at the end of the phase)
#define PP_BLOCK(TAIL_, I)                                                                                   \
  {                                                                                                          \
    const unsigned dst_ = lds0 + (((I) + 3) & 3) * 8192;                                                     \
    const bool live_ = (I) + 3 < nb;                                                                         \
    const unsigned sk_ = (b0 + (I) + 3) * stepk, sv_ = (b0 + (I) + 3) * stepv;                               \
    PP_MFMA0(sc[0], kf[0], qf[0][0], negm[0]);                                                               \
    PP_MFMA0(sc[1], kf[0], qf[1][0], negm[1]);                                                               \
    PP_MFMA(oacc[0][0], vf[0][0], pb[0][0].v);                                                               \
    PP_MFMA(oacc[1][0], vf[0][0], pb[1][0].v);                                                               \
    dma_buf16(live_ ? rsrcK : rsrcZ, offk[0] + sk_, 0, dst_);                                                \
    PP_MFMA(sc[0], kf[1], qf[0][1]);                                                                         \
    PP_MFMA(sc[1], kf[1], qf[1][1]);                                                                         \
    PP_MFMA(oacc[0][1], vf[0][1], pb[0][0].v);                                                               \
    PP_MFMA(oacc[1][1], vf[0][1], pb[1][0].v);                                                               \
    dma_buf16(live_ ? rsrcV : rsrcZ, offv[0] + sv_, 0, dst_ + 4096);                                         \
    PP_MFMA(sc[0], kf[2], qf[0][2]);                                                                         \
    PP_MFMA(sc[1], kf[2], qf[1][2]);                                                                         \
    PP_MFMA(oacc[0][0], vf[1][0], pb[0][1].v);                                                               \
    PP_MFMA(oacc[1][0], vf[1][0], pb[1][1].v);                                                               \
    dma_buf16(live_ ? rsrcK : rsrcZ, offk[1] + sk_, 0, dst_ + 1024);                                         \
    PP_MFMA(sc[0], kf[3], qf[0][3]);                                                                         \
    PP_MFMA(sc[1], kf[3], qf[1][3]);                                                                         \
    PP_MFMA(oacc[0][1], vf[1][1], pb[0][1].v);                                                               \
    PP_MFMA(oacc[1][1], vf[1][1], pb[1][1].v);                                                               \
    dma_buf16(live_ ? rsrcV : rsrcZ, offv[1] + sv_, 0, dst_ + 4096 + 1024);                                  \
    PP_BAR(true); ++bars;                                                                                    \
    PP_SOFTMAX(false, TAIL_, I)                                                                              \
    PP_BAR(false); ++bars;                                                                                   \
  }
    const int nloop = (last_partial && nb > 1) ? nb - 1 : nb;
    for (int i = 1; i < nloop; ++i) PP_BLOCK(false, i)
    if (nloop < nb) PP_BLOCK(true, nb - 1)
#undef PP_BLOCK
    PP_PV();  // the last block's P.V
  }
  for (; bars < nphase_all; ++bars) PP_BAR(false);
  __builtin_amdgcn_s_waitcnt(0x0070);  // the stages past the end have landed before the ring becomes the merge area
  __syncthreads();
#undef PP_READ_K
#undef PP_PV
#undef PP_QK
#undef PP_READ_V
#undef PP_SOFTMAX
#undef PP_MFMA
#undef PP_MFMA0

  // merge: quarters 1 - 3 hand (O, m, l) to quarter 0's wave of the same rows, lane to lane, through the tile images
  // (every read of them is behind the last barrier)
  float* mgO = (float*)smem;  // [quarter - 1][row group][64 accumulator registers][64 lanes]: 3 x 2 x 16 KB
  if (kq > 0) {
    float* o_ = mgO + ((kq - 1) * 2 + rg) * 4096;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o_[((qb * 2 + dt) * 16 + i) * 64 + lane] = oacc[qb][dt][i];
      sML[kq - 1][rg][qb * 64 + lane] = nb > 0 ? mrun[qb] : -3.0e38f;  // a quarter without keys weighs nothing
    }
  }
  // l travels in a second pass through the same words' neighbours: keep it simple - a separate array in the merge area
  float* mgL = mgO + 3 * 2 * 4096;  // [quarter - 1][row group][chain][lane]
  if (kq > 0) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) mgL[(((kq - 1) * 2 + rg) * 2 + qb) * 64 + lane] = lrun[qb];
  }
  __syncthreads();
  if (kq > 0) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mq[3], lq[3];
    float m = nb > 0 ? mrun[qb] : -3.0e38f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      mq[q] = sML[q][rg][qb * 64 + lane];
      lq[q] = mgL[((q * 2 + rg) * 2 + qb) * 64 + lane];
      m = fmaxf(m, mq[q]);
    }
    const float a0 = nb > 0 ? __builtin_amdgcn_exp2f(mrun[qb] - m) : 0.f;
    float aq[3];
    float lsum = lrun[qb] * a0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { aq[q] = __builtin_amdgcn_exp2f(mq[q] - m); lsum += lq[q] * aq[q]; }
    const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32));
    if (qrow[qb] < p.nq) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * dt + 8 * g + 4 * hh;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float acc = oacc[qb][dt][4 * g + e] * a0;
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += mgO[(q * 2 + rg) * 4096 + ((qb * 2 + dt) * 16 + 4 * g + e) * 64 + lane] * aq[q];
            o[e] = acc * inv;
          }
          *(uint2*)(O + (long long)qrow[qb] * p.ldo + d) = make_uint2(pack2h(o[0], o[1]), pack2h(o[2], o[3]));
        }
    }
  }
#undef PP_BAR
}

static int attention_launch(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v, int ldv,
                            int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq, int nkv, float scale,
                            bool log2q, void* stream) {
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
  p.nq = nq; p.nkv = nkv; p.scale_log2 = log2q ? 1.0f : scale * 1.44269504088896340736f;
  // long self-attention: the 64-rows-per-wave structure (attn_variant: -1 never, 1 always, 0 = by shape)
  const int variant = sdmi_opt(OPT_attn_variant);
  const int min_nq = sdmi_opt(OPT_attn_x2_min_nq);
  if (variant > 0 || (variant == 0 && nkv >= 1024 && nq >= (min_nq > 0 ? min_nq : 2048))) {
    p.nqt = (nq + 127) / 128; p.heads = heads;
    // two-group 8-wave kernel: on request only (attn_pp = 1) - it does not beat attn64x2_kernel
    const bool pp_ok = (long long)nkv * ldk * 2 < (1ll << 31) && (long long)nkv * ldv * 2 < (1ll << 31);  // 32-bit K / V offsets
    if (pp_ok && sdmi_opt(OPT_attn_pp) > 0) {
      constexpr int bytes = 4 * 2 * 2 * 4096 * 2;  // 128 KB
      static unsigned long long attr_done = 0;
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!((attr_done >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute((const void*)attn64pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_done |= 1ull << (dev & 63);
      }
      hipLaunchKernelGGL(attn64pp_kernel, dim3(p.nqt * heads * batch), dim3(512), bytes, (hipStream_t)stream, p);
      sdmi_note_kernel("attn64pp_kernel");
      return sdmi_check_launch("attention (two wave groups)");
    }
    // attn_pipe_off = 1: the lockstep block of round 2 (A/B switch; same bits)
    if (sdmi_opt(OPT_attn_pipe_off)) hipLaunchKernelGGL((attn64x2_kernel<false, false>), dim3(p.nqt * heads * batch), dim3(256), 0, (hipStream_t)stream, p);
    else if (pp_ok && !sdmi_opt(OPT_attn_buf_off)) hipLaunchKernelGGL((attn64x2_kernel<true, true>), dim3(p.nqt * heads * batch), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn64x2_kernel<true, false>), dim3(p.nqt * heads * batch), dim3(256), 0, (hipStream_t)stream, p);
    sdmi_note_kernel("attn64x2_kernel");
    return sdmi_check_launch("attention (64 rows per wave)");
  }
  p.nqt = (nq + 127) / 128; p.heads = heads;
  dim3 grid(p.nqt * heads * batch), block(256);
  hipLaunchKernelGGL(attn64_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
  sdmi_note_kernel("attn64_kernel");
  return sdmi_check_launch("attention");
}

extern "C" int sdmi_cross_attention_lnq(const void* x, int ldx, int64_t sx, int c, const void* wq_packed, int ldw, const float* ln_u,
                                        const float* ln_v, const float* rowstats, int parts, float eps, const void* k, int ldk,
                                        int64_t sk, const void* v, int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch,
                                        int heads, int nq, int nkv, void* stream) {
  SDMI_REQUIRE(x && wq_packed && ln_u && ln_v && rowstats && k && v && o, "cross_attention_lnq: null pointer");
  SDMI_REQUIRE(batch > 0 && heads > 0 && nq > 0 && nkv > 0 && c > 0 && c % 64 == 0 && parts > 0 && eps > 0.f, "cross_attention_lnq: bad shape (c=%d must be a multiple of 64)", c);
  SDMI_REQUIRE(ldx % 8 == 0 && ldx >= c && ldw % 8 == 0 && ldw >= c && sx % 8 == 0, "cross_attention_lnq: x / weight strides");
  SDMI_REQUIRE(ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && ldk >= heads * 64 && ldv >= heads * 64 && ldo >= heads * 64, "cross_attention_lnq: k / v / o strides");
  SDMI_REQUIRE((((uintptr_t)x | (uintptr_t)wq_packed | (uintptr_t)k | (uintptr_t)v | (uintptr_t)ln_u | (uintptr_t)ln_v) & 15) == 0 &&
                   (((uintptr_t)o | (uintptr_t)rowstats) & 7) == 0, "cross_attention_lnq: pointer alignment");
  SDMI_REQUIRE(sk % 8 == 0 && sv % 8 == 0 && so % 4 == 0 && ldk < (1 << 22) && ldv < (1 << 22), "cross_attention_lnq: k / v strides");
  SDMI_REQUIRE((long long)((nq + 127) / 128) * heads * batch < (1ll << 31), "cross_attention_lnq: grid too large");
  AttnParams p = {};
  p.q = nullptr; p.k = (const u16*)k; p.v = (const u16*)v; p.o = (u16*)o;
  p.sq = 0; p.sk = sk; p.sv = sv; p.so = so; p.ldq = 0; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.nq = nq; p.nkv = nkv; p.scale_log2 = 1.0f; p.nqt = (nq + 127) / 128; p.heads = heads;
  p.x = (const u16*)x; p.sx = sx; p.ldx = ldx; p.wq = (const u16*)wq_packed; p.ldwq = ldw; p.lnu = ln_u; p.lnv = ln_v;
  p.rstats = rowstats; p.parts = parts; p.eps = eps; p.C = c;
  SDMI_REQUIRE(c % 320 == 0 && c <= 640, "cross_attention_lnq: c must be 320 or 640 (the head's weights are staged whole in the LDS; wider streams: LN-folded GEMM + sdmi_attention_log2q)");
  const int lds = ((64 * (c / 8 + 1) + 255) / 256) * 256 * 16;
  {
    static unsigned long long attr_done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      (void)hipFuncSetAttribute((const void*)attn64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_done |= 1ull << (dev & 63);
    }
  }
  hipLaunchKernelGGL(attn64_kernel<true>, dim3(p.nqt * heads * batch), dim3(256), lds, (hipStream_t)stream, p);
  sdmi_note_kernel("attn64_kernel<true>");
  return sdmi_check_launch("cross attention with query projection");
}

extern "C" int sdmi_attention(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                              int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq, int nkv,
                              float scale, void* stream) {
  return attention_launch(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, batch, heads, nq, nkv, scale, false, stream);
}

extern "C" int sdmi_attention_log2q(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                                    int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq,
                                    int nkv, void* stream) {
  return attention_launch(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, batch, heads, nq, nkv, 1.0f, true, stream);
}

// ------------------------------------------------------------------------------------------ causal attention, short sequences
// The text encoder's self-attention (77 tokens, 16 heads of 64, additive causal mask: open_clip build_attention_mask,
// used at LatentDiffusionInpainting.py:241): runs once per prompt on a few thousand rows, so it is written for clarity,
// not for the matrix cores: one wave per (batch, head, query row), the head dimension on the lanes, fp32 online softmax
// over the keys 0 .. row.
__global__ __launch_bounds__(256) void attn_causal_rows_kernel(const AttnParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * 4 + wv;
  const long long total = (long long)p.nq * p.heads * p.nqt;  // nqt carries the batch here
  if (wid >= total) return;
  const int row = (int)(wid % p.nq);
  const int head = (int)((wid / p.nq) % p.heads), bz = (int)(wid / ((long long)p.nq * p.heads));
  const u16* Q = p.q + (long long)bz * p.sq + head * 64;
  const u16* K = p.k + (long long)bz * p.sk + head * 64;
  const u16* V = p.v + (long long)bz * p.sv + head * 64;
  const float q = h2f(Q[(long long)row * p.ldq + lane]) * p.scale_log2;
  float m = -1e30f, l = 0.f, acc = 0.f;
  for (int j = 0; j <= row && j < p.nkv; ++j) {
    const float s = wave_sum(q * h2f(K[(long long)j * p.ldk + lane]));
    const float mnew = fmaxf(m, s);
    const float a = __builtin_amdgcn_exp2f(m - mnew), e = __builtin_amdgcn_exp2f(s - mnew);
    l = l * a + e;
    acc = acc * a + e * h2f(V[(long long)j * p.ldv + lane]);
    m = mnew;
  }
  p.o[(long long)bz * p.so + (long long)row * p.ldo + head * 64 + lane] = f2h(acc / l);
}

extern "C" int sdmi_attention_causal(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                                     int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int n,
                                     float scale, void* stream) {
  SDMI_REQUIRE(q && k && v && o, "attention_causal: null pointer");
  SDMI_REQUIRE(batch > 0 && heads > 0 && n > 0 && n <= 4096, "attention_causal: bad shape (sequences up to 4096 tokens)");
  SDMI_REQUIRE(ldq >= heads * 64 && ldk >= heads * 64 && ldv >= heads * 64 && ldo >= heads * 64, "attention_causal: row stride < heads*64");
  SDMI_REQUIRE(scale > 0.f, "attention_causal: scale must be positive");
  AttnParams p;
  p.q = (const u16*)q; p.k = (const u16*)k; p.v = (const u16*)v; p.o = (u16*)o;
  p.sq = sq; p.sk = sk; p.sv = sv; p.so = so; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.nq = n; p.nkv = n; p.heads = heads; p.nqt = batch; p.scale_log2 = scale * 1.44269504088896340736f;
  const long long waves = (long long)n * heads * batch;
  SDMI_REQUIRE((waves + 3) / 4 < (1ll << 31), "attention_causal: grid too large");
  hipLaunchKernelGGL(attn_causal_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  sdmi_note_kernel("attn_causal_rows_kernel");
  return sdmi_check_launch("attention_causal");
}
