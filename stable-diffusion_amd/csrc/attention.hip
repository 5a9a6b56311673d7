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
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    hx8 pb[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[qb][j] = (h16)__builtin_amdgcn_exp2f(s[qb][8 * s2 + j]);
      // row sums of the rounded P (what the PV product sees), two elements per instruction
      union { hx8 v; hx2 h[4]; } u;
      u.v = pb[qb];
      const hx2 ones = {(h16)1.0f, (h16)1.0f};
#pragma unroll
      for (int j = 0; j < 4; ++j) lrun[qb] = DOT2H(u.h[j], ones, lrun[qb]);
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

// K/V staging of this kernel is LDS-DMA issued from inline asm (M0 = LDS base of the wave's 1 KB piece, then
// global_load_lds_dwordx4): the compiler does not see an LDS write, so fragment reads through run-time selected buffer
// pointers are not held back behind "vmcnt(0)" for a DMA they cannot alias (with the builtin that needs separately
// named arrays and one copy of the loop per buffer order; two streams would need four copies).  The waits are explicit:
// s_waitcnt vmcnt(0) in front of the barrier that ends a tile.  No compiler-generated code in this kernel uses M0.
__device__ __forceinline__ void attn2_dma16(const void* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
}

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
  auto stage = [&](int tile, int buf) {
    if ((tile + 1) * 64 <= p.nkv) stage_full(tile, buf);
    else stage_tail(tile, buf);
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
        attn2_block<true, false>(cK, cV, 0, qf, oacc, negm, mrun, lrun, k0, p.nkv, r, hh, dblk, tq, tp);
        attn2_block<false, false>(cK, cV, 1, qf, oacc, negm, mrun, lrun, k0 + 32, p.nkv, r, hh, dblk, tq, tp);
      } else {
        attn2_block<true, true>(cK, cV, 0, qf, oacc, negm, mrun, lrun, k0, p.nkv, r, hh, dblk, tq, tp);
        if (k0 + 32 < p.nkv) attn2_block<false, true>(cK, cV, 1, qf, oacc, negm, mrun, lrun, k0 + 32, p.nkv, r, hh, dblk, tq, tp);
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
      attn2_block<false, false>(cK, cV, 0, qf, oacc, negm, mrun, lrun, tile * 64, p.nkv, r, hh, dblk, tq, tp);
      attn2_block<false, false>(cK, cV, 1, qf, oacc, negm, mrun, lrun, tile * 64 + 32, p.nkv, r, hh, dblk, tq, tp);
      ATTN2_END_TILE();
    }
    if (tile < t1) {  // the partial last tile
      const u16* cK = sbase + (done & 1) * 8192;
      const u16* cV = cK + 4096;
      const int k0 = tile * 64;
      attn2_block<false, true>(cK, cV, 0, qf, oacc, negm, mrun, lrun, k0, p.nkv, r, hh, dblk, tq, tp);
      if (k0 + 32 < p.nkv) attn2_block<false, true>(cK, cV, 1, qf, oacc, negm, mrun, lrun, k0 + 32, p.nkv, r, hh, dblk, tq, tp);
      ATTN2_END_TILE();
      ++done;
    }
  }
  for (; done < nbar; ++done) ATTN2_END_TILE();
#undef ATTN2_END_TILE

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
    hipLaunchKernelGGL(attn64x2_kernel, dim3(p.nqt * heads * batch), dim3(256), 0, (hipStream_t)stream, p);
    sdmi_note_kernel("attn64x2_kernel");
    return sdmi_check_launch("attention (64 rows per wave)");
  }
  p.nqt = (nq + 127) / 128; p.heads = heads;
  dim3 grid(p.nqt * heads * batch), block(256);
  hipLaunchKernelGGL(attn64_kernel, grid, block, 0, (hipStream_t)stream, p);
  sdmi_note_kernel("attn64_kernel");
  return sdmi_check_launch("attention");
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
