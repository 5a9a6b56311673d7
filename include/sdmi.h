/* sdmi.h - C ABI of libsdmi355.so: gfx950 (MI355X) kernels for the SD-2.x inpainting hot path.
 *
 * The reference (ccharest93/Stable-Diffusion, /root/reference) has no FFI: its operator layer is
 * torch.nn / torch.nn.functional plus xformers.ops.memory_efficient_attention.  Each entry point
 * below names the reference operator call(s) it replaces (file:line relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, a negative SDMI_E* code otherwise; sdmi_last_error()
 *     returns a thread-local message.  Nothing throws, exits or synchronises the device.
 *   - all pointers are DEVICE pointers owned by the caller; the library allocates nothing.
 *     Scratch space comes in through explicit workspace arguments.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.
 *   - activations: "NHWC h16" (h16 = the build's 16-bit type, see sdmi_act_dtype()) = [N][H*W][C] row-major with row stride ld (elements, multiple of 8);
 *     this is also the (B, tokens, C) layout of the transformer blocks, so no permutes exist.
 *   - packed weights: h16 [N_out][K], K contiguous (for a 3x3 conv K = (ky*3+kx)*Cin + c).
 */
#ifndef SDMI_H
#define SDMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_OK 0
#define SDMI_EINVAL (-1)   /* bad argument / unsupported shape */
#define SDMI_ELAUNCH (-2)  /* HIP launch error */

/* epilogue / mode flags */
#define SDMI_F_BIAS 1u      /* + bias[n]                       (fp32 [N])                    */
#define SDMI_F_ROWBIAS 2u   /* + rowbias[sample(m)*ldrb + n]   (fp32, row stride ldrb; timestep-emb add) */
#define SDMI_F_RESIDUAL 4u  /* + residual[m][n]                (bf16, stride ldr)            */
#define SDMI_F_GEGLU 8u     /* out[m][j] = x*gelu(gate), weights packed by sdmi_pack_linear(geglu=1) */
#define SDMI_F_OUT_NCHW_F32 16u /* store fp32 NCHW instead of bf16 NHWC (module boundary)   */
#define SDMI_F_UPSAMPLE2X 32u   /* conv: input is read through a nearest 2x upsample gather  */
#define SDMI_F_SILU 64u     /* groupnorm_apply / small linear: SiLU                           */
#define SDMI_F_OUT_F32 128u /* gemm: store fp32 row-major (ldc in elements) instead of bf16  */
#define SDMI_F_OUT_U8_IMAGE 256u /* conv3x3: store uint8 NHWC (n, hout, wout, cout), tightly packed, of
                                   * clamp((y + 1) / 2, 0, 1) * 255 truncated: the image post-processing at
                                   * LatentDiffusionInpainting.py:228-230 fused into the decoder's conv_out */

int sdmi_version(void);
/* 16-bit storage type ("h16") of activations and packed weights this build uses: 0 = IEEE fp16 (default), 1 = bf16 */
int sdmi_act_dtype(void);
const char* sdmi_last_error(void);
/* name of the kernel variant the last sdmi_conv3x3 / sdmi_gemm call on this thread launched (tile shape is chosen per
 * problem), e.g. "igemm_kernel<128, 128, true>": lets a profiler key its numbers on the symbol rocprofv3 reports */
const char* sdmi_last_kernel(void);

/* ---- weight pre-pack (run once per parameter, after load_state_dict) ---------------------- */
/* fp32 OIHW (Cout,Cin,3,3) -> bf16 [Cout][9][Cin].  replaces nothing: layout change for sdmi_conv3x3. */
int sdmi_pack_conv3x3(const float* w_oihw, void* w_packed, int cout, int cin, void* stream);
/* fp32 [N][K] (nn.Linear weight or 1x1 conv) -> bf16 [N][K]; geglu=1 interleaves the value/gate rows
 * of GEGLU.proj in 16-row groups so that x*gelu(gate) is lane-local in the GEMM epilogue. */
int sdmi_pack_linear(const float* w, void* w_packed, int n, int k, int geglu, void* stream);

/* ---- layout at the module boundary (reference tensors are NCHW fp32: SURVEY 8(b)) ---------- */
int sdmi_nchw_f32_to_nhwc_h16(const float* src, void* dst, int n, int c, int hw, int ldd, void* stream);
int sdmi_nhwc_h16_to_nchw_f32(const void* src, float* dst, int n, int c, int hw, int lds, void* stream);
/* fp32 [rows][cols] -> bf16 [rows][ld]  (text context (B,77,1024) entering to_k / to_v) */
int sdmi_f32_to_h16_rows(const float* src, void* dst, int rows, int cols, int ld, void* stream);

/* ---- GroupNorm(32) (+SiLU): nn.GroupNorm + nn.SiLU at layers/ResBlock.py:65-66,:72-73; Unet.py:104-106;
 *      AutoencoderKL.py:193-194; layers/SpatialTransformer.py:134,:212 ----------------------- */
/* bytes of workspace sdmi_groupnorm_stats needs */
int64_t sdmi_groupnorm_ws_bytes(int n, int hw, int c);
/* pass 1: per-(sample,group) statistics -> scale_shift fp32 [N][C][2] = {gamma*rstd, beta-mean*gamma*rstd} */
int sdmi_groupnorm_stats(const void* x, int n, int hw, int c, int ldx, int groups, const float* gamma,
                         const float* beta, float eps, float* scale_shift, void* ws, void* stream);
/* pass 2: y = act(x*scale+shift), bf16 NHWC -> bf16 NHWC (flags: SDMI_F_SILU) */
int sdmi_groupnorm_apply(const void* x, int ldx, const float* scale_shift, void* y, int ldy, int n, int hw, int c,
                         uint32_t flags, void* stream);

/* GroupNorm(+SiLU) as ONE call (same reference lines).  When a (sample, group) slice (hw * c/groups elements) fits the
 * registers of one block it runs as a single launch that reads x once (two-pass variance from registers); otherwise it
 * runs sdmi_groupnorm_stats + sdmi_groupnorm_apply and needs scale_shift (fp32 [N][C][2]) and ws (sdmi_groupnorm_ws_bytes). */
int sdmi_groupnorm(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups, const float* gamma,
                   const float* beta, float eps, uint32_t flags, float* scale_shift, void* ws, void* stream);

/* ---- GroupNorm(+SiLU) that allocates nothing.  NO counterpart in the reference: it exists for callers that cannot
 *      allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 *      Same arithmetic and bits as sdmi_groupnorm; the scale/shift table and the partial sums live in a static device
 *      array of sdmi_groupnorm_static_max_bytes() (1 MiB) per GPU instead of caller-provided buffers.  Shapes that need
 *      more (n*c*8 rounded up to 256 + sdmi_groupnorm_ws_bytes(n, hw, c) bytes) are rejected with SDMI_EINVAL.  Calls on
 *      one GPU share that array: calls that overlap in time on one GPU (different streams) race.  The array's address is
 *      looked up once per device (hipGetSymbolAddress, not a capturable operation): sdmi_groupnorm_static_init() does that
 *      for the current device; the first sdmi_groupnorm_static call on a device does it too unless `stream` is capturing,
 *      in which case it fails with SDMI_EINVAL instead of breaking the capture.  `stream` must belong to the current device. */
int sdmi_groupnorm_static(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups, const float* gamma,
                          const float* beta, float eps, uint32_t flags, void* stream);
int64_t sdmi_groupnorm_static_max_bytes(void);
int sdmi_groupnorm_static_init(void);

/* ---- LayerNorm: nn.LayerNorm at layers/SpatialTransformer.py:177-184 ----------------------- */
int sdmi_layernorm(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy,
                   int rows, int c, void* stream);

/* ---- 3x3 implicit-GEMM convolution on MFMA: nn.Conv2d k=3 at layers/ResBlock.py:67,:74 (pad 1),
 *      Downsample :36-37 (explicit pad + stride 2), Upsample :17-19 (nearest 2x + pad + conv, flag
 *      SDMI_F_UPSAMPLE2X), Unet.py:107, AutoencoderKL.py:172; epilogue fuses the conv bias, the
 *      timestep-embedding add of layers/ResBlock.py:70 (SDMI_F_ROWBIAS) and the skip add of :76
 *      (SDMI_F_RESIDUAL).
 *   x: NHWC bf16 (n,h,w,cin) stride ldx; hout/wout/pad_t/pad_l/stride describe the geometry AFTER the
 *   optional 2x upsample (input pixel = (oy*stride+ky-pad_t, ox*stride+kx-pad_l), zero outside).   */
int sdmi_conv3x3(const void* x, int ldx, int n, int h, int w, int cin, const void* w_packed, int cout,
                 int hout, int wout, int stride, int pad_t, int pad_l, const float* bias, const float* rowbias,
                 int ldrb, const void* residual, int ldr, void* out, int ldc, uint32_t flags, void* ws, int64_t ws_bytes,
                 void* stream);
/* bytes of split-K workspace sdmi_conv3x3 can use for this problem (0 = never splits).  Under-filled grids (few
 * output tiles, long K: the 12x12 / 24x24 UNet levels) are cut along K into fp32 partials that a second kernel sums in
 * a fixed order and finishes with the epilogue.  The factor depends only on the per-sample shape, so results do not
 * depend on the batch.  Passing ws = NULL (or too few bytes) disables the split. */
int64_t sdmi_conv3x3_ws_bytes(int n, int cin, int cout, int hout, int wout);

/* Split-K without the second kernel (sdmi_conv3x3 / sdmi_gemm and their _ex forms): with SDMI_F_WS_TICKETS in `flags` the
 * block that finishes an output tile last sums the partials (same fixed order, same bits) and runs the epilogue itself.
 * Arrivals are counted in a bank of library-owned device counters that belongs to ONE workspace: sdmi_ws_tickets_init(ws)
 * registers a workspace on the current device (16 banks per device: SDMI_EINVAL when all are taken;
 * sdmi_ws_tickets_release(ws) gives a bank back once no launch or captured graph uses the workspace any more).  A
 * launch with the flag on an unregistered workspace runs the reduce kernel as usual.  A workspace serves launches that
 * are ordered against each other (one stream, one chain of a graph); launches that may overlap in time need a workspace
 * each - with or without the flag - and therefore never share counters.  No reference counterpart (the reference has no
 * split-K).  Opt-in: it takes one launch off the dependent chain of an under-filled conv / GEMM, but measured slower
 * than the reduce kernel on MI355X at the UNet's shapes. */
#define SDMI_F_WS_TICKETS 1024u
int sdmi_ws_tickets_init(void* ws, void* stream);
int sdmi_ws_tickets_release(void* ws);

/* ---- first-layer conv with tiny Cin reading the caller's NCHW fp32 tensor directly:
 *      Unet.py:49 (9->320), AutoencoderKL.py:136 (4->512), :61 (3->128).  w is the fp32 OIHW weight. */
int sdmi_conv3x3_in(const float* x_nchw, int n, int cin, int h, int w, const float* w_oihw, const float* bias,
                    int cout, void* out, int ldc, void* stream);
/* the same, also leaving colstats fp32 [n * h * w / 32][cout][2] (per-channel sum / sum of squares of every 32-pixel part, as
 * sdmi_epilogue_ext.colstats) for the GroupNorm that follows (Unet.py:49 -> layers/ResBlock.py:65); h * w % 64 == 0; NULL = none */
int sdmi_conv3x3_in_ex(const float* x_nchw, int n, int cin, int h, int w, const float* w_oihw, const float* bias,
                       int cout, void* out, int ldc, float* colstats, void* stream);

/* ---- GEMM out[M][N] = A[M][K] * W[N][K]^T (+epilogue): nn.Linear and 1x1 nn.Conv2d at
 *      layers/SpatialTransformer.py:22-26 (GEGLU), :62-66, :110-129, :199, :204; layers/ResBlock.py:62;
 *      AutoencoderKL.py:203-204.  batch>1 runs independent problems with element strides sa/sw/sc
 *      (used for the VAE single-head attention score / value products).  rows_per_sample maps a row
 *      to its sample for SDMI_F_ROWBIAS. */
int sdmi_gemm(const void* a, int lda, int64_t sa, const void* w, int ldw, int64_t sw, void* out, int ldc,
              int64_t sc, int m, int n, int k, int batch, const float* bias, const float* rowbias,
              int ldrb, int rows_per_sample, const void* residual, int ldr, float alpha, uint32_t flags, void* ws,
              int64_t ws_bytes, void* stream);
int64_t sdmi_gemm_ws_bytes(int m, int n, int k, int batch, int rows_per_sample, uint32_t flags);

/* ---- epilogue extensions of sdmi_gemm / sdmi_conv3x3 (sdmi_gemm_ex / sdmi_conv3x3_ex; ext = NULL gives the plain calls).
 *      They move the reference's normalisation passes into the kernels on either side of them:
 *      - colstats: the producing conv / GEMM also writes per-channel (sum, sum of squares) of its fp32 results over each
 *        wave tile's rows, fp32 [row_parts][N][2] (fixed order, no atomics); sdmi_groupnorm_from_colstats folds them, so the
 *        GroupNorm that follows (layers/ResBlock.py:65,:72; layers/SpatialTransformer.py:212) never re-reads the tensor for
 *        its statistics.
 *      - rowstats: the producing GEMM also writes per-row (sum, sum of squares) over each column part, fp32
 *        [M][col_parts][2], for a LayerNorm that follows (layers/SpatialTransformer.py:182-184).
 *      - SDMI_F_LNFOLD: this GEMM consumes such row statistics and computes LayerNorm(x) @ W^T without a LayerNorm pass:
 *        w was packed by sdmi_pack_linear_ln (gamma folded into the weights), the epilogue applies
 *        out = rstd[m] * (acc - mean[m] * ln_u[n]) + bias[n], bias = the ln_v of sdmi_pack_linear_ln.
 *      sdmi_*_stats_parts tell how many parts a problem writes (0 = it cannot: odd shapes; a split-K launch leaves colstats
 *      from its reduce kernel but no rowstats); asking for statistics
 *      a launch cannot deliver is an error, never a silent skip. */
#define SDMI_F_LNFOLD 512u
/* GroupNorm(+SiLU) of the RESULT, written by the second pass of a split-K launch (sdmi_epilogue_ext.gn_out): the conv -> GroupNorm
 * -> SiLU sequences of layers/ResBlock.py:67-73 on the 24x24 / 12x12 levels, where every conv splits K and its result is small.
 * y[m][ldy] receives act(GroupNorm(out)); the raw result is still written to `out` when keep_raw != 0 (someone else reads it: the
 * residual stream), else `out` is not touched.  Bit-identical to the plain call followed by sdmi_groupnorm. */
typedef struct sdmi_gn_out {
  void* y;
  int ldy;
  const float* gamma;
  const float* beta;
  float eps;
  int groups;
  uint32_t flags;   /* SDMI_F_SILU or 0 */
  int keep_raw;
} sdmi_gn_out;
typedef struct sdmi_epilogue_ext {
  float* colstats;          /* out, or NULL */
  float* rowstats;          /* out, or NULL */
  const float* ln_rowstats; /* SDMI_F_LNFOLD: [M][ln_parts][2] written by the producer of the GEMM's input */
  const float* ln_u;        /* SDMI_F_LNFOLD: [N] in packed row order (sdmi_pack_linear_ln) */
  int ln_parts;
  float ln_eps;
  const float* gn_in;       /* sdmi_conv3x3_ex only, with SDMI_F_SILU: x is the RAW tensor and this is its GroupNorm table, fp32
                             * [n][cin][2] = (scale, shift) per sample and channel (sdmi_groupnorm_colstats_finalize or
                             * sdmi_groupnorm_stats); the conv computes conv(SiLU(x * scale + shift)) - the GN + SiLU + conv of
                             * layers/ResBlock.py:65-67, :72-74 and AutoencoderKL.py:193-194 in ONE pass over x, bit-identical to
                             * sdmi_groupnorm_apply followed by the plain conv.  Only where one 128-column block covers all output
                             * channels: stride 1, pad 1, cout <= 128, cin % 64 == 0, h*w % 128 == 0; other shapes are an error. */
  const sdmi_gn_out* gn_out; /* NULL, or the GroupNorm that follows (split-K launches with small results only: sdmi_conv3x3_gn_out_ok) */
} sdmi_epilogue_ext;
int sdmi_gemm_ex(const void* a, int lda, int64_t sa, const void* w, int ldw, int64_t sw, void* out, int ldc,
                 int64_t sc, int m, int n, int k, int batch, const float* bias, const float* rowbias,
                 int ldrb, int rows_per_sample, const void* residual, int ldr, float alpha, uint32_t flags, void* ws,
                 int64_t ws_bytes, void* stream, const sdmi_epilogue_ext* ext);
int sdmi_conv3x3_ex(const void* x, int ldx, int n, int h, int w, int cin, const void* w_packed, int cout,
                    int hout, int wout, int stride, int pad_t, int pad_l, const float* bias, const float* rowbias,
                    int ldrb, const void* residual, int ldr, void* out, int ldc, uint32_t flags, void* ws, int64_t ws_bytes,
                    void* stream, const sdmi_epilogue_ext* ext);
/* 1 when sdmi_conv3x3_ex can take sdmi_epilogue_ext.gn_in for this stride-1 pad-1 same-size conv (input row stride ldx), else 0 */
int sdmi_conv3x3_gn_in_ok(int n, int h, int w, int cin, int cout, int ldx);
/* 1 when this conv splits K and its result is small enough for sdmi_epilogue_ext.gn_out, else 0 (flags as for sdmi_conv3x3) */
int sdmi_conv3x3_gn_out_ok(int n, int h, int w, int cin, int cout, int hout, int wout, int stride, int pad_t, int pad_l, uint32_t flags,
                           int groups);
/* parts of the statistics the calls above write for this problem: *row_parts (colstats) and *col_parts (rowstats); 0 where
 * the problem cannot emit them.  Either pointer may be NULL.  (upsample: SDMI_F_UPSAMPLE2X in flags) */
int sdmi_gemm_stats_parts(int m, int n, int k, int batch, int rows_per_sample, uint32_t flags, int* row_parts, int* col_parts);
int sdmi_conv3x3_stats_parts(int n, int h, int w, int cin, int cout, int hout, int wout, int stride, int pad_t, int pad_l,
                             uint32_t flags, int* row_parts);
/* GroupNorm(+SiLU) whose statistics come from colstats written by the producer(s) of x (parts of 32 rows).  Up to two
 * channel segments, in channel order (a skip-connection concat, Unet.py:132, has two producers and a group may straddle
 * them): segment s covers channels [seg_c0[s], seg_c0[s] + seg_c[s]) of x; seg_stats[s] is the producer's array
 * [sample * seg_parts[s] + part][seg_ld[s]][2] (seg_parts = row parts PER SAMPLE = hw / 32, seg_ld = the producer's N) and
 * the segment's channels start at column seg_col0[s] of it.  scale_shift: scratch of sdmi_groupnorm_colstats_ss_bytes(). */
int64_t sdmi_groupnorm_colstats_ss_bytes(int n, int c, int groups);
/* the fold alone: scale_shift[n][c][2] from the producers' colstats, no apply pass (for sdmi_epilogue_ext.gn_in) */
int sdmi_groupnorm_colstats_finalize(int n, int hw, int c, int groups, const float* gamma, const float* beta, float eps, int nseg,
                                     const float* const* seg_stats, const int* seg_parts, const int* seg_ld, const int* seg_col0,
                                     const int* seg_c0, const int* seg_c, float* scale_shift, void* stream);
int sdmi_groupnorm_from_colstats(const void* x, int ldx, void* y, int ldy, int n, int hw, int c, int groups,
                                 const float* gamma, const float* beta, float eps, uint32_t flags, int nseg,
                                 const float* const* seg_stats, const int* seg_parts, const int* seg_ld, const int* seg_col0,
                                 const int* seg_c0, const int* seg_c, float* scale_shift, void* stream);
/* nn.Linear weight with a LayerNorm folded in (layers/SpatialTransformer.py:182-184 feeding :62-66 / :22-26):
 * w_packed[n][k] = h16(gamma[k] * w[n][k]) (geglu: rows interleaved as sdmi_pack_linear does), u[n] = sum_k w_packed[n][k]
 * (fp32, packed row order), v[n] = sum_k beta[k] * w[n][k] + bias[n] (original row order, like a bias; bias may be NULL). */
int sdmi_pack_linear_ln(const float* w, const float* gamma, const float* beta, const float* bias, void* w_packed, float* u,
                        float* v, int n, int k, int geglu, void* stream);

/* tuning hook (benchmarks only; no reference counterpart): force the tile variant (index into the kernel's variant table,
 * 0 = 256x128, 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64) and / or the split-K factor of every following
 * sdmi_conv3x3 / sdmi_gemm call of the calling host thread (thread-local, like the error state); -1 = automatic (the default). */
void sdmi_tune_override(int tile, int splits);
/* tuning hook: 0 = run stride-1 3x3 convolutions on the generic im2col kernel, 1 = on the shared-window kernel (same bits,
 * fewer bytes staged), -1 = automatic (default: shared window wherever it applies, env SDMI_CONV_WIN=0 turns it off). */
void sdmi_tune_conv_window(int on);

/* experiment switch for A/B measurements inside one process (no reference counterpart; defaults are the shipped
 * behaviour): "gemm_small_m_tile", "attn_variant".  Unknown names return SDMI_EINVAL. */
int sdmi_set_option(const char* name, int value);

/* Profiling aid, no counterpart in the reference: `event` (a hipEvent_t, or NULL to clear) is recorded by the NEXT
 * sdmi_gemm / sdmi_conv3x3 call that splits K, on that call's stream, between its main kernel and its fixed-order
 * reduce kernel; one-shot.  bench.py's roofline pass uses it to time the two kernels apart. */
void sdmi_prof_mid_event(void* event);

/* ---- skinny linear for M <= 64 rows: Unet.time_embed (Unet.py:41-45) and every ResBlock.emb_proj
 *      (layers/ResBlock.py:70) batched into one launch. x fp32 [M][K]; w bf16 [N][K]; out fp32 [M][N];
 *      SDMI_F_SILU applies SiLU to x on load. */
int sdmi_linear_small_m(const float* x, const void* w, const float* bias, float* out, int m, int n, int k,
                        uint32_t flags, void* stream);

/* ---- sinusoidal timestep embedding: Unet.py:7-25.  freqs fp32 [half] is the host-computed table
 *      exp(-ln(1e4) i/half); out fp32 [M][2*half] = [cos | sin]. */
int sdmi_timestep_embedding(const int64_t* t, const float* freqs, float* out, int m, int half, void* stream);

/* ---- fused flash attention, head_dim 64: xformers.ops.memory_efficient_attention + the head
 *      split/merge reshapes at layers/SpatialTransformer.py:76-95.  q/k/v/o are bf16 token-major
 *      (B, tokens, heads*64) views with row strides ld* and per-batch element strides s*; head h lives
 *      in columns [64h, 64h+64).  o = softmax(q k^T * scale) v. */
int sdmi_attention(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                   int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq, int nkv,
                   float scale, void* stream);
/* ---- cross-attention with its query projection inside the kernel: LayerNorm -> to_q -> memory_efficient_attention of
 *      layers/SpatialTransformer.py:184, :62, :76-95 in ONE launch (the text context's K / V are projected once per prompt).
 *      x: the UN-normalised token stream (B, nq, c), row stride ldx, batch stride sx; rowstats: [B * nq][parts][2] (sum, sumsq)
 *      of its rows, left by its producer (sdmi_gemm_ex, ext.rowstats); wq_packed / ln_u / ln_v: to_q through
 *      sdmi_pack_linear_ln with scale * log2(e) folded into the weights ([heads * 64][c]); k / v / o as sdmi_attention.
 *      c = 320 or 640 (the head's 64 weight rows are staged whole in the LDS).  Not bit-identical to sdmi_gemm_ex(LNFOLD) + sdmi_attention_log2q (another MFMA shape sums K in another
 *      order): equal to it within the rounding of q. */
int sdmi_cross_attention_lnq(const void* x, int ldx, int64_t sx, int c, const void* wq_packed, int ldw, const float* ln_u,
                             const float* ln_v, const float* rowstats, int parts, float eps, const void* k, int ldk, int64_t sk,
                             const void* v, int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq,
                             int nkv, void* stream);
/*      Same, for q that already carries scale * log2(e): o = softmax_base2(q k^T) v.  The caller folds the factor into the
 *      query projection's weights before they are rounded to 16 bits (to_q at layers/SpatialTransformer.py:60,78), so the
 *      kernel exponentiates the MFMA result as it stands and q sees no second rounding. */
int sdmi_attention_log2q(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                         int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int nq, int nkv,
                         void* stream);

/* ---- fused flash attention, ONE head of dim 512: the VAE mid block's attention (layers/SpatialTransformer.py:132-161,
 *      xformers.ops.memory_efficient_attention on (B, HW, 512)) without an N x N score buffer.  q/k/v/o are h16 token-major
 *      (B, n, >= 512) views with row strides ld* and per-batch element strides s*; o = softmax(q k^T * scale) v. */
int sdmi_attention_d512(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v, int ldv,
                        int64_t sv, void* o, int ldo, int64_t so, int batch, int n, float scale, void* stream);
/*      Same for q that already carries scale * log2(e) (folded into the q projection's weight and bias). */
int sdmi_attention_d512_log2q(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v,
                              int ldv, int64_t sv, void* o, int ldo, int64_t so, int batch, int n, void* stream);

/* ---- causal multi-head attention (head_dim 64) for short sequences: the text encoder's self-attention with open_clip's
 *      additive causal mask (LatentDiffusionInpainting.py:238-241, r(x, attn_mask=...)).  Same views as sdmi_attention;
 *      query i attends to keys 0..i.  Runs once per prompt (77 tokens): a plain fp32 kernel, not an MFMA one. */
int sdmi_attention_causal(const void* q, int ldq, int64_t sq, const void* k, int ldk, int64_t sk, const void* v, int ldv,
                          int64_t sv, void* o, int ldo, int64_t so, int batch, int heads, int n, float scale, void* stream);

/* ---- text encoder pieces (LatentDiffusionInpainting.py:232-245, get_conditional):
 *      sdmi_embed_tokens: out[row] = h16(table[tokens[row]] + pos[row % seq_len]) (token_embedding + positional_embedding,
 *      :234-235); tokens int64 [rows], table fp32 [vocab][width], pos fp32 [seq_len][width], out h16 [rows][ldo].
 *      sdmi_gelu_rows: exact (erf) GELU in place on h16 [rows][ldx] (the MLP activation of a resblock, :241). */
int sdmi_embed_tokens(const int64_t* tokens, const float* table, const float* pos, void* out, int rows, int seq_len, int width,
                      int vocab, int ldo, void* stream);
int sdmi_gelu_rows(void* x, int ldx, int rows, int cols, void* stream);

/* ---- inpainting pre-processing: pad_image + make_batch_sd + the mask branch of inpaint()'s c_cat loop
 *      (LatentDiffusionInpainting.py:58-63, :65-89, :281-285) in one pass.  image uint8 HWC (h0, w0, 3); mask uint8
 *      (h0, w0, mask_channels), 3 = RGB (reduced to PIL's "L"), 1 = L; (h, w) the padded size (edge padding on the
 *      bottom / right, as np.pad(mode='edge')).  Outputs, each optional (NULL = skip), fp32 NCHW repeated num_samples
 *      times: image = px / 127.5 - 1, mask = (L / 255 >= 0.5), masked_image = image * (mask < 0.5), and mask_latent =
 *      F.interpolate(mask, (h/8, w/8)) (nearest) written at mask_latent + s * mask_latent_stride (floats; e.g. channel 0
 *      of the (n, 5, h/8, w/8) c_concat tensor).  Exact fp32 arithmetic: bit-identical to the reference's host code. */
int sdmi_inpaint_prepare(const void* image_hwc_u8, const void* mask_u8, int h0, int w0, int mask_channels, int h, int w,
                         int num_samples, float* image, float* mask, float* masked_image, float* mask_latent,
                         int64_t mask_latent_stride, void* stream);

/* ---- posterior sample of the VAE encoder times the latent scale: AutoencoderKL.py:8-21 (clamp, exp(0.5 logvar),
 *      mean + std * noise) and LatentDiffusionInpainting.py:289-291.  moments fp32 (n, 8, h, w) = [mean | logvar]; noise
 *      fp32 (n, 4, h, w); out + s * out_sample_stride receives sample s's 4 channels (channels 1..4 of c_concat). */
int sdmi_posterior_sample(const float* moments, const float* noise, float scale, float* out, int64_t out_sample_stride,
                          int n, int hw, void* stream);

/* ---- fused classifier-free guidance + DDIM update: LatentDiffusionInpainting.py:198-212.
 *   eps fp32 NCHW (2b,4,h,w) = [uncond | cond]; x, noise fp32 (b,4,h,w); coef = {sqrt_one_minus_at, a_t,
 *   a_prev, sigma_t} as the fp32 values torch.full() would hold; writes x_out (may alias x), pred_x0
 *   (optional, may be NULL) and, when xc_out != NULL, copies x_out into channels [0,4) of both CFG halves
 *   of the next UNet input xc (2b, xc_channels, h, w). */
int sdmi_cfg_ddim_step(const float* eps, const float* x, const float* noise, float guidance, float s1m, float a_t,
                       float a_prev, float sigma, float temperature, float* x_out, float* pred_x0, float* xc_out,
                       int xc_channels, int b, int chw, void* stream);

/* ---- range audit, no counterpart in the reference (fp32 there never clips).  The fp16 build SATURATES on store: a value
 *      beyond +-65504 is stored as +-65504, silently.  sdmi_count_saturated adds to *count_u64 (device uint64, zeroed by the
 *      caller) the elements of an h16 [rows][ld] view (first `cols` columns) with |x| >= 65504 or non-finite, so that a
 *      caller can audit a checkpoint layer by layer (stable-diffusion_amd/engine.py RANGE_AUDIT does) and switch to the bf16
 *      build (-DSDMI_BF16) when any layer clips. */
int sdmi_count_saturated(const void* x, int ld, int rows, int cols, void* count_u64, void* stream);

/* ---- measurement probes, no counterpart in the reference (roofline fractions are reported against the
 *      nominal and the MEASURED peak of the box).  sdmi_probe_mfma launches `blocks` blocks of 4 waves that each issue
 *      8 * iters v_mfma_f32_32x32x16 (2*32*32*16 flop each) and nothing else; sdmi_probe_copy streams `bytes` from src
 *      to dst with 16-byte accesses.  The caller times them (HIP events).  scratch: >= 4 bytes of device memory. */
int sdmi_probe_mfma(float* scratch, int blocks, int iters, void* stream);
int sdmi_probe_copy(const void* src, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
