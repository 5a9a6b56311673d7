// elementwise.hip - boundary layout transforms, weight pre-pack, the tiny-Cin first convolution, the skinny
// (M <= 64) linear of the timestep-embedding path, the sinusoidal embedding, and the fused CFG + DDIM update.
#include "common.h"

// ------------------------------------------------------------------------------------ layout
// NCHW fp32 <-> NHWC bf16 through a 32(pixels) x 32(channels) LDS tile (+1 pad), both sides coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, u16* __restrict__ dst, int c, int hw, int ldd) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int ch = c0 + i, px = p0 + tx;
    tile[i][tx] = (ch < c && px < hw) ? src[((long long)n * c + ch) * hw + px] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int px = p0 + i, ch = c0 + tx;
    if (px < hw && ch < c) dst[((long long)n * hw + px) * ldd + ch] = f2h(tile[tx][i]);
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const u16* __restrict__ src, float* __restrict__ dst, int c, int hw, int lds) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int px = p0 + i, ch = c0 + tx;
    tile[i][tx] = (px < hw && ch < c) ? h2f(src[((long long)n * hw + px) * lds + ch]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int ch = c0 + i, px = p0 + tx;
    if (ch < c && px < hw) dst[((long long)n * c + ch) * hw + px] = tile[tx][i];
  }
}
extern "C" int sdmi_nchw_f32_to_nhwc_h16(const float* src, void* dst, int n, int c, int hw, int ldd, void* stream) {
  SDMI_REQUIRE(src && dst && n > 0 && c > 0 && hw > 0 && ldd >= c && n <= 65535, "nchw_to_nhwc: bad argument");
  dim3 grid((hw + 31) / 32, (c + 31) / 32, n);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, c, hw, ldd);
  return sdmi_check_launch("nchw_to_nhwc");
}
extern "C" int sdmi_nhwc_h16_to_nchw_f32(const void* src, float* dst, int n, int c, int hw, int lds, void* stream) {
  SDMI_REQUIRE(src && dst && n > 0 && c > 0 && hw > 0 && lds >= c && n <= 65535, "nhwc_to_nchw: bad argument");
  dim3 grid((hw + 31) / 32, (c + 31) / 32, n);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)src, dst, c, hw, lds);
  return sdmi_check_launch("nhwc_to_nchw");
}

__global__ __launch_bounds__(256) void f32_to_bf16_rows_kernel(const float* __restrict__ src, u16* __restrict__ dst, int cols, int ld, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cols;
    const int cidx = (int)(i - r * cols);
    dst[r * ld + cidx] = f2h(src[i]);
  }
}
extern "C" int sdmi_f32_to_h16_rows(const float* src, void* dst, int rows, int cols, int ld, void* stream) {
  SDMI_REQUIRE(src && dst && rows > 0 && cols > 0 && ld >= cols, "f32_to_bf16_rows: bad argument");
  const long long total = (long long)rows * cols;
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, cols, ld, total);
  return sdmi_check_launch("f32_to_bf16_rows");
}

// ------------------------------------------------------------------------------------ weight pre-pack
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, u16* __restrict__ o, int cin, long long total) {
  // o[co][tap][ci] = w[co][ci][tap]
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % cin);
    const long long r = i / cin;
    const int tap = (int)(r % 9);
    const long long co = r / 9;
    o[i] = f2h(w[(co * cin + ci) * 9 + tap]);
  }
}
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ w, u16* __restrict__ o, int n, int k, int geglu, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / k;
    const int kk = (int)(i - row * k);
    long long srow = row;
    if (geglu) {  // packed row 32*b + 16*s + j  <-  s ? gate row n/2 + 16*b + j : value row 16*b + j
      const long long b = row >> 5;
      const int s = (int)((row >> 4) & 1), j = (int)(row & 15);
      srow = (s ? (long long)(n >> 1) : 0) + 16 * b + j;
    }
    o[i] = f2h(w[srow * k + kk]);
  }
}
extern "C" int sdmi_pack_conv3x3(const float* w_oihw, void* w_packed, int cout, int cin, void* stream) {
  SDMI_REQUIRE(w_oihw && w_packed && cout > 0 && cin > 0, "pack_conv3x3: bad argument");
  const long long total = (long long)cout * 9 * cin;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_oihw, (u16*)w_packed, cin, total);
  return sdmi_check_launch("pack_conv3x3");
}
extern "C" int sdmi_pack_linear(const float* w, void* w_packed, int n, int k, int geglu, void* stream) {
  SDMI_REQUIRE(w && w_packed && n > 0 && k > 0, "pack_linear: bad argument");
  SDMI_REQUIRE(!geglu || n % 32 == 0, "pack_linear: geglu needs n%%32==0 (n=%d)", n);
  const long long total = (long long)n * k;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, (u16*)w_packed, n, k, geglu, total);
  return sdmi_check_launch("pack_linear");
}

// LayerNorm folded into the Linear that follows it: one wave per packed row.  w'[n][k] = h16(gamma[k] w[n][k]); u[row] = the
// fp32 sum of the ROUNDED packed row (what the MFMA sums when every input element equals the row mean); v[n] = beta . w[n] + b[n].
__global__ __launch_bounds__(256) void pack_linear_ln_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             u16* __restrict__ o, float* __restrict__ u, float* __restrict__ v,
                                                             int n, int k, int geglu) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // packed row
  if (row >= n) return;
  int srow = row;
  if (geglu) {
    const int b = row >> 5, s = (row >> 4) & 1, j = row & 15;
    srow = (s ? (n >> 1) : 0) + 16 * b + j;
  }
  const float* wr = w + (long long)srow * k;
  float su = 0.f, sv = 0.f;
  for (int kk = lane; kk < k; kk += 64) {
    const float wv = wr[kk];
    const u16 h = f2h(gamma[kk] * wv);
    o[(long long)row * k + kk] = h;
    su += h2f(h);
    sv += beta[kk] * wv;
  }
  su = wave_sum(su);
  sv = wave_sum(sv);
  if (lane == 0) {
    u[row] = su;
    v[srow] = sv + (bias ? bias[srow] : 0.f);
  }
}
extern "C" int sdmi_pack_linear_ln(const float* w, const float* gamma, const float* beta, const float* bias, void* w_packed, float* u,
                                   float* v, int n, int k, int geglu, void* stream) {
  SDMI_REQUIRE(w && gamma && beta && w_packed && u && v && n > 0 && k > 0, "pack_linear_ln: bad argument");
  SDMI_REQUIRE(!geglu || n % 32 == 0, "pack_linear_ln: geglu needs n%%32==0 (n=%d)", n);
  hipLaunchKernelGGL(pack_linear_ln_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, bias, (u16*)w_packed, u, v,
                     n, k, geglu);
  return sdmi_check_launch("pack_linear_ln");
}

// ------------------------------------------------------------------------------------ first conv (tiny Cin)
// x NCHW fp32 (the caller's tensor), w OIHW fp32, out NHWC h16.  lane = pixel (64 consecutive pixels per wave: coalesced
// NCHW reads), wave = group of 8 output channels: the weight index is wave-uniform, so the 8 x 9*CIN weights arrive
// through scalar loads and every FMA takes its weight from an SGPR; the 9*CIN input taps sit in registers.
// colstats (may be null; host: h * w % 64 == 0): per-channel (sum, sumsq) of the fp32 results over each 32-pixel part, the
// layout the conv / GEMM epilogues of csrc/igemm.hip leave for sdmi_groupnorm_from_colstats: the GroupNorms that read this
// conv's output (the first ResBlock's norm1 and, through the skip connection, the last output block's) skip their statistics pass.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, u16* __restrict__ out, int h,
                                                         int wd, int cout, int ldc, float* __restrict__ colstats) {
  const int n = blockIdx.z;
  const int hw = h * wd;
  const int pix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int cg = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // group of 8 output channels (wave-uniform)
  if (pix >= hw || cg * 8 >= cout) return;
  const int oy = pix / wd, ox = pix - oy * wd;
  float in[CIN * 9];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      in[ci * 9 + tap] = (iy >= 0 && iy < h && ix >= 0 && ix < wd) ? x[((long long)n * CIN + ci) * hw + iy * wd + ix] : 0.f;
    }
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int co = cg * 8 + e;
    float acc = 0.f;
    if (co < cout) {
      acc = bias[co];
      const float* wp = w + (long long)co * CIN * 9;
#pragma unroll
      for (int i = 0; i < CIN * 9; ++i) acc += in[i] * wp[i];
    }
    o[e] = acc;
  }
  u16* op = out + ((long long)n * hw + pix) * ldc + cg * 8;
  if (cg * 8 + 7 < cout) *(uint4*)op = pack8(o);
  else for (int e = 0; e < 8; ++e) if (cg * 8 + e < cout) op[e] = f2h(o[e]);
  if (colstats) {  // every lane of the wave is live here (h * w % 64 == 0): sums over the 32 lanes of each half, fixed order
    const long long part = ((long long)n * hw + pix) >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float sm = o[e], sq = o[e] * o[e];
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) { sm += __shfl_xor(sm, d); sq += __shfl_xor(sq, d); }
      if ((threadIdx.x & 31) == 0 && cg * 8 + e < cout) *(float2*)(colstats + (part * cout + cg * 8 + e) * 2) = make_float2(sm, sq);
    }
  }
}
extern "C" int sdmi_conv3x3_in(const float* x_nchw, int n, int cin, int h, int w, const float* w_oihw, const float* bias,
                               int cout, void* out, int ldc, void* stream) {
  return sdmi_conv3x3_in_ex(x_nchw, n, cin, h, w, w_oihw, bias, cout, out, ldc, nullptr, stream);
}
extern "C" int sdmi_conv3x3_in_ex(const float* x_nchw, int n, int cin, int h, int w, const float* w_oihw, const float* bias,
                                  int cout, void* out, int ldc, float* colstats, void* stream) {
  SDMI_REQUIRE(x_nchw && w_oihw && bias && out, "conv3x3_in: null pointer");
  SDMI_REQUIRE(!colstats || ((h * w) % 64 == 0 && ((uintptr_t)colstats & 7) == 0), "conv3x3_in: colstats need h * w %% 64 == 0 and an 8-byte aligned buffer");
  SDMI_REQUIRE(n > 0 && h > 0 && w > 0 && cout > 0 && n <= 65535, "conv3x3_in: bad shape");
  SDMI_REQUIRE(ldc % 8 == 0 && ldc >= cout && ((uintptr_t)out & 15) == 0, "conv3x3_in: ldc/alignment");
  dim3 grid((h * w + 63) / 64, ((cout + 7) / 8 + 3) / 4, n), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (cin) {
    case 3: hipLaunchKernelGGL(conv3x3_in_kernel<3>, grid, block, 0, st, x_nchw, w_oihw, bias, (u16*)out, h, w, cout, ldc, colstats); break;
    case 4: hipLaunchKernelGGL(conv3x3_in_kernel<4>, grid, block, 0, st, x_nchw, w_oihw, bias, (u16*)out, h, w, cout, ldc, colstats); break;
    case 9: hipLaunchKernelGGL(conv3x3_in_kernel<9>, grid, block, 0, st, x_nchw, w_oihw, bias, (u16*)out, h, w, cout, ldc, colstats); break;
    default: sdmi_set_error("conv3x3_in: cin=%d unsupported (3, 4, 9)", cin); return SDMI_EINVAL;
  }
  return sdmi_check_launch("conv3x3_in");
}

// ------------------------------------------------------------------------------------ skinny linear
// out[m][n] = bias[n] + sum_k f(x[m][k]) w[n][k]; one wave per output column n, lanes split K (16-byte
// weight loads, weights streamed exactly once), rows handled 8 at a time.
__global__ __launch_bounds__(256) void linear_small_m_kernel(const float* __restrict__ x, const u16* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out, int m,
                                                             int n, int k, int silu) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= n) return;
  const u16* wr = w + (long long)col * k;
  for (int m0 = 0; m0 < m; m0 += 8) {
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    for (int kk = lane * 8; kk < k; kk += 512) {
      float wf[8];
      unpack8(*(const uint4*)(wr + kk), wf);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (m0 + r < m) {
          const float4 a = *(const float4*)(x + (long long)(m0 + r) * k + kk);
          const float4 b = *(const float4*)(x + (long long)(m0 + r) * k + kk + 4);
          float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = silu ? silu_f(xv[e]) : xv[e];
            acc[r] += v * wf[e];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float s = wave_sum(acc[r]);
      if (lane == 0 && m0 + r < m) out[(long long)(m0 + r) * n + col] = s + (bias ? bias[col] : 0.f);
    }
  }
}
extern "C" int sdmi_linear_small_m(const float* x, const void* w, const float* bias, float* out, int m, int n, int k,
                                   uint32_t flags, void* stream) {
  SDMI_REQUIRE(x && w && out && m > 0 && n > 0 && k > 0, "linear_small_m: bad argument");
  SDMI_REQUIRE(m <= 64, "linear_small_m: m=%d > 64 (use sdmi_gemm)", m);
  SDMI_REQUIRE(k % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "linear_small_m: k%%8 / alignment");
  hipLaunchKernelGGL(linear_small_m_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (const u16*)w, bias, out,
                     m, n, k, (flags & SDMI_F_SILU) ? 1 : 0);
  return sdmi_check_launch("linear_small_m");
}

// ------------------------------------------------------------------------------------ timestep embedding
__global__ void timestep_embedding_kernel(const long long* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                                          int m, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * half) return;
  const int r = i / half, j = i - r * half;
  const float a = (float)t[r] * freqs[j];  // fp32 product as in the reference; the trig in fp64 then rounds once
  out[(long long)r * 2 * half + j] = (float)cos((double)a);
  out[(long long)r * 2 * half + half + j] = (float)sin((double)a);
}
extern "C" int sdmi_timestep_embedding(const int64_t* t, const float* freqs, float* out, int m, int half, void* stream) {
  SDMI_REQUIRE(t && freqs && out && m > 0 && half > 0, "timestep_embedding: bad argument");
  const int total = m * half;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)t,
                     freqs, out, m, half);
  return sdmi_check_launch("timestep_embedding");
}

// ------------------------------------------------------------------------------------------ inpainting pre-processing
// pad_image + make_batch_sd + the mask branch of inpaint()'s c_cat loop (LatentDiffusionInpainting.py:58-89, :281-285) in one
// pass over the padded image: edge padding to (H, W), image / 127.5 - 1, mask -> L -> {0, 1}, image * (mask < 0.5), the
// repeat over num_samples, and the nearest-neighbour resize of the mask to (H/8, W/8) (F.interpolate's default picks
// source pixel (8i, 8j)).  Every value is exact fp32 arithmetic on uint8 inputs: bit-identical to the reference's numpy /
// torch host code.
__global__ __launch_bounds__(256) void inpaint_prepare_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ msk,
                                                              int h0, int w0, int mc, int H, int W, int n,
                                                              float* __restrict__ image_out, float* __restrict__ mask_out,
                                                              float* __restrict__ masked_out, float* __restrict__ mask_lat,
                                                              long long mask_lat_stride) {
  const long long hw = (long long)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const int ys = y < h0 ? y : h0 - 1, xs = x < w0 ? x : w0 - 1;  // np.pad(mode='edge') on the bottom / right
    const unsigned char* ip = img + ((long long)ys * w0 + xs) * 3;
    const unsigned char* mp = msk + ((long long)ys * w0 + xs) * mc;
    // PIL "L" from RGB (ITU-R 601-2, fixed point); a single-channel mask is its own L
    const unsigned l8 = mc == 1 ? mp[0] : ((unsigned)mp[0] * 19595u + (unsigned)mp[1] * 38470u + (unsigned)mp[2] * 7471u + 0x8000u) >> 16;
    const float m = ((float)l8 / 255.0f) < 0.5f ? 0.f : 1.f;
    float px[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = (float)ip[c] / 127.5f - 1.0f;
    for (int s = 0; s < n; ++s) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (image_out) image_out[((long long)s * 3 + c) * hw + i] = px[c];
        if (masked_out) masked_out[((long long)s * 3 + c) * hw + i] = m < 0.5f ? px[c] : px[c] * 0.0f;  // x * False keeps the sign of zero
      }
      if (mask_out) mask_out[(long long)s * hw + i] = m;
      if (mask_lat && (y & 7) == 0 && (x & 7) == 0) mask_lat[(long long)s * mask_lat_stride + (long long)(y >> 3) * (W >> 3) + (x >> 3)] = m;
    }
  }
}

extern "C" int sdmi_inpaint_prepare(const void* image_hwc_u8, const void* mask_u8, int h0, int w0, int mask_channels, int h, int w,
                                    int num_samples, float* image, float* mask, float* masked_image, float* mask_latent,
                                    int64_t mask_latent_stride, void* stream) {
  SDMI_REQUIRE(image_hwc_u8 && mask_u8, "inpaint_prepare: null input");
  SDMI_REQUIRE(h0 > 0 && w0 > 0 && h >= h0 && w >= w0 && num_samples > 0, "inpaint_prepare: bad sizes");
  SDMI_REQUIRE(mask_channels == 1 || mask_channels == 3, "inpaint_prepare: mask has 1 (L) or 3 (RGB) channels");
  SDMI_REQUIRE(!mask_latent || (h % 8 == 0 && w % 8 == 0), "inpaint_prepare: the latent mask needs h, w multiples of 8");
  const long long hw = (long long)h * w;
  const int blocks = (int)((hw + 255) / 256 < 4096 ? (hw + 255) / 256 : 4096);
  hipLaunchKernelGGL(inpaint_prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)image_hwc_u8,
                     (const unsigned char*)mask_u8, h0, w0, mask_channels, h, w, num_samples, image, mask, masked_image, mask_latent,
                     (long long)mask_latent_stride);
  return sdmi_check_launch("inpaint_prepare");
}

// DiagonalGaussianDistribution.sample() times the latent scale factor (AutoencoderKL.py:8-21, LatentDiffusionInpainting.py:289-291):
// out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise), written into 4 channels of a (n, C, h, w) tensor
// (the c_concat buffer, channels 1..4).
__global__ __launch_bounds__(256) void posterior_sample_kernel(const float* __restrict__ moments, const float* __restrict__ noise,
                                                               float scale, float* __restrict__ out, long long out_stride, int n,
                                                               int hw) {
  const long long total = (long long)n * 4 * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int s = (int)(i / (4ll * hw));
    const long long r = i - (long long)s * 4 * hw;  // channel * hw + pixel
    const float mean = moments[(long long)s * 8 * hw + r];
    const float logvar = fminf(fmaxf(moments[(long long)s * 8 * hw + 4ll * hw + r], -30.0f), 20.0f);
    out[(long long)s * out_stride + r] = scale * (mean + expf(0.5f * logvar) * noise[i]);
  }
}

extern "C" int sdmi_posterior_sample(const float* moments, const float* noise, float scale, float* out, int64_t out_sample_stride,
                                     int n, int hw, void* stream) {
  SDMI_REQUIRE(moments && noise && out, "posterior_sample: null pointer");
  SDMI_REQUIRE(n > 0 && hw > 0 && out_sample_stride >= 4ll * hw, "posterior_sample: bad sizes");
  const long long total = (long long)n * 4 * hw;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, moments, noise, scale, out,
                     (long long)out_sample_stride, n, hw);
  return sdmi_check_launch("posterior_sample");
}

// ------------------------------------------------------------------------------------------ text encoder pieces
// token_embedding(tokens) + positional_embedding (LatentDiffusionInpainting.py:234-235; open_clip CLIP.encode_text):
// out[b*T + t][:] = h16(table[tokens[b][t]][:] + pos[t][:]).  One block per token row.
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table,
                                                           const float* __restrict__ pos, u16* __restrict__ out, int T, int width,
                                                           int vocab, int ldo) {
  const int row = blockIdx.x, tpos = row % T;
  long long tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);  // host checks the range; never read outside the table
  const float* src = table + tok * (long long)width;
  const float* ps = pos + (long long)tpos * width;
  for (int c = threadIdx.x * 4; c < width; c += 1024) {
    const float4 a = *(const float4*)(src + c), b = *(const float4*)(ps + c);
    *(uint2*)(out + (long long)row * ldo + c) = make_uint2(pack2h(a.x + b.x, a.y + b.y), pack2h(a.z + b.z, a.w + b.w));
  }
}

extern "C" int sdmi_embed_tokens(const int64_t* tokens, const float* table, const float* pos, void* out, int rows, int seq_len,
                                 int width, int vocab, int ldo, void* stream) {
  SDMI_REQUIRE(tokens && table && pos && out, "embed_tokens: null pointer");
  SDMI_REQUIRE(rows > 0 && seq_len > 0 && rows % seq_len == 0 && width > 0 && width % 4 == 0 && vocab > 0 && ldo >= width && ldo % 4 == 0,
               "embed_tokens: bad sizes (rows=%d seq_len=%d width=%d)", rows, seq_len, width);
  SDMI_REQUIRE((((uintptr_t)table | (uintptr_t)pos) & 15) == 0 && ((uintptr_t)out & 7) == 0, "embed_tokens: alignment");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, tokens, table, pos, (u16*)out, seq_len, width, vocab, ldo);
  return sdmi_check_launch("embed_tokens");
}

// exact (erf) GELU in place on an h16 matrix: the activation between c_fc and c_proj of the text encoder's MLP
// (open_clip ResidualAttentionBlock.mlp = Linear -> nn.GELU() -> Linear).
__global__ __launch_bounds__(256) void gelu_rows_kernel(u16* __restrict__ x, int ldx, int rows, int cols) {
  const int cc = cols >> 3;
  const long long total = (long long)rows * cc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cc), c = (int)(i - (long long)r * cc);
    uint4* ptr = (uint4*)(x + (long long)r * ldx + c * 8);
    float f[8];
    unpack8(*ptr, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.5f * f[e] * (1.0f + erff(f[e] * 0.70710678118654752f));
    *ptr = pack8(f);
  }
}

extern "C" int sdmi_gelu_rows(void* x, int ldx, int rows, int cols, void* stream) {
  SDMI_REQUIRE(x && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols, "gelu_rows: cols and ldx must be multiples of 8");
  SDMI_REQUIRE(((uintptr_t)x & 15) == 0, "gelu_rows: alignment");
  const long long total = (long long)rows * (cols / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gelu_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (u16*)x, ldx, rows, cols);
  return sdmi_check_launch("gelu_rows");
}

// ------------------------------------------------------------------------------------ range audit (debug aid)
// The fp16 build saturates on store (f2h16): an activation beyond +-65504 is clipped, silently.  This counts the elements of
// an h16 [rows][ld] view (cols used) that sit AT the clip value (|x| >= 65504, or non-finite) into *count (uint64, the caller
// zeroes it): the drop-in's range audit (engine.RANGE_AUDIT) calls it behind every conv / GEMM to say which layer of a
// given checkpoint would need the bf16 build.  Integer atomics: the count does not depend on the order of arrival.
__global__ __launch_bounds__(256) void count_saturated_kernel(const u16* __restrict__ x, int ld, int rows, int cols,
                                                              unsigned long long* __restrict__ count) {
  unsigned local = 0;
  const long long total = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const float v = fabsf(h2f(x[(long long)r * ld + c]));
    local += !(v < 65504.0f);   // true for the clip value, inf and NaN
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, (unsigned long long)local);
}

extern "C" int sdmi_count_saturated(const void* x, int ld, int rows, int cols, void* count_u64, void* stream) {
  SDMI_REQUIRE(x && count_u64 && rows > 0 && cols > 0 && ld >= cols, "count_saturated: bad argument");
  SDMI_REQUIRE(((uintptr_t)count_u64 & 7) == 0, "count_saturated: the counter must be 8-byte aligned");
  const long long total = (long long)rows * cols;
  long long blocks = (total + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(count_saturated_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)x, ld, rows, cols,
                     (unsigned long long*)count_u64);
  return sdmi_check_launch("count_saturated");
}
