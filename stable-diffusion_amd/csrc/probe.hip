// probe.hip - two measurement kernels behind bench.py's "measured peak" figures (report roofline fractions
// against the nominal AND the measured peak of the box).  No counterpart in the reference; nothing on the product path
// calls them.
#include "common.h"

// Back-to-back v_mfma_f32_32x32x16 on FOUR independent accumulator chains, 8 waves per CU (two per SIMD): the matrix-pipe
// issue rate, nothing else (no memory, no LDS).  2 * 32 * 32 * 16 flop per MFMA and wave.  (Rounds 1-2 ran two chains per
// wave and read 1.89-1.92 PFLOP/s: dependent pairs do not fill the pipe - 17.3 ns per MFMA against 13.8; four chains
// sustain 2.38-2.43, scratch/mb/mfma_peak.hip.)
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* out, int iters) {
  const int lane = threadIdx.x;
  hx8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (h16)(0.001f * lane + j); b[j] = (h16)(0.002f * lane - j); }
  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc0 = MFMA32(a, b, acc0, 0, 0, 0);
      acc1 = MFMA32(a, b, acc1, 0, 0, 0);
      acc2 = MFMA32(a, b, acc2, 0, 0, 0);
      acc3 = MFMA32(a, b, acc3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  if (s == 12345.678f) out[0] = s;  // keeps the loop alive, never true
}

// Streaming copy, 16 bytes per lane, 4 loads in flight per lane, grid-stride.
__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i = (long long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i + j * 256 < n16) v[j] = src[i + j * 256];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i + j * 256 < n16) dst[i + j * 256] = v[j];
  }
}

extern "C" int sdmi_probe_mfma(float* scratch, int blocks, int iters, void* stream) {
  SDMI_REQUIRE(scratch && blocks > 0 && iters > 0, "probe_mfma: bad argument");
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
  return sdmi_check_launch("probe_mfma");
}

extern "C" int sdmi_probe_copy(const void* src, void* dst, int64_t bytes, void* stream) {
  SDMI_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0, "probe_copy: bytes must be a positive multiple of 16");
  SDMI_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "probe_copy: alignment");
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                     (long long)(bytes / 16));
  return sdmi_check_launch("probe_copy");
}
