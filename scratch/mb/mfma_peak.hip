// What the matrix cores of THIS box sustain: every SIMD of the chip runs a stream of independent v_mfma_f32_32x32x16_f16
// (4 accumulators per wave, nothing else), 1 or 2 waves per SIMD, short and long runs (clock ramp / power management).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K16>
__global__ __launch_bounds__(512) void mfma_stream(float* out, int iters, unsigned long long* clk) {
  hx8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (K16) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 8 * 512 * 4); hipMalloc(&clk, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) for (int iters : {2000, 20000, 200000}) {
    const int blocks = 256;
    hipLaunchKernelGGL(mfma_stream<1>, dim3(blocks), dim3(threads), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_stream<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)blocks * (threads / 64) * iters * 16.0;
    printf("waves/SIMD %d  iters %6d: %8.3f ms  %7.1f TFLOP/s  %5.2f ns per MFMA per SIMD  (s_memtime ticks %llu)\n", threads / 256, iters, ms,
           nm * 32768.0 / ms * 1e-9, ms * 1e6 / (nm / 1024.0), c);
  }
  return 0;
}
