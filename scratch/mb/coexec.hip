// Microbenchmark: do MFMA and VALU work of DIFFERENT waves on one SIMD overlap?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only, 1: VALU only, 2: both back to back in each wave, 3: both, interleaved 1 MFMA : NV/8 VALU,
// 4: blocks alternate roles (even block MFMA only, odd block VALU only)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void mb(float* out, int iters) {
  const int lane = threadIdx.x;
  hx8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.001f + j); b[j] = (_Float16)(lane * 0.002f - j); }
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = lane * 0.5f + j;
  const float c0 = 1.0001f, c1 = 0.0001f;
  const bool do_m = MODE == 0 || MODE == 2 || MODE == 3 || (MODE == 4 && ((blockIdx.x >> 8) & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || MODE == 3 || (MODE == 4 && ((blockIdx.x >> 8) & 1) == 1);
#define MF(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[(J) & 1]) : "v"(a), "v"(b))
#define VF(K) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(K) & 7]) : "v"(c0), "v"(c1))
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        MF(j);
#pragma unroll
        for (int k = 0; k < NV / 8; ++k) VF(k);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int j = 0; j < 8; ++j) MF(j);
      }
      if (do_v) {
#pragma unroll
        for (int k = 0; k < NV; ++k) VF(k);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NV>
static void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mb<MODE, NV>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mb<MODE, NV>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s NV=%3d waves/SIMD=%d : %8.1f ns/iter  (%.0f cyc @2.4GHz)\n", name, NV, blocks_per_cu, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 4096);
  for (int w = 1; w <= 4; ++w) {
    run<0, 64>("MFMA only (8 x 32x32x16)", w, out);
    run<1, 64>("VALU only", w, out);
    run<2, 64>("both, back to back per wave", w, out);
    run<3, 64>("both, interleaved per wave", w, out);
    if (w % 2 == 0) run<4, 64>("alternate blocks MFMA / VALU", w, out);
    run<3, 32>("both, interleaved per wave", w, out);
    run<3, 48>("both, interleaved per wave", w, out);
    run<1, 128>("VALU only", w, out);
    run<2, 128>("both, back to back per wave", w, out);
    run<3, 128>("both, interleaved per wave", w, out);
  }
  hipFree(out);
  return 0;
}
