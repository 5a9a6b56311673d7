// Do packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32) ever give a wrong result on MI355X when other waves of the
// SIMD run MFMAs?  (round 2: a compiler-generated v_pk chain in the GEMM epilogue's statistics lost about one square in
// 10^4, never the same one; round 3: these instructions do not overlap with another wave's MFMAs - coexec2.hip - so they
// appear to share the matrix pipe.)  All values are small integers: every sum is exact in fp32, any difference is an error.
//   mode 0: waves 4-7 run the packed chain (asm), waves 0-3 exit         mode 1: waves 0-3 run MFMAs beside it
//   mode 2: every wave alternates its own MFMAs (builtin) and a compiler-generated packed chain on the MFMA results
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void stress(unsigned long long* bad, int iters) {
  extern __shared__ char smem_[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  hx8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)((lane + j) % 3); b[j] = (_Float16)((lane * 3 + j) % 2); }
  f32x16 acc[2];
  unsigned long long nbad = 0;
  if (MODE == 2) {
    for (int it = 0; it < iters; ++it) {
      for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 1], 0, 0, 0);
      // packed chain on the results (the compiler places the wait states it knows of)
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
      float sr = 0.f, qr = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 v = {acc[0][i], acc[0][i + 1]}, w = {acc[1][i], acc[1][i + 1]};
        s2 += v; q2 += v * v; s2 += w; q2 += w * w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float t;
        asm volatile("v_mul_f32 %2, %3, %3\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %3" : "+v"(sr), "+v"(qr), "=&v"(t) : "v"(acc[0][i]));
        asm volatile("v_mul_f32 %2, %3, %3\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %3" : "+v"(sr), "+v"(qr), "=&v"(t) : "v"(acc[1][i]));
      }
      if (s2[0] + s2[1] != sr) ++nbad;
      if (q2[0] + q2[1] != qr) ++nbad;
      a[it & 7] = (_Float16)((it + lane) % 3);
    }
  } else if (wv < 4) {
    if (MODE == 0) return;
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    if (s == 12345.678f) bad[1] = 1;
  } else {
    f32x2 v[8];
    for (int k = 0; k < 8; ++k) { v[k][0] = (float)((lane * 7 + k) % 13); v[k][1] = (float)((lane * 5 + 3 * k) % 11); }
    for (int it = 0; it < iters; ++it) {
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f}, t2;
      float sr = 0.f, qr = 0.f, t;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t2) : "v"(v[k]));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q2) : "v"(t2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s2) : "v"(v[k]));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_mul_f32 %2, %3, %3\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %3" : "+v"(sr), "+v"(qr), "=&v"(t) : "v"(v[k][0]));
        asm volatile("v_mul_f32 %2, %3, %3\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %3" : "+v"(sr), "+v"(qr), "=&v"(t) : "v"(v[k][1]));
      }
      if (s2[0] + s2[1] != sr) ++nbad;
      if (q2[0] + q2[1] != qr) ++nbad;
      v[it & 7][0] = (float)((it + lane) % 13);
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}
template <int MODE>
static void run(const char* what, unsigned long long* d, int iters) {
  hipMemset(d, 0, 16);
  hipFuncSetAttribute((const void*)stress<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL(stress<MODE>, dim3(256 * 4), dim3(512), 100 * 1024, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double checks = 2.0 * 256 * 4 * (MODE == 2 ? 512 : 256) * iters;
  printf("%-70s: %llu wrong of %.3g checked sums\n", what, h[0], checks);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  run<0>("packed chain alone (waves 4-7)", d, 200000);
  run<1>("packed chain beside MFMA waves on the same SIMDs", d, 200000);
  run<2>("MFMAs, then a compiler-generated packed chain on their results, per wave", d, 100000);
  return 0;
}
