// Which VALU instruction kinds overlap with another wave's MFMAs on the same SIMD?  (gfx950)
// One block of 8 waves per CU (waves w and w + 4 share a SIMD: hwid.hip): waves 0-3 run 8 MFMAs per iteration, waves 4-7 run
// NV vector instructions of kind VT.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ROLE, int VT, int NV, int PRIO>  // PRIO: the MFMA wave raises its priority; ROLE 0: MFMA blocks only, 1: vector blocks only, 2: both kinds of blocks
__global__ __launch_bounds__(512) void mb(float* out, int iters) {
  extern __shared__ char smem_[];
  const int lane = threadIdx.x & 255;
  hx8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.001f + j); b[j] = (_Float16)(lane * 0.002f - j); }
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = lane * 0.0005f + j * 0.01f;
  const float c0 = 1.0001f, c1 = 0.0001f;
  const bool mrole = (threadIdx.x >> 8) == 0;
  if (ROLE == 0 && !mrole) return;
  if (ROLE == 1 && mrole) return;
#ifdef ACC_AGPR
#define MF(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[(J) & 1]) : "v"(a), "v"(b))
#else
#define MF(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(J) & 1]) : "v"(a), "v"(b))
#endif
  if (PRIO && mrole) asm volatile("s_setprio 3");
  for (int it = 0; it < iters; ++it) {
    if (mrole) {
#pragma unroll
      for (int j = 0; j < 8; ++j) MF(j);
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float& x = v[k & 7];
        if (VT == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        if (VT == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        if (VT == 2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        if (VT == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        if (VT == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        if (VT == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(c1));
        if (VT == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[(k & 3) * 2]) : "v"(*(const double*)&v[0]));
        if (VT == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        if (VT == 8) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(c1));
        if (VT == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[(k & 3) * 2]) : "v"(*(const double*)&v[0]));
        if (VT == 10) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(c0), "v"(c1));
        if (VT == 11) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
        if (VT == 12) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        if (VT == 13) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        if (VT == 14) asm volatile("v_sub_f32 %0, %0, %1\n\tv_exp_f32 %0, %0" : "+v"(x) : "v"(c1));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[0] = s;
}
template <int ROLE, int VT, int NV, int PRIO>
static float run(float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)mb<ROLE, VT, NV, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((mb<ROLE, VT, NV, PRIO>), dim3(256), dim3(512), 100 * 1024, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mb<ROLE, VT, NV, PRIO>), dim3(256), dim3(512), 100 * 1024, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
template <int VT, int NV>
static void kind(const char* name, float* out) {
  const float m = run<0, VT, NV, 0>(out), v = run<1, VT, NV, 0>(out), both = run<2, VT, NV, 0>(out), bothp = run<2, VT, NV, 1>(out);
  printf("%-22s x%3d: MFMA wave alone %6.1f ns | vector wave alone %6.1f ns (%.1f cycles each @2.4) | side by side %6.1f ns (overlap %.0f %%) | MFMA wave at priority 3: %6.1f ns (overlap %.0f %%)\n", name, NV, m, v,
         v * 2.4 / NV, both, 100.0 * (m + v - both) / (m < v ? m : v), bothp, 100.0 * (m + v - bothp) / (m < v ? m : v));
}
int main() {
  float* out; hipMalloc(&out, 4096);
  kind<0, 32>("v_fma_f32", out);
  kind<1, 16>("v_exp_f32", out);
  kind<1, 32>("v_exp_f32", out);
  kind<2, 32>("v_dot2c_f32_f16", out);
  kind<3, 32>("v_cvt_pk_f16_f32", out);
  kind<4, 32>("v_max3_f32", out);
  kind<5, 32>("v_mov_b32", out);
  kind<6, 32>("v_pk_mul_f32", out);
  kind<7, 32>("v_add_f32", out);
  kind<8, 32>("v_pk_add_f16", out);
  kind<9, 32>("v_pk_add_f32", out);
  kind<10, 32>("v_dot2_f32_f16", out);
  kind<11, 32>("v_exp_f16", out);
  kind<12, 32>("v_cvt_pkrtz_f16_f32", out);
  kind<13, 32>("v_fmac_f32", out);
  kind<0, 64>("v_fma_f32", out);
  kind<7, 64>("v_add_f32", out);
  kind<14, 32>("v_sub + v_exp", out);
  return 0;
}
