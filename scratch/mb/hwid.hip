// Which SIMD does wave w of a 512-thread workgroup run on?  (the two-group attention kernel pairs waves that share a SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void hwid_kernel(unsigned* out) {
  extern __shared__ char smem[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
  unsigned* d; const int nb = 6;
  hipMalloc(&d, nb * 8 * 4);
  for (int lds : {0, 128 * 1024}) {
    hipFuncSetAttribute((const void*)hwid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(hwid_kernel, dim3(nb), dim3(512), lds, 0, d);
    unsigned h[nb * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
      printf("lds %6d block %d:", lds, b);
      for (int w = 0; w < 8; ++w) printf("  w%d simd %u wave %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
      printf("\n");
    }
  }
  return 0;
}
