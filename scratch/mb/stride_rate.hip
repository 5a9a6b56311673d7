// L2 -> LDS staging rate of GEMM-shaped tiles: a block walks the K-tiles of a 128-row panel of a row-major matrix, each
// DMA instruction fetching 8 rows x 128 B (the fill shape of igemm_kernel / conv_win_kernel), as a function of the ROW STRIDE.
// Two K-tiles in flight per block (ring of 2), BLOCKS_PER_CU blocks per CU by LDS size.  argv: blocks panels lds_per_block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

__global__ __launch_bounds__(256) void panel_kernel(const char* __restrict__ src, long long stride, int ktiles_row, int ntiles, int* sink, int npanels) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const char* base = src + (long long)(blockIdx.x % npanels) * 128 * stride;
  // piece j of wave wv: rows 32 j + 8 wv + lane / 8, 16-byte chunk lane % 8 of the 128-byte K-tile segment
  const char* p[4];
  for (int j = 0; j < 4; ++j) p[j] = base + (long long)(32 * j + 8 * wv + (lane >> 3)) * stride + (lane & 7) * 16;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) GLDS(p[j] + (i % ktiles_row) * 128, smem + i * 16384 + (j * 4 + wv) * 1024);
  int acc = 0;
  for (int k = 0; k < ntiles; ++k) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int slot = k & 1;
    acc += *(const int*)(smem + slot * 16384 + t * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int nk = (k + 2) % ktiles_row;
    for (int j = 0; j < 4; ++j) GLDS(p[j] + nk * 128, smem + slot * 16384 + (j * 4 + wv) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 768;
  const int npanels = argc > 2 ? atoi(argv[2]) : 144;
  const int lds = argc > 3 ? atoi(argv[3]) : 49152;   // 48 KB: three blocks per CU, as the 128 x 64 GEMM tile
  const int ntiles = 40;
  const long long strides[] = {640, 768, 1280, 1408, 2560, 2688, 5120, 5248, 1920, 3840, 10240, 10368};
  char* src; int* sink;
  const size_t bytes_alloc = (size_t)npanels * 128 * 10368 + (1 << 20);
  hipMalloc(&src, bytes_alloc); hipMalloc(&sink, 64);
  hipMemset(src, 1, bytes_alloc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (long long s : strides) {
    const int kr = (int)((s / 128) < 5 ? (s / 128) : (s / 128 / 128 * 128 > 0 ? s / 128 : s / 128));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(panel_kernel, dim3(blocks), dim3(256), lds, 0, src, s, kr, ntiles, sink, npanels);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(panel_kernel, dim3(blocks), dim3(256), lds, 0, src, s, kr, ntiles, sink, npanels);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0, bytes = (double)blocks * ntiles * 16384;
    printf("row stride %6lld B (%3d K-tiles per row, footprint %5.1f MB) blocks %4d lds %3d KB: %7.1f us per launch, %6.2f TB/s in all, %.3f us per 16 KB tile and block\n",
           s, kr, (double)npanels * 128 * s / 1e6, blocks, lds / 1024, us, bytes / us / 1e6, us / ntiles);
  }
  return 0;
}
