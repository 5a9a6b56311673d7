// Per-CU rate of moving 16 KB tiles from an L2/MALL-resident buffer into the LDS: LDS-DMA (global_load_lds_dwordx4) against
// global_load_dwordx4 -> ds_write_b128, one 256-thread block per CU, DEPTH tiles in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

template <int DEPTH>
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, long long tile_stride, int ntiles, int* sink, int nstreams) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, wv = t >> 6;
  const char* base = src + (long long)(blockIdx.x % nstreams) * tile_stride * ntiles;
  // a 16 KB tile = 4 DMA instructions per thread (each wave-instruction moves 1 KB)
  for (int i = 0; i < DEPTH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) GLDS(base + (long long)i * tile_stride + (j * 4 + wv) * 1024 + (t & 63) * 16, smem + i * 16384 + (j * 4 + wv) * 1024);
  int acc = 0;
  for (int k = 0; k < ntiles; ++k) {
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int slot = k % DEPTH;
    acc += *(const int*)(smem + slot * 16384 + t * 16);  // touch the tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int nk = k + DEPTH < ntiles ? k + DEPTH : k;   // keep the DMA count constant: re-read a live tile past the end
#pragma unroll
    for (int j = 0; j < 4; ++j) GLDS(base + (long long)nk * tile_stride + (j * 4 + wv) * 1024 + (t & 63) * 16, smem + slot * 16384 + (j * 4 + wv) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678) sink[0] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void reg_kernel(const char* __restrict__ src, long long tile_stride, int ntiles, int* sink, int nstreams) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const char* base = src + (long long)(blockIdx.x % nstreams) * tile_stride * ntiles;
  u32x4 r[DEPTH][4];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r[i][j] = *(const u32x4*)(base + (long long)i * tile_stride + j * 4096 + t * 16);
  int acc = 0;
#pragma unroll 1
  for (int k0 = 0; k0 < ntiles; k0 += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const int k = k0 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(u32x4*)(smem + (i & 1) * 16384 + j * 4096 + t * 16) = r[i][j];
      const int nk = k + DEPTH < ntiles ? k + DEPTH : k;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[i][j] = *(const u32x4*)(base + (long long)nk * tile_stride + j * 4096 + t * 16);
      __syncthreads();
      acc += *(const int*)(smem + (i & 1) * 16384 + ((t * 16 + 4096) & 16383));
      __syncthreads();
    }
  }
  if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const int nstreams = argc > 2 ? atoi(argv[2]) : blocks;   // distinct 1 MB streams the blocks read (footprint = nstreams MB)
  const int ntiles = 64;                       // 1 MB per block
  const long long tile = 16384;
  char* src; int* sink;
  hipMalloc(&src, (size_t)blocks * ntiles * tile); hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)blocks * ntiles * tile);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int lds) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, tile, ntiles, sink, nstreams);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, tile, ntiles, sink, nstreams);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0, bytes = (double)blocks * ntiles * tile;
    printf("%-28s blocks %4d streams %3d: %7.1f us per launch, %6.1f GB/s per block, %6.2f TB/s in all, %.3f us per 16 KB tile\n", name, blocks, nstreams, us,
           bytes / blocks / us / 1e3, bytes / us / 1e6, us / ntiles);
  };
  run("LDS-DMA depth 1", dma_kernel<1>, 16384);
  run("LDS-DMA depth 2", dma_kernel<2>, 32768);
  run("LDS-DMA depth 4", dma_kernel<4>, 65536);
  run("LDS-DMA depth 8", dma_kernel<8>, 131072);
  run("registers depth 1", reg_kernel<1>, 32768);
  run("registers depth 2", reg_kernel<2>, 32768);
  run("registers depth 4", reg_kernel<4>, 32768);
  run("registers depth 8", reg_kernel<8>, 32768);
  return 0;
}
