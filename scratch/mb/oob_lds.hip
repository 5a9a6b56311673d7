// What does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) leave in the
// LDS: zeros, or the old bytes?  The round-3 loaders mask rows (vertical conv padding, rows past M) by giving the lane a
// voffset >= num_records, so the answer must be "zeros".  Also checks that soffset takes no part in the range check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const unsigned* src, unsigned* out, unsigned num_records, int soff) {
  __shared__ __attribute__((aligned(16))) unsigned smem[1024];
  const int t = threadIdx.x;
  for (int i = t; i < 1024; i += 256) smem[i] = 0xAAAA0000u + i;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, num_records, 0x00020000);
  // lanes with t % 4 == 3 are out of range
  unsigned voff = (t % 4 == 3) ? 0x80000000u : (unsigned)t * 16;
  unsigned ldsaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem + __builtin_amdgcn_readfirstlane(t >> 6) * 1024;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsaddr), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 1024; i += 256) out[i] = smem[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0x55000000u + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r(1024);
  for (int pass = 0; pass < 2; ++pass) {
    const int soff = pass ? 4096 : 0;     // pass 1: soffset moves every lane by 4 KB
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, 0x80000000u, soff);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    int zeros = 0, old = 0, other = 0, good = 0, bad = 0;
    for (int t = 0; t < 256; ++t)
      for (int e = 0; e < 4; ++e) {
        const unsigned v = r[t * 4 + e];
        if (t % 4 == 3) { if (v == 0) ++zeros; else if (v == 0xAAAA0000u + t * 4 + e) ++old; else ++other; }
        else { if (v == 0x55000000u + soff / 4 + t * 4 + e) ++good; else ++bad; }
      }
    printf("soffset %d: in-range dwords good %d bad %d | out-of-range dwords: zeros %d, old LDS bytes %d, other %d\n", soff, good, bad, zeros, old, other);
  }
  return 0;
}
