// Development tool: LDS-DMA (__builtin_amdgcn_global_load_lds, 16 bytes per lane) copies 72- and 24-byte rows of 64 robots into LDS,
// 16-byte and 8-byte aligned sources, partial EXEC on the last instruction: 0 mismatches on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// one wave copies `bytes` (multiple of 16) from src to LDS via LDS-DMA, then writes LDS out to dst
__global__ void k(const double* __restrict__ src, double* __restrict__ dst, int bytes) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const char* s = reinterpret_cast<const char*>(src);
  for (int off = 0; off < bytes; off += 1024) {
    const int o = off + lane * 16;
    if (o < bytes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + o),
                                       (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lds) + off), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < bytes / 8; i += 64) dst[i] = lds[i];
}
int main() {
  const int bytes = 4608 + 1536;  // 72-byte and 24-byte rows of 64 robots
  std::vector<double> h(bytes / 8), o(bytes / 8, -1.0);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1000.0 + i;
  double *d, *e;
  hipMalloc(&d, bytes + 64); hipMalloc(&e, bytes);
  for (int shift = 0; shift <= 8; shift += 8) {  // 16-byte aligned source, then only 8-byte aligned
    hipMemcpy((char*)d + shift, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemset(e, 0xff, bytes);
    k<<<1, 64, bytes>>>((const double*)((char*)d + shift), e, bytes);
    hipError_t rc = hipDeviceSynchronize();
    hipMemcpy(o.data(), e, bytes, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += o[i] != h[i];
    printf("shift %d: rc %d, %d mismatches of %zu\n", shift, (int)rc, bad, h.size());
  }
  return 0;
}
