// Development tool (round 3): what does a wave pay, at the start of a kernel, for its kernel-argument words and for a
// dependent scalar load from a constant buffer?  Each stage is an s_load of words not touched before + s_waitcnt, stamped
// with s_memrealtime (100 MHz).  2048 single-wave workgroups = one per wave slot of the product's launches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_kernarg.hip -o tools/_build/ubench_kernarg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { const long* p[26]; };  // 208 bytes of kernel arguments, like the product's
__global__ __launch_bounds__(64) void k(Args a, const long* cbuf, unsigned long long* out) {
  unsigned long long t0, t1, t2, t3, t4, t5;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  long v0 = (long)a.p[0];                 // first 64-byte line of the argument segment
  asm volatile("s_nop 0" : "+s"(v0));
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
  long v1 = (long)a.p[9];                 // second line
  asm volatile("s_nop 0" : "+s"(v1));
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2));
  long v2 = (long)a.p[17] + (long)a.p[25]; // third and fourth line, one wait
  asm volatile("s_nop 0" : "+s"(v2));
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3));
  typedef const __attribute__((address_space(4))) long clong;
  clong* cb = (clong*)(unsigned long long)cbuf;
  long v3 = cb[0];                      // dependent scalar load from a device buffer (the constants)
  asm volatile("s_nop 0" : "+s"(v3));
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t4));
  long v4 = cb[64];                     // another line of it
  asm volatile("s_nop 0" : "+s"(v4));
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t5));
  if (threadIdx.x == 0) {
    unsigned long long* o = out + blockIdx.x * 8;
    o[0] = t0; o[1] = t1 - t0; o[2] = t2 - t1; o[3] = t3 - t2; o[4] = t4 - t3; o[5] = t5 - t4; o[6] = (unsigned long long)(v0 + v1 + v2 + v3 + v4);
  }
}
int main() {
  const int blocks = 2048;
  long* cb; unsigned long long* out;
  hipMalloc(&cb, 4096); hipMemset(cb, 0, 4096);
  hipMalloc(&out, blocks * 64);
  Args a; for (int i = 0; i < 26; i++) a.p[i] = cb;
  std::vector<unsigned long long> h(blocks * 8);
  for (int rep = 0; rep < 3; rep++) {
    k<<<blocks, 64>>>(a, cb, out);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, blocks * 64, hipMemcpyDeviceToHost);
    double s[6] = {0}; unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < blocks; b++) { for (int j = 1; j < 6; j++) s[j] += h[b * 8 + j]; if (h[b*8] < tmin) tmin = h[b*8]; if (h[b*8] > tmax) tmax = h[b*8]; }
    printf("launch %d: mean ns per stage: kernarg line 0 %.0f | line 1 %.0f | lines 2+3 %.0f | constant buffer first word %.0f | second line %.0f   (entry stamps spread over %.2f us)\n",
           rep, s[1] / blocks * 10, s[2] / blocks * 10, s[3] / blocks * 10, s[4] / blocks * 10, s[5] / blocks * 10, (tmax - tmin) / 100.0);
  }
  return 0;
}
