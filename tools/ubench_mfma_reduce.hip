// Development tool: can v_mfma_f64_4x4x4f64 with an all-ones B serve as a 4-lane all-reduce?  Prints which lanes are
// summed into which, and the cycles for 27 such reductions vs the DPP version (4 moves + 2 adds per double).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int dpp_i(int v, int ctrl) { return ctrl == 0xB1 ? __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ double dpp_d(double v, int ctrl) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)dpp_i((int)(unsigned)b, ctrl), hi = (unsigned)dpp_i((int)(unsigned)(b >> 32), ctrl);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ void k(double* out, unsigned long long* cyc) {
  const int lane = threadIdx.x;
  // mapping probe: A = 2^lane-ish distinct values so that sums identify their terms
  double a = (double)(1ull << (lane % 16)) * (1.0 + 65536.0 * (lane / 16));
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, 0.0, 0, 0, 0);
  out[lane] = d;
  double v[27], w[27];
#pragma unroll
  for (int i = 0; i < 27; i++) v[i] = lane * 0.5 + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 27; i++) w[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[i], 1.0, 0.0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 27; i++) asm volatile("" : "+v"(w[i]));
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double u[27];
#pragma unroll
  for (int i = 0; i < 27; i++) { double x = v[i]; x += dpp_d(x, 0xB1); x += dpp_d(x, 0x4E); u[i] = x; }
#pragma unroll
  for (int i = 0; i < 27; i++) asm volatile("" : "+v"(u[i]));
  unsigned long long t2 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double acc = 0;
#pragma unroll
  for (int i = 0; i < 27; i++) acc += w[i] + u[i];
  out[64 + lane] = acc;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 128 * 8); hipMalloc(&c, 16);
  k<<<1, 64>>>(d, c); k<<<1, 64>>>(d, c); hipDeviceSynchronize();
  double h[128]; unsigned long long hc[2];
  hipMemcpy(h, d, 128 * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    // decode: value = sum over source lanes s of 2^(s%16) * (1 + 65536*(s/16))
    const double v = h[l];
    printf("lane %2d <- ", l);
    for (int blk = 0; blk < 4; blk++) {
      // contributions of block blk have weight (1 + 65536*blk); blocks do not mix in this instruction, so test each
      const double wgt = 1.0 + 65536.0 * blk;
      const double q = v / wgt;
      if (q == (double)(long long)q && q < 65536.0 && q > 0) { for (int b = 0; b < 16; b++) if (((long long)q >> b) & 1) printf("%d ", 16 * blk + b); }
    }
    printf("\n");
    if (l == 19) { l = 59; }
  }
  printf("27 reductions: mfma %llu cycles, dpp %llu cycles\n", hc[0], hc[1]);
  return 0;
}
