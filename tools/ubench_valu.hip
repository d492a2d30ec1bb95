// Development tool: per-instruction issue / dependent-latency cost of the FP64 VALU, DPP and transcendental
// instructions the recalculation body is made of (one wave on one SIMD, s_memtime around unrolled streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void k(unsigned long long* out, double seed) {
  double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d = a + 1, e = a + 2, f = a + 3;
  float g = threadIdx.x, h = g + 1, g2 = g + 2, h2 = g + 3;
  unsigned long long t[24];
  int n = 0;
#define T() t[n++] = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  T();
  REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)                 // 1 dependent fma
  T();
  REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                     : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)                  // 2 independent fma x4 (64 instr)
  T();
  REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)                             // 3 dependent add
  T();
  REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)                             // 4 dependent mul
  T();
  REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)                                          // 5 dependent rsq
  T();
  REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));)  // 6 independent rsq
  T();
  REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)                                          // 7 dependent rcp
  T();
  REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(g));)  // 8 dependent dpp mov
  T();
  REP16(asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "+v"(g), "+v"(h), "+v"(g2), "+v"(h2));)                                 // 9 4 dpp movs, pairwise dependent
  T();
  // 10: the all-reduce step as the compiler emits it: 2 dpp movs of a double then add (dependent chain)
  REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "+v"(g), "+v"(g2), "+v"(h2) : "v"(h));)
  T();
  REP64(asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(d));)                             // 11 dependent min
  T();
  REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(g) : "v"(h));)                    // 12 dependent cndmask
  T();
  REP64(asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a), "v"(d) : "vcc");)                  // 13 cmp f64
  T();

  for (int i = 0; i + 1 < n; i++) if (threadIdx.x == 0) out[i] = t[i + 1] - t[i];
  if (a + d + e + f + g + h + g2 + h2 == 12345.678) out[63] = 1;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  k<<<1, 64>>>(d, 1.0); k<<<1, 64>>>(d, 1.0); hipDeviceSynchronize();
  unsigned long long h[64]; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"dep fma_f64", "indep fma_f64 (x4)", "dep add_f64", "dep mul_f64", "dep rsq_f64", "indep rsq_f64", "dep rcp_f64", "dep dpp mov",
                      "dpp mov pairs", "nop+2 dpp mov", "dep min_f64", "dep cndmask", "cmp_lt_f64"};
  for (int i = 0; i < 13; i++) printf("%-22s %6.1f cycles/instr (64 instr: %llu)\n", nm[i], h[i] / 64.0, h[i]);
  return 0;
}
