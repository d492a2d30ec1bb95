// Development tool: the product kernel with phase-clock hooks switched on.  Builds an alternative library
// (tools/_build/libqc_balance_clk.so) whose block 0 attributes s_memtime cycles to the phases of the kernel:
//   0 first restock (load + assembly)   1 loop bookkeeping / refill / result push   2 face coefficients + local M
//   3 group reduction (DPP)             4 6x6 Cholesky                              5 triangular solves
//   6 forces + gradient                 7 ratio test, multipliers, state update    8 final flush (output transform + store)
//   9 empty (marker cost, counted twice per iterate)
//   10 iterate calls                    11 s_memrealtime ticks (100 MHz) start -> end, 12 s_memtime ticks start -> end
//   13 / 14 / 15: the torque pass of the widened tick (task lists, swing-leg tasks, stance-leg tasks; 8 keeps the GRF stores)
//   16 + k: phase k of the recalculations of the 4-lane tail (one / two lanes-per-robot kernels), 26 their count; 1 = the re-pack
// Each marker costs one s_memtime + s_waitcnt (~50-100 cycles): compare phases, do not read totals as product time.
#include <hip/hip_runtime.h>
__device__ unsigned long long qc_clk_global[32];
__shared__ unsigned long long qc_clk_lds[32];
__shared__ int qc_clk_off;  // 0, or 16 while the 4-lane tail runs (slots 16 + phase)
#define QC_CLK_ELECT() (threadIdx.x < 64 && __lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))  // wave 0 of the workgroup
#define QC_CLK(from, to)                                                               \
  do {                                                                                 \
    const unsigned long long t_ = __builtin_readcyclecounter();                        \
    if (QC_CLK_ELECT()) {                                                              \
      atomicAdd(&qc_clk_lds[(from) + qc_clk_off], t_);                                 \
      atomicAdd(&qc_clk_lds[(to) + qc_clk_off], 0ull - t_);                            \
      if ((to) == 2) atomicAdd(&qc_clk_lds[10 + qc_clk_off], 1ull);                    \
    }                                                                                  \
  } while (0)
#define QC_CLK_X(from, to)  /* absolute slots */                                       \
  do {                                                                                 \
    const unsigned long long t_ = __builtin_readcyclecounter();                        \
    if (QC_CLK_ELECT()) {                                                              \
      atomicAdd(&qc_clk_lds[from], t_);                                                \
      atomicAdd(&qc_clk_lds[to], 0ull - t_);                                           \
    }                                                                                  \
  } while (0)
#define QC_CLK_ABS(from, to) QC_CLK_X(from, to)  /* torque pass of the widened tick: 13 task lists, 14 swing-leg tasks, 15 stance-leg tasks */
#define QC_CLK_TAIL_BEGIN() do { QC_CLK_X(7, 1); } while (0)                 /* slot 1: the re-pack */
#define QC_CLK_TAIL_LOOP() do { QC_CLK_X(1, 23); qc_clk_off = 16; } while (0)
#define QC_CLK_TAIL_END() do { qc_clk_off = 0; QC_CLK_X(23, 7); } while (0)
#define QC_CLK_BEGIN()                                                                 \
  if (threadIdx.x < 32) qc_clk_lds[threadIdx.x] = 0;                                   \
  if (threadIdx.x == 0) qc_clk_off = 0;                                                \
  __syncthreads();                                                                     \
  if (threadIdx.x == 0) {                                                              \
    qc_clk_lds[0] = 0ull - __builtin_readcyclecounter();                               \
    qc_clk_lds[12] = qc_clk_lds[0];                                                    \
    qc_clk_lds[11] = 0ull - __builtin_amdgcn_s_memrealtime();                          \
  }                                                                                    \
  __syncthreads()
#define QC_CLK_END(last)                                                               \
  do {                                                                                 \
    __syncthreads();                                                                   \
    if (threadIdx.x == 0) {                                                            \
      const unsigned long long t_ = __builtin_readcyclecounter();                      \
      qc_clk_lds[last] += t_;                                                          \
      qc_clk_lds[12] += t_;                                                            \
      qc_clk_lds[11] += __builtin_amdgcn_s_memrealtime();                              \
    }                                                                                  \
    __syncthreads();                                                                   \
    if (blockIdx.x == QC_CLK_BLOCK && threadIdx.x < 32) atomicAdd(&qc_clk_global[threadIdx.x], qc_clk_lds[threadIdx.x]); \
  } while (0)
#define QC_CLK_PIN(arr)                                                               \
  do {                                                                                 \
    _Pragma("unroll") for (unsigned i_ = 0; i_ < sizeof(arr) / sizeof(arr[0]); i_++) asm volatile("" : "+v"(arr[i_])); \
  } while (0)
#ifndef QC_CLK_BLOCK
#define QC_CLK_BLOCK 0
#endif
#include "../quadruped_control_amd/csrc/qc_balance.hip"

extern "C" int qc_clk_read(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(qc_clk_global), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(qc_clk_global), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
