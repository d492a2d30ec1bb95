// Development tool (round 3): the product kernels with a per-workgroup TIMELINE - four s_memrealtime stamps (100 MHz,
// one clock for the whole chip) per workgroup: kernel entry, inputs landed (after the fill's loads: s_waitcnt vmcnt(0)),
// solve finished (before the flush), end - plus the hardware slot (XCC / SE / CU / SIMD / wave slot) it ran on.
// tools/timeline.py turns them into "waves in each phase over time" and "bytes landed per microsecond": the picture of
// how the HBM-bound load phase and the VALU-bound solve overlap.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on tools/timeline.hip -o tools/_build/libqc_timeline.so
#include <hip/hip_runtime.h>
#define QC_TL_MAX (1 << 16)
__device__ unsigned long long qc_tl[QC_TL_MAX * 5];
__device__ unsigned long long qc_tl_w1[QC_TL_MAX];  // hardware slot of the workgroup's SECOND wave (two-wave workgroups of the paired kernel)
#define QC_TL_STAMP(k)                                                                                   \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < QC_TL_MAX) qc_tl[blockIdx.x * 5 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define QC_CLK_BEGIN()                                                                                   \
  do {                                                                                                   \
    QC_TL_STAMP(0);                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < QC_TL_MAX) {                                                    \
      unsigned hw, xcc;                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                   \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                 \
      qc_tl[blockIdx.x * 5 + 4] = ((unsigned long long)xcc << 32) | hw;                                  \
    }                                                                                                    \
    if (threadIdx.x == 64 && blockIdx.x < QC_TL_MAX) {                                                   \
      unsigned hw, xcc;                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                   \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                 \
      qc_tl_w1[blockIdx.x] = ((unsigned long long)xcc << 32) | hw;                                       \
    }                                                                                                    \
  } while (0)
#define QC_CLK(from, to)                                                                                 \
  do {                                                                                                   \
    if ((from) == 0 && (to) == 2) {                                                                      \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
      QC_TL_STAMP(1);                                                                                    \
    } else if ((from) == 7 && (to) == 8) {                                                               \
      QC_TL_STAMP(2);                                                                                    \
    }                                                                                                    \
  } while (0)
#define QC_CLK_END(last)                                                                                 \
  do {                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    QC_TL_STAMP(3);                                                                                      \
  } while (0)
#define QC_CLK_TAIL_BEGIN()
#define QC_CLK_TAIL_LOOP()
#define QC_CLK_TAIL_END()
#define QC_CLK_PIN(arr)
#include "../quadruped_control_amd/csrc/qc_balance.hip"

extern "C" int qc_timeline_read_w1(unsigned long long* out, long blocks) {
  if (blocks > QC_TL_MAX) blocks = QC_TL_MAX;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(qc_tl_w1), sizeof(unsigned long long) * blocks) == hipSuccess ? 0 : -1;
}
extern "C" int qc_timeline_read(unsigned long long* out, long blocks) {
  if (blocks > QC_TL_MAX) blocks = QC_TL_MAX;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(qc_tl), sizeof(unsigned long long) * 5 * blocks) == hipSuccess ? 0 : -1;
}
