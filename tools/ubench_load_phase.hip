// Development tool (round 3): the batch-load phase of a one-fill wave in isolation - 64 robots per 64-lane
// workgroup, nine input arrays (388 B per robot) in, one 96-byte row + status out - with the product's
// row-per-lane gathers against LDS-DMA staging (global_load_lds_dwordx4: coalesced 1 KB per instruction, no VGPRs)
// of the three big arrays (Rwb, Rwb_d, feet: 240 B per robot = 15 KB per fill) or of all nine (25 KB per fill).
// Occupancy is pinned through the dynamic LDS size (8 or 6 workgroups per CU), as registers pin it in the product.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_load_phase.hip -o tools/_build/ubench_load_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// copy `bytes` (multiple of 16 except a tail that is dropped - callers pass multiples of 8 and we round the count up
// to 16 when the row span allows) from global to LDS with LDS-DMA; all 64 lanes call it
__device__ __forceinline__ void dma_span(const char* __restrict__ src, char* lds_dst, int bytes, int lane) {
  for (int off = 0; off < bytes; off += 1024) {
    const int o = off + lane * 16;
    if (o < bytes) __builtin_amdgcn_global_load_lds((gptr_t)(src + o), (lptr_t)(lds_dst + off), 16, 0, 0);
  }
}
template <int ROW>
__device__ __forceinline__ void row_global(const double* __restrict__ a, long r, double (&v)[ROW]) {
  const double* q = a + r * ROW;
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = q[k];
}
template <int ROW>
__device__ __forceinline__ void row_lds(const double* img, int lane, double (&v)[ROW]) {
  const double* q = img + lane * ROW;
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = q[k];
}

struct In {
  const double *A9, *B9, *V[6], *F12;
  const unsigned* st;
};

// MODE 0: gathers only.  1: DMA of A9, B9, F12 + gathers of the six 3-vectors.  2: DMA of everything.
// 3: as 1, but the gathers of the 3-vectors are issued BEFORE the wait on the DMA (both in flight together).
template <int MODE>
__global__ __launch_bounds__(64) void load_phase(In in, double* __restrict__ out, int* __restrict__ status, long n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  const long r0 = (long)blockIdx.x * 64;
  const long r = r0 + lane;
  const int k = n - r0 < 64 ? (int)(n - r0) : 64;  // robots of this fill
  double a[9], b[9], v[6][3], f[12];
  if (MODE == 0) {
    if (lane < k) {
      row_global<9>(in.A9, r, a); row_global<9>(in.B9, r, b);
#pragma unroll
      for (int j = 0; j < 6; j++) row_global<3>(in.V[j], r, v[j]);
      row_global<12>(in.F12, r, f);
    }
  } else {
    char* img = lds;
    dma_span((const char*)(in.A9 + r0 * 9), img, k * 72, lane);
    dma_span((const char*)(in.B9 + r0 * 9), img + 4608, k * 72, lane);
    dma_span((const char*)(in.F12 + r0 * 12), img + 9216, k * 96, lane);
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 6; j++) dma_span((const char*)(in.V[j] + r0 * 3), img + 15360 + 1536 * j, k * 24, lane);
    }
    if (MODE == 3 && lane < k) {
#pragma unroll
      for (int j = 0; j < 6; j++) row_global<3>(in.V[j], r, v[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (single-wave workgroup: a no-op fence for the compiler's LDS ordering)
    if (lane < k) {
      row_lds<9>((const double*)img, lane, a);
      row_lds<9>((const double*)(img + 4608), lane, b);
      row_lds<12>((const double*)(img + 9216), lane, f);
      if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 6; j++) row_lds<3>((const double*)(img + 15360 + 1536 * j), lane, v[j]);
      } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 6; j++) row_global<3>(in.V[j], r, v[j]);
      }
    }
  }
  if (lane < k) {
    const unsigned sw = in.st[r];
    double s[12];
#pragma unroll
    for (int q = 0; q < 12; q++)
      s[q] = f[q] + a[q % 9] * b[(q + 1) % 9] + v[0][q % 3] + v[1][q % 3] * v[2][q % 3] + v[3][q % 3] + v[4][q % 3] + v[5][q % 3] + (double)(sw & 1u);
    double* o = out + r * 12;
#pragma unroll
    for (int q = 0; q < 12; q++) o[q] = s[q];
    status[r] = (int)(sw >> 8);
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2097152;
  const int rows[9] = {9, 9, 3, 3, 3, 3, 3, 3, 12};
  std::vector<double*> d(9);
  for (int i = 0; i < 9; i++) {
    hipMalloc(&d[i], n * rows[i] * 8 + 64);
    std::vector<double> h(n * rows[i]);
    for (size_t j = 0; j < h.size(); j++) h[j] = (double)((j * 2654435761u + i) % 1000) * 1e-3;
    hipMemcpy(d[i], h.data(), h.size() * 8, hipMemcpyHostToDevice);
  }
  unsigned* st; hipMalloc(&st, n * 4); hipMemset(st, 1, n * 4);
  double *out, *ref; int* status;
  hipMalloc(&out, n * 96); hipMalloc(&ref, n * 96); hipMalloc(&status, n * 4);
  In in{d[0], d[1], {d[2], d[3], d[4], d[5], d[6], d[7]}, d[8], st};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* nm[4] = {"row-per-lane gathers (product)", "LDS-DMA of Rwb/Rwb_d/feet + gathers", "LDS-DMA of all nine arrays", "as 1, gathers issued before the DMA wait"};
  const unsigned blocks = (unsigned)((n + 63) / 64);
  std::vector<double> h0(n * 12), h1(n * 12);
  for (int occ = 0; occ < 2; occ++) {  // LDS per workgroup: 20 KB (8 per CU) / 26 KB (6 per CU, what the full image needs)
    const size_t ldsb = occ == 0 ? 20 * 1024 : 26 * 1024;
    for (int mode = 0; mode < 4; mode++) {
      if (mode == 2 && occ == 0) continue;  // the full image does not fit 20 KB
      double* o = mode == 0 ? ref : out;
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        const int reps = 5;
        for (int q = 0; q < reps; q++) {
          if (mode == 0) load_phase<0><<<blocks, 64, ldsb>>>(in, o, status, n);
          else if (mode == 1) load_phase<1><<<blocks, 64, ldsb>>>(in, o, status, n);
          else if (mode == 2) load_phase<2><<<blocks, 64, ldsb>>>(in, o, status, n);
          else load_phase<3><<<blocks, 64, ldsb>>>(in, o, status, n);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms / reps < best) best = ms / reps;
      }
      long bad = 0;
      if (mode != 0) {
        hipMemcpy(h0.data(), ref, n * 96, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), out, n * 96, hipMemcpyDeviceToHost);
        for (long j = 0; j < n * 12; j++) bad += h0[j] != h1[j];
      }
      printf("n=%ld lds=%zuK %-42s %8.1f us  %6.0f GB/s (488 B/robot)  mismatches %ld  [%s]\n", n, ldsb / 1024, nm[mode], best * 1e3,
             488.0 * n / (best * 1e-3) / 1e9, bad, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
