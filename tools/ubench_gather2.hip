// Development tool: which side of the row-per-lane pattern costs bandwidth - the gathers (9 input arrays, 384 B per
// record) or the 96-byte row scatter of the results?  Variants: full (gather + scatter), gather only (+ one coalesced
// double per record), scatter only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ROW>
__device__ __forceinline__ void rows_direct(const double* __restrict__ a, long r, double (&v)[ROW]) {
  const double* q = a + r * ROW;
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = q[k];
}
template <int MODE>  // 0 full, 1 gather only, 2 scatter only, 3 scatter through LDS (coalesced 16-byte pieces)
__global__ __launch_bounds__(64) void k(const double* __restrict__ A9, const double* __restrict__ B9, const double* __restrict__ V0, const double* __restrict__ V1,
                                       const double* __restrict__ V2, const double* __restrict__ V3, const double* __restrict__ V4,
                                       const double* __restrict__ V5, const double* __restrict__ F12, double* __restrict__ out, double* __restrict__ out1, long n, long chunk) {
  __shared__ double stage[64 * 12];
  const int lane = threadIdx.x;
  long r0 = (long)blockIdx.x * chunk;
  const long end = r0 + chunk < n ? r0 + chunk : n;
  for (; r0 < end; r0 += 64) {
    const long r = r0 + lane;
    double s[12];
    if (MODE != 2 && MODE != 3) {
      double a[9], b[9], v0[3], v1[3], v2[3], v3[3], v4[3], v5[3], f[12];
      rows_direct<9>(A9, r, a); rows_direct<9>(B9, r, b);
      rows_direct<3>(V0, r, v0); rows_direct<3>(V1, r, v1); rows_direct<3>(V2, r, v2);
      rows_direct<3>(V3, r, v3); rows_direct<3>(V4, r, v4); rows_direct<3>(V5, r, v5);
      rows_direct<12>(F12, r, f);
#pragma unroll
      for (int q = 0; q < 12; q++) s[q] = f[q] + a[q % 9] * b[(q + 1) % 9] + v0[q % 3] + v1[q % 3] * v2[q % 3] + v3[q % 3] + v4[q % 3] + v5[q % 3];
    } else {
#pragma unroll
      for (int q = 0; q < 12; q++) s[q] = (double)(r + q);
    }
    if (MODE == 1) {
      double t = 0;
#pragma unroll
      for (int q = 0; q < 12; q++) t += s[q];
      out1[r] = t;
    } else if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 12; q++) stage[lane * 12 + q] = s[q];
      __syncthreads();
      double2* o = reinterpret_cast<double2*>(out + r0 * 12);
      const double2* st = reinterpret_cast<const double2*>(stage);
#pragma unroll
      for (int j = 0; j < 6; j++) o[lane + 64 * j] = st[lane + 64 * j];
      __syncthreads();
    } else {
      double* o = out + r * 12;
#pragma unroll
      for (int q = 0; q < 12; q++) o[q] = s[q];
    }
  }
}
int main() {
  const long n = 2097152, chunk = 1024;
  std::vector<double*> d(11);
  const int rows[11] = {9, 9, 3, 3, 3, 3, 3, 3, 12, 12, 1};
  for (int i = 0; i < 11; i++) { hipMalloc(&d[i], n * rows[i] * 8); hipMemset(d[i], 0, n * rows[i] * 8); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* nm[4] = {"gather 384 B + scatter 96 B", "gather 384 B (+8 B coalesced)", "scatter 96 B rows", "scatter 96 B via LDS, coalesced"};
  const double bytes[4] = {480.0, 392.0, 96.0, 96.0};
  for (int mode = 0; mode < 4; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      const int reps = 5;
      for (int r = 0; r < reps; r++) {
#define L(M) k<M><<<(n + chunk - 1) / chunk, 64>>>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], n, chunk)
        if (mode == 0) L(0); else if (mode == 1) L(1); else if (mode == 2) L(2); else L(3);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-34s %.1f us per pass, %.0f GB/s\n", nm[mode], ms / reps * 1e3, bytes[mode] * n / (ms / reps * 1e-3) / 1e9);
    }
  return 0;
}
