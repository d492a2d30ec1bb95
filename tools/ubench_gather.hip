// Development tool: how fast can 64 lanes fetch one record each from row-major [n][ROW] double arrays?
//   A  row per lane straight from global memory (what the dense assembly does: 16-byte pieces at a 24/72/96-byte stride)
//   B  coalesced 16-byte pieces -> LDS (row-major image of the 64 records) -> each lane reads its row from LDS
// Same bytes, same number of load instructions; prints GB/s of input bytes for both (2 waves/SIMD via launch bounds + LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ROW>
__device__ __forceinline__ void rows_direct(const double* __restrict__ a, long r0, int lane, double (&v)[ROW]) {
  const double* q = a + (r0 + lane) * ROW;
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = q[k];
}
// 64 records of ROW doubles = 32*ROW 16-byte pieces; piece e = lane + 64 j
template <int ROW>
__device__ __forceinline__ void rows_via_lds(const double* __restrict__ a, long r0, int lane, double* __restrict__ stage, double (&v)[ROW]) {
  constexpr int PIECES = 32 * ROW, PER = (PIECES + 63) / 64;
  const double2* src = reinterpret_cast<const double2*>(a + r0 * ROW);
  double2 t[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int e = lane + 64 * j;
    t[j] = src[e < PIECES ? e : PIECES - 1];
  }
  double2* st = reinterpret_cast<double2*>(stage);
#pragma unroll
  for (int j = 0; j < PER; j++)
    if (lane + 64 * j < PIECES) st[lane + 64 * j] = t[j];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = stage[lane * ROW + k];
  __syncthreads();
}

// B v2: all coalesced loads of a 64-record block are issued first (same registers the rows end up in), then three
// LDS passes (A9+B9 | six V3 | F12) turn pieces into rows: 6 wave barriers per block instead of 18, no global wait
// between arrays.
template <int ROW>
__device__ __forceinline__ void pieces_load(const double* __restrict__ a, long r0, int lane, double2 (&t)[(32 * ROW + 63) / 64]) {
  constexpr int PIECES = 32 * ROW, PER = (PIECES + 63) / 64;
  const double2* src = reinterpret_cast<const double2*>(a + r0 * ROW);
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int e = lane + 64 * j;
    t[j] = src[e < PIECES ? e : PIECES - 1];  // unconditional (clamped) so the pieces stay in registers
  }
}
template <int ROW>
__device__ __forceinline__ void pieces_to_lds(double* __restrict__ stage, int lane, const double2 (&t)[(32 * ROW + 63) / 64]) {
  constexpr int PIECES = 32 * ROW, PER = (PIECES + 63) / 64;
  double2* st = reinterpret_cast<double2*>(stage);
#pragma unroll
  for (int j = 0; j < PER; j++)
    if (lane + 64 * j < PIECES) st[lane + 64 * j] = t[j];
}
template <int ROW>
__device__ __forceinline__ void rows_from_lds(const double* __restrict__ stage, int lane, double (&v)[ROW]) {
#pragma unroll
  for (int k = 0; k < ROW; k++) v[k] = stage[lane * ROW + k];
}
template <int VIA_LDS>
__global__ __launch_bounds__(64, 1) void k(const double* __restrict__ A9, const double* __restrict__ B9, const double* __restrict__ V0, const double* __restrict__ V1,
                                          const double* __restrict__ V2, const double* __restrict__ V3, const double* __restrict__ V4,
                                          const double* __restrict__ V5, const double* __restrict__ F12, double* __restrict__ out, long n, long chunk) {
  extern __shared__ double lds[];  // 18.2 KB like the product (occupancy), first 6 KB used as the stage
  const int lane = threadIdx.x;
  long r0 = (long)blockIdx.x * chunk;
  const long end = r0 + chunk < n ? r0 + chunk : n;
  for (; r0 < end; r0 += 64) {
    double a[9], b[9], v0[3], v1[3], v2[3], v3[3], v4[3], v5[3], f[12];
    if (VIA_LDS == 2) {
      double2 ta[5], tb[5], t0[2], t1[2], t2[2], t3[2], t4[2], t5[2], tf[6];
      pieces_load<9>(A9, r0, lane, ta); pieces_load<9>(B9, r0, lane, tb);
      pieces_load<3>(V0, r0, lane, t0); pieces_load<3>(V1, r0, lane, t1); pieces_load<3>(V2, r0, lane, t2);
      pieces_load<3>(V3, r0, lane, t3); pieces_load<3>(V4, r0, lane, t4); pieces_load<3>(V5, r0, lane, t5);
      pieces_load<12>(F12, r0, lane, tf);
      pieces_to_lds<9>(lds, lane, ta); pieces_to_lds<9>(lds + 576, lane, tb);
      __syncthreads();
      rows_from_lds<9>(lds, lane, a); rows_from_lds<9>(lds + 576, lane, b);
      __syncthreads();
      pieces_to_lds<3>(lds, lane, t0); pieces_to_lds<3>(lds + 192, lane, t1); pieces_to_lds<3>(lds + 384, lane, t2);
      pieces_to_lds<3>(lds + 576, lane, t3); pieces_to_lds<3>(lds + 768, lane, t4); pieces_to_lds<3>(lds + 960, lane, t5);
      __syncthreads();
      rows_from_lds<3>(lds, lane, v0); rows_from_lds<3>(lds + 192, lane, v1); rows_from_lds<3>(lds + 384, lane, v2);
      rows_from_lds<3>(lds + 576, lane, v3); rows_from_lds<3>(lds + 768, lane, v4); rows_from_lds<3>(lds + 960, lane, v5);
      __syncthreads();
      pieces_to_lds<12>(lds, lane, tf);
      __syncthreads();
      rows_from_lds<12>(lds, lane, f);
      __syncthreads();
    } else if (VIA_LDS == 1) {
      rows_via_lds<9>(A9, r0, lane, lds, a); rows_via_lds<9>(B9, r0, lane, lds, b);
      rows_via_lds<3>(V0, r0, lane, lds, v0); rows_via_lds<3>(V1, r0, lane, lds, v1); rows_via_lds<3>(V2, r0, lane, lds, v2);
      rows_via_lds<3>(V3, r0, lane, lds, v3); rows_via_lds<3>(V4, r0, lane, lds, v4); rows_via_lds<3>(V5, r0, lane, lds, v5);
      rows_via_lds<12>(F12, r0, lane, lds, f);
    } else {
      rows_direct<9>(A9, r0, lane, a); rows_direct<9>(B9, r0, lane, b);
      rows_direct<3>(V0, r0, lane, v0); rows_direct<3>(V1, r0, lane, v1); rows_direct<3>(V2, r0, lane, v2);
      rows_direct<3>(V3, r0, lane, v3); rows_direct<3>(V4, r0, lane, v4); rows_direct<3>(V5, r0, lane, v5);
      rows_direct<12>(F12, r0, lane, f);
    }
    double s[12];
#pragma unroll
    for (int q = 0; q < 12; q++) s[q] = f[q] + a[q % 9] * b[(q + 1) % 9] + v0[q % 3] + v1[q % 3] * v2[q % 3] + v3[q % 3] + v4[q % 3] + v5[q % 3];
    double* o = out + (r0 + lane) * 12;
#pragma unroll
    for (int q = 0; q < 12; q++) o[q] = s[q];
  }
}
int main() {
  const long n = 2097152;
  std::vector<double*> d(10);
  const int rows[10] = {9, 9, 3, 3, 3, 3, 3, 3, 12, 12};
  for (int i = 0; i < 10; i++) { hipMalloc(&d[i], n * rows[i] * 8); hipMemset(d[i], 0, n * rows[i] * 8); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)n * (48 + 12) * 8;
  const int lds_sizes[3] = {18200, 9000, 0};
  for (long chunk = 1024; chunk >= 64; chunk /= 4)
  for (int li = 0; li < 3; li += 2)
  for (int via = 0; via < 3; via++) {
    const int lds_bytes = via ? (lds_sizes[li] < 10400 ? 10400 : lds_sizes[li]) : lds_sizes[li];
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      const int reps = 5;
      for (int r = 0; r < reps; r++) {
        if (via == 2) k<2><<<(n + chunk - 1) / chunk, 64, lds_bytes>>>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], n, chunk);
        else if (via) k<1><<<(n + chunk - 1) / chunk, 64, lds_bytes>>>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], n, chunk);
        else k<0><<<(n + chunk - 1) / chunk, 64, lds_bytes>>>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], n, chunk);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("chunk %4ld LDS %5d B/WG  %s: %.1f us per pass, %.0f GB/s (480 B per record)\n", chunk, lds_bytes, via == 2 ? "B2 loads first, 3 LDS passes" : via ? "B coalesced + LDS transpose" : "A row per lane, direct", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    }
  }
  return 0;
}
