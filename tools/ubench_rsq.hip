// Development tool: accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton steps (against long double on the host).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double y0 = __builtin_amdgcn_rsq(d);
  double t = d * y0, e = __builtin_fma(-t, y0, 1.0);
  double y1 = __builtin_fma(y0 * 0.5, e, y0);
  t = d * y1; e = __builtin_fma(-t, y1, 1.0);
  double y2 = __builtin_fma(y1 * 0.5, e, y1);
  double r0 = __builtin_amdgcn_rcp(d);
  double r1 = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-d, r1, 1.0), r1, r1);
  out[6 * i] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y2; out[6 * i + 3] = r0; out[6 * i + 4] = r1; out[6 * i + 5] = r2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  long double worst[6] = {0};
  for (int i = 0; i < n; i++) {
    const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
    for (int j = 0; j < 6; j++) {
      const long double ref = j < 3 ? rs : rc, err = fabsl(((long double)o[6 * i + j] - ref) / ref);
      if (err > worst[j]) worst[j] = err;
    }
  }
  const char* nm[6] = {"v_rsq_f64", "rsq + 1 Newton", "rsq + 2 Newton", "v_rcp_f64", "rcp + 1 Newton", "rcp + 2 Newton"};
  for (int j = 0; j < 6; j++) printf("%-16s max rel err %.3Le = 2^%.1Lf\n", nm[j], worst[j], log2l(worst[j]));
  return 0;
}
