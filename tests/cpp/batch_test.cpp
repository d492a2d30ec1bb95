// A batch caller in the reference's host language (C++), on the bare C ABI: what a massively parallel simulator
// that replaces the one-robot loop of commander_node.cpp:337-338, 507-512 would link.  Needs a GPU.
//
//   1. QC_CHECK_ABI() first (this translation unit is compiled apart from libqc_balance.so).
//   2. The five known-answer robots of SURVEY.md 8(c) through qc_control_batch_host and through qc_control_batch on
//      hipMalloc'ed arrays: closed forms for KAT1-4, the committed golden values for KAT5.
//   3. A few hundred generated robots (4-foot stance and trot diagonals, mu = 0.6) through both entry points and one by
//      one through qc_control: the three paths must agree, every force must lie in its friction cone, swing legs get 0.
//   4. Two ticks of qc_control_batch with out.active_set of tick t fed back as `warm` of tick t+1.
// Prints "OK <tag>" / "FAIL <tag> ..." lines that tests/test_gpu_adapter.py parses; exit status = number of failures.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <qc_balance.h>

namespace
{
int fails = 0;

#define HIP_OK(call)                                                                           \
  do                                                                                           \
  {                                                                                            \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
    {                                                                                          \
      std::printf("FAIL hip %s: %s\n", #call, hipGetErrorString(e_));                          \
      return 100;                                                                              \
    }                                                                                          \
  } while (0)

void check(bool ok, const char* tag, const char* detail = "")
{
  if (ok) std::printf("OK %s\n", tag);
  else { std::printf("FAIL %s %s\n", tag, detail); fails++; }
}

// mit_cheetah_config.yaml:66-99 + commander_node.cpp:289-334 (the constants every reference configuration uses)
qc_params cheetah(double mu, double fzmin, double fzmax)
{
  qc_params p;
  std::memset(&p, 0, sizeof p);
  p.mu = mu; p.mass = 11.0; p.fzmin = fzmin; p.fzmax = fzmax;
  const double ib[3] = { 0.011253, 0.036203, 0.042673 }, sd[6] = { 1, 1, 1, 10, 10, 5 };
  for (int i = 0; i < 3; i++) p.Ib[4 * i] = ib[i];
  for (int i = 0; i < 6; i++) p.S[7 * i] = sd[i];
  for (int i = 0; i < 12; i++) p.W[13 * i] = 1e-5;
  p.kff[2] = 0.15;
  for (int i = 0; i < 3; i++) { p.kp_p[i] = 100; p.kd_p[i] = 50; p.kp_w[i] = 5000; p.kd_w[i] = 500; }
  p.max_iter = 200;
  return p;
}

// struct-of-arrays batch in host memory, one array per argument of control() (qc_batch_in)
struct Batch
{
  size_t n = 0;
  std::vector<double> Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet;
  std::vector<std::uint8_t> stance;
  void push(const double* R, const double* Rd, const double* px, const double* v, const double* om, const double* pxd,
            const double* vd, const double* omd, const double* f, const std::uint8_t* s)
  {
    Rwb.insert(Rwb.end(), R, R + 9); Rwb_d.insert(Rwb_d.end(), Rd, Rd + 9);
    x.insert(x.end(), px, px + 3); xdot.insert(xdot.end(), v, v + 3); w.insert(w.end(), om, om + 3);
    x_d.insert(x_d.end(), pxd, pxd + 3); xdot_d.insert(xdot_d.end(), vd, vd + 3); w_d.insert(w_d.end(), omd, omd + 3);
    feet.insert(feet.end(), f, f + 12); stance.insert(stance.end(), s, s + 4);
    n++;
  }
  qc_batch_in view() const
  {
    qc_batch_in in;
    std::memset(&in, 0, sizeof in);  // every optional member NULL
    in.Rwb = Rwb.data(); in.Rwb_d = Rwb_d.data(); in.x = x.data(); in.xdot = xdot.data(); in.w = w.data();
    in.x_d = x_d.data(); in.xdot_d = xdot_d.data(); in.w_d = w_d.data(); in.feet = feet.data(); in.stance = stance.data();
    return in;
  }
};

struct Result
{
  std::vector<double> grf;
  std::vector<std::int32_t> status, iterations;
  std::vector<std::uint32_t> active;
  explicit Result(size_t n) : grf(12 * n, -1.0), status(n, -1), iterations(n, -1), active(n, 0) {}
  qc_batch_out view()
  {
    qc_batch_out o;
    std::memset(&o, 0, sizeof o);
    o.grf_body = grf.data(); o.status = status.data(); o.iterations = iterations.data(); o.active_set = active.data();
    return o;
  }
};

// device copy of a Batch + outputs, as a simulator would hold them
struct DeviceBatch
{
  size_t n = 0;
  std::vector<void*> owned;
  qc_batch_in in;
  qc_batch_out out;
  template <class T> T* up(const std::vector<T>& h)
  {
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    owned.push_back(d);
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return static_cast<T*>(d);
  }
  template <class T> T* alloc(size_t count)
  {
    void* d = nullptr;
    if (hipMalloc(&d, count * sizeof(T)) != hipSuccess) return nullptr;
    owned.push_back(d);
    return static_cast<T*>(d);
  }
  bool init(const Batch& b)
  {
    n = b.n;
    std::memset(&in, 0, sizeof in);
    std::memset(&out, 0, sizeof out);
    in.Rwb = up(b.Rwb); in.Rwb_d = up(b.Rwb_d); in.x = up(b.x); in.xdot = up(b.xdot); in.w = up(b.w);
    in.x_d = up(b.x_d); in.xdot_d = up(b.xdot_d); in.w_d = up(b.w_d); in.feet = up(b.feet); in.stance = up(b.stance);
    out.grf_body = alloc<double>(12 * n); out.status = alloc<std::int32_t>(n);
    out.iterations = alloc<std::int32_t>(n); out.active_set = alloc<std::uint32_t>(n);
    return in.Rwb && in.Rwb_d && in.x && in.xdot && in.w && in.x_d && in.xdot_d && in.w_d && in.feet && in.stance &&
           out.grf_body && out.status && out.iterations && out.active_set;
  }
  bool down(Result& r) const
  {
    return hipMemcpy(r.grf.data(), out.grf_body, 12 * n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(r.status.data(), out.status, n * sizeof(std::int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(r.iterations.data(), out.iterations, n * sizeof(std::int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(r.active.data(), out.active_set, n * sizeof(std::uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
  }
  ~DeviceBatch() { for (void* p : owned) (void)hipFree(p); }
};

const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
const double ZERO[3] = { 0, 0, 0 };
const std::uint8_t ALL[4] = { 1, 1, 1, 1 };

void nominal_feet(double dx, double z, double* f)  // foot_planner.cpp:28-30 + stand height commander_node.cpp:354, body frame
{
  const double px[4] = { -0.196, 0.196, -0.196, 0.196 }, py[4] = { 0.127, 0.127, -0.127, -0.127 };
  for (int i = 0; i < 4; i++) { f[3 * i] = px[i] - dx; f[3 * i + 1] = py[i]; f[3 * i + 2] = -z; }
}

double maxabs_diff(const std::vector<double>& a, const std::vector<double>& b)
{
  double m = 0;
  for (size_t i = 0; i < a.size(); i++) m = std::fmax(m, std::fabs(a[i] - b[i]));
  return m;
}

// splitmix64: a self-contained generator (the parity tests proper use the package's counter-based PRNG from Python)
struct Rng
{
  std::uint64_t s;
  double u(double lo, double hi)
  {
    s += 0x9E3779B97F4A7C15ull;
    std::uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return lo + (hi - lo) * (static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0));
  }
};

void rodrigues(const double* r, double* R)
{
  const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const double a = th > 1e-12 ? std::sin(th) / th : 1.0, b = th > 1e-12 ? (1.0 - std::cos(th)) / (th * th) : 0.5;
  const double K[9] = { 0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0 };
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      double kk = 0;
      for (int k = 0; k < 3; k++) kk += K[3 * i + k] * K[3 * k + j];
      R[3 * i + j] = (i == j) + a * K[3 * i + j] + b * kk;
    }
}

// the distribution of SURVEY.md 8(d) config 2 (every third robot on a trot diagonal instead of four feet)
Batch generated(size_t n, std::uint64_t seed)
{
  Batch b;
  Rng g{ seed };
  for (size_t r = 0; r < n; r++)
  {
    double rv[3], R[9], x[3], v[3], om[3], feet[12];
    for (double& c : rv) c = g.u(-0.1, 0.1);
    rodrigues(rv, R);
    x[0] = g.u(-0.03, 0.03); x[1] = g.u(-0.03, 0.03); x[2] = 0.26 + g.u(-0.03, 0.03);
    for (double& c : v) c = g.u(-0.3, 0.3);
    for (double& c : om) c = g.u(-0.5, 0.5);
    const double px[4] = { -0.196, 0.196, -0.196, 0.196 }, py[4] = { 0.127, 0.127, -0.127, -0.127 };
    std::uint8_t st[4] = { 1, 1, 1, 1 };
    if (r % 3 == 2) { const bool d = (r / 3) % 2; st[d ? 0 : 1] = 0; st[d ? 3 : 2] = 0; }
    for (int i = 0; i < 4; i++)
    {
      const double pw[3] = { px[i] + g.u(-0.03, 0.03) - x[0], py[i] + g.u(-0.03, 0.03) - x[1], (st[i] ? 0.0 : g.u(0.0, 0.08)) - x[2] };
      for (int k = 0; k < 3; k++) feet[3 * i + k] = R[k] * pw[0] + R[3 + k] * pw[1] + R[6 + k] * pw[2];  // Rwb^T (p_world - x)
    }
    const double xd[3] = { 0, 0, 0.26 };
    b.push(R, I3, x, v, om, xd, ZERO, ZERO, feet, st);
  }
  return b;
}
}  // namespace

int main()
{
  if (QC_CHECK_ABI() != QC_OK) { std::printf("FAIL abi %s\n", qc_last_error()); return 1; }
  std::printf("OK abi v%d\n", qc_abi_version());
  HIP_OK(hipSetDevice(0));

  // ---- known answers --------------------------------------------------------------------------------------------
  struct Kat { const char* tag; double mu, fzmin, fzmax, dx, xz; std::uint8_t st[4]; };
  const Kat kats[5] = { { "kat1", 0.8, 10, 120, 0.0, 0.26, { 1, 1, 1, 1 } }, { "kat2", 0.8, 10, 120, 0.0, 0.36, { 1, 1, 1, 1 } },
                        { "kat3", 0.8, 10, 120, 0.0, 0.26, { 1, 0, 0, 1 } }, { "kat4", 0.8, 10, 15, 0.0, 0.26, { 1, 1, 1, 1 } },
                        { "kat5", 0.6, 10, 120, -0.1, 0.26, { 1, 1, 1, 1 } } };
  for (const Kat& k : kats)
  {
    const qc_params p = cheetah(k.mu, k.fzmin, k.fzmax);
    qc_handle* h = nullptr;
    if (qc_create(&p, 0, &h) != QC_OK) { std::printf("FAIL %s create %s\n", k.tag, qc_last_error()); return 1; }
    Batch b;
    double feet[12];
    const double x[3] = { k.dx, 0, k.xz }, xd[3] = { 0, 0, 0.26 };
    nominal_feet(k.dx, k.xz, feet);
    for (int rep = 0; rep < 3; rep++) b.push(I3, I3, x, ZERO, ZERO, xd, ZERO, ZERO, feet, k.st);  // n = 3: a batch, not the n = 1 route
    Result rh(b.n), rd(b.n);
    const qc_batch_in in = b.view();
    const qc_batch_out oh = rh.view();
    int rc = qc_control_batch_host(h, b.n, &in, nullptr, &oh);
    DeviceBatch db;
    if (!db.init(b)) { std::printf("FAIL %s hipMalloc\n", k.tag); return 1; }
    int rc2 = qc_control_batch(h, db.n, &db.in, nullptr, &db.out, nullptr);
    HIP_OK(hipDeviceSynchronize());
    if (!db.down(rd)) { std::printf("FAIL %s copy back\n", k.tag); return 1; }
    bool ok = rc == QC_OK && rc2 == QC_OK;
    for (const Result* r : { &rh, &rd })
      for (size_t i = 0; i < b.n && ok; i++)
      {
        const double* g = &r->grf[12 * i];
        ok = r->status[i] == QC_SOLVED;
        for (int leg = 0; leg < 4 && ok; leg++)
        {
          double ex = 0, ez = 0;  // expected body-frame force = -f_world (Rwb = I)
          if (!std::strcmp(k.tag, "kat1")) ez = -70.1415 / 4.00001;
          if (!std::strcmp(k.tag, "kat2")) ez = -10.0;
          if (!std::strcmp(k.tag, "kat3")) ez = k.st[leg] ? -70.1415 / 2.00001 : 0.0;
          if (!std::strcmp(k.tag, "kat4")) ez = -15.0;
          if (!std::strcmp(k.tag, "kat5"))  // tests/golden/balance_golden.json KAT5_cone_saturation
          { ex = (leg % 2 == 0) ? -22.245772728327704 : -6.0; ez = (leg % 2 == 0) ? -37.07628788054616 : -10.0; }
          ok = std::fabs(g[3 * leg] - ex) < 2e-7 && std::fabs(g[3 * leg + 1]) < 2e-7 && std::fabs(g[3 * leg + 2] - ez) < 2e-7;
          if (!k.st[leg]) ok = ok && g[3 * leg] == 0.0 && g[3 * leg + 1] == 0.0 && g[3 * leg + 2] == 0.0;
        }
      }
    char detail[160];
    std::snprintf(detail, sizeof detail, "rc %d/%d status %d/%d fz %.12g/%.12g (%s)", rc, rc2, rh.status[0], rd.status[0], rh.grf[2], rd.grf[2],
                  rc || rc2 ? qc_last_error() : "");
    check(ok, k.tag, detail);
    qc_destroy(h);
  }

  // ---- generated robots: host arrays vs device arrays vs one robot per call --------------------------------------
  const qc_params p = cheetah(0.6, 10, 120);
  qc_handle* h = nullptr;
  if (qc_create(&p, 0, &h) != QC_OK) { std::printf("FAIL create %s\n", qc_last_error()); return 1; }
  const size_t n = 600;
  const Batch b = generated(n, 0x5EED0C99ull);
  Result rh(n), rd(n), r1(n);
  {
    const qc_batch_in in = b.view();
    const qc_batch_out oh = rh.view();
    const int rc = qc_control_batch_host(h, n, &in, nullptr, &oh);
    check(rc == QC_OK, "batch_host_rc", rc ? qc_last_error() : "");
  }
  DeviceBatch db;
  if (!db.init(b)) { std::printf("FAIL hipMalloc\n"); return 1; }
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  {
    const int rc = qc_control_batch(h, n, &db.in, nullptr, &db.out, stream);  // asynchronous on the caller's stream
    check(rc == QC_OK, "batch_device_rc", rc ? qc_last_error() : "");
    HIP_OK(hipStreamSynchronize(stream));
    if (!db.down(rd)) { std::printf("FAIL copy back\n"); return 1; }
  }
  size_t solved = 0, cone_bad = 0, swing_bad = 0;
  for (size_t i = 0; i < n; i++)
  {
    solved += rd.status[i] == QC_SOLVED && rh.status[i] == QC_SOLVED;
    const double* R = &b.Rwb[9 * i];
    for (int leg = 0; leg < 4; leg++)
    {
      const double* gb = &rd.grf[12 * i + 3 * leg];
      if (!b.stance[4 * i + leg]) { swing_bad += !(gb[0] == 0.0 && gb[1] == 0.0 && gb[2] == 0.0); continue; }
      double fw[3];  // f_world = -Rwb g_body (balance_controller.cpp:218-232 inverted)
      for (int k = 0; k < 3; k++) fw[k] = -(R[3 * k] * gb[0] + R[3 * k + 1] * gb[1] + R[3 * k + 2] * gb[2]);
      const double tol = 1e-7;
      cone_bad += !(std::fabs(fw[0]) <= p.mu * fw[2] + tol && std::fabs(fw[1]) <= p.mu * fw[2] + tol && fw[2] >= p.fzmin - tol &&
                    fw[2] <= p.fzmax + tol);
    }
  }
  char detail[200];
  std::snprintf(detail, sizeof detail, "%zu of %zu solved", solved, n);
  check(solved == n, "batch_all_solved", detail);
  std::snprintf(detail, sizeof detail, "%zu stance feet outside the cone, %zu swing feet with a force", cone_bad, swing_bad);
  check(cone_bad == 0 && swing_bad == 0, "batch_cone_and_swing", detail);
  const double d_hd = maxabs_diff(rh.grf, rd.grf);
  std::snprintf(detail, sizeof detail, "max |host - device| = %.3g N", d_hd);
  check(d_hd <= 1e-9, "batch_host_equals_device", detail);
  std::printf("info host_vs_device_max_abs %.3g (0 = bit-equal)\n", d_hd);

  // one robot per call, the way commander_node.cpp:507-508 does it (hot-started from the previous robot's working set)
  int rc1 = QC_OK;
  for (size_t i = 0; i < n && rc1 == QC_OK; i++)
    rc1 = qc_control(h, &b.Rwb[9 * i], &b.Rwb_d[9 * i], &b.x[3 * i], &b.xdot[3 * i], &b.w[3 * i], &b.x_d[3 * i], &b.xdot_d[3 * i], &b.w_d[3 * i],
                     &b.feet[12 * i], &b.stance[4 * i], &r1.grf[12 * i], &r1.status[i]);
  const double d_1 = maxabs_diff(r1.grf, rd.grf);
  std::snprintf(detail, sizeof detail, "rc %d max |qc_control - batch| = %.3g N", rc1, d_1);
  check(rc1 == QC_OK && d_1 <= 1e-6, "single_equals_batch", detail);  // another kernel instantiation and start: the same minimiser

  // ---- tick t's active set as tick t+1's warm start ----------------------------------------------------------------
  {
    const int rc = qc_control_batch(h, n, &db.in, db.out.active_set, &db.out, stream);
    HIP_OK(hipStreamSynchronize(stream));
    Result rw(n);
    if (!db.down(rw)) { std::printf("FAIL copy back\n"); return 1; }
    long it_cold = 0, it_warm = 0;
    for (size_t i = 0; i < n; i++) { it_cold += rd.iterations[i]; it_warm += rw.iterations[i]; }
    const double d_w = maxabs_diff(rw.grf, rd.grf);
    std::snprintf(detail, sizeof detail, "rc %d max |warm - cold| = %.3g N, recalculations cold %ld warm %ld", rc, d_w, it_cold, it_warm);
    check(rc == QC_OK && d_w <= 1e-6 && it_warm < it_cold, "warm_start_round_trip", detail);
    std::printf("info recalculations cold %.2f warm %.2f per robot\n", double(it_cold) / n, double(it_warm) / n);
  }

  // ---- misuse: the ABI reports, it does not crash --------------------------------------------------------------------
  {
    qc_batch_in bad = db.in;
    bad.x = nullptr;
    const int rc = qc_control_batch(h, n, &bad, nullptr, &db.out, stream);
    check(rc == QC_ERR_INVALID && std::strlen(qc_last_error()) > 0, "null_argument_refused", qc_last_error());
  }
  HIP_OK(hipStreamDestroy(stream));
  qc_destroy(h);
  std::printf("%s (%d failures)\n", fails ? "FAILED" : "PASSED", fails);
  return fails;
}
