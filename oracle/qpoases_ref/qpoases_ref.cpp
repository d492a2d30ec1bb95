// TEST INFRASTRUCTURE (optional, never part of the product): the LITERAL solver call sequence of the reference,
//   quadruped_controller/src/quadruped_controller/balance_controller.cpp:84-91 (SQProblem(12, 20), nWSR 200,
//   cputime 0.01 s, PL_NONE) and :165-216 (init on the first tick, hotstart with new H, g, A, lbA, ubA afterwards,
//   isSolved / getPrimalSolution),
// run on QP data the oracle assembled (H, g, C, lbC, ubC exactly as copy_to_real_t lays them out, row-major).
// It only builds where qpOASES itself is installed (the reference pins "qpOASES master @326a651", README.md:102,
// located through qpOASES_SOURCE_DIR / qpOASES_BINARY_DIR as cmake/FindqpOASES.cmake:34-45 does) - see run.py,
// which reports "skipped" when it is not.  Nothing here emulates qpOASES.
//
// usage: qpoases_ref <problems.txt> <results.txt> [--fresh] [--no-cputime-cap] [--repeat R]
//   problems.txt: N, then per instance 144 H + 12 g + 240 C + 20 lbC + 20 ubC doubles (row-major)
//   results.txt : per instance "status nWSR cpu_seconds x0 ... x11"; status 0 solved, 1 init/hotstart returned an
//                 error (the reference's empty ForceMap), 2 !isSolved().  Last line: "seconds_per_pass <t>".
//   default     : ONE SQProblem for the whole list - init for the first instance, hotstart afterwards, as one
//                 controller object does tick after tick; --fresh makes a new SQProblem (init) per instance.
#include <qpOASES.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

USING_NAMESPACE_QPOASES

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s problems.txt results.txt [--fresh] [--no-cputime-cap] [--repeat R]\n", argv[0]); return 2; }
  bool fresh = false, cap = true;
  int repeat = 1;
  for (int i = 3; i < argc; i++) {
    if (!std::strcmp(argv[i], "--fresh")) fresh = true;
    else if (!std::strcmp(argv[i], "--no-cputime-cap")) cap = false;
    else if (!std::strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = std::atoi(argv[++i]);
  }
  std::FILE* in = std::fopen(argv[1], "r");
  if (!in) { std::perror(argv[1]); return 2; }
  int n = 0;
  if (std::fscanf(in, "%d", &n) != 1 || n <= 0) { std::fprintf(stderr, "bad problem file\n"); return 2; }
  const int per = 144 + 12 + 240 + 20 + 20;
  std::vector<real_t> data((size_t)n * per);
  for (size_t i = 0; i < data.size(); i++) {
    double v;
    if (std::fscanf(in, "%lf", &v) != 1) { std::fprintf(stderr, "short problem file\n"); return 2; }
    data[i] = (real_t)v;
  }
  std::fclose(in);
  std::vector<real_t> x((size_t)n * 12, 0.0), cpu(n, 0.0);
  std::vector<int> status(n, 1), wsr(n, 0);
  double pass_s = 0.0;
  for (int rep = 0; rep < repeat; rep++) {
    SQProblem* qp = new SQProblem(12, 20);      // num_variables_qp_, num_constraints_qp_ (balance_controller.hpp)
    qp->setPrintLevel(PL_NONE);                 // balance_controller.cpp:94-95
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; k++) {
      real_t* p = &data[(size_t)k * per];
      real_t *H = p, *g = p + 144, *C = p + 156, *lbC = p + 396, *ubC = p + 416;
      if (fresh && k > 0) { delete qp; qp = new SQProblem(12, 20); qp->setPrintLevel(PL_NONE); }
      int_t nWSR = 200;                          // nWSR_, balance_controller.cpp:85
      real_t cputime = 0.01;                     // cpu_time_, balance_controller.cpp:91
      returnValue rv;
      if (!qp->isInitialised()) rv = qp->init(H, g, C, 0, 0, lbC, ubC, nWSR, cap ? &cputime : 0);       // :177-178
      else rv = qp->hotstart(H, g, C, 0, 0, lbC, ubC, nWSR, cap ? &cputime : 0);                          // :191-193
      wsr[k] = (int)nWSR;
      cpu[k] = cap ? cputime : 0.0;
      if (rv != SUCCESSFUL_RETURN) { status[k] = 1; continue; }                                           // :180-188, :195-201
      if (qp->isSolved()) { qp->getPrimalSolution(&x[(size_t)k * 12]); status[k] = 0; }                   // :206-210
      else status[k] = 2;                                                                                // :212-216
    }
    pass_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    delete qp;
  }
  std::FILE* out = std::fopen(argv[2], "w");
  if (!out) { std::perror(argv[2]); return 2; }
  for (int k = 0; k < n; k++) {
    std::fprintf(out, "%d %d %.9g", status[k], wsr[k], (double)cpu[k]);
    for (int j = 0; j < 12; j++) std::fprintf(out, " %.17g", (double)x[(size_t)k * 12 + j]);
    std::fprintf(out, "\n");
  }
  std::fprintf(out, "seconds_per_pass %.9g\n", pass_s);
  std::fclose(out);
  return 0;
}
