// qc_balance_controller.hpp - C++ host-side drop-in for the reference class
//   quadruped_controller::BalanceController
//   (quadruped_controller/include/quadruped_controller/balance_controller.hpp:66-179)
// on top of the C ABI in qc_balance.h.  Header-only; link with -lqc_balance.
//
// Same namespace, class name, constructor and control() signatures, same
// helper functions (copy_to_real_t / copy_from_real_t / print_real_t,
// balance_controller.hpp:38-63), same boundary types (types.hpp:91-119) and
// the same error behaviour:
//   * std::out_of_range if a leg is missing from foot_map / gait_map
//     (std::map::at, balance_controller.cpp:115,223,312);
//   * solver failure -> error message + EMPTY ForceMap (balance_controller.cpp:182-216);
//   * only stance legs appear in the returned map (balance_controller.cpp:222-228).
// What changes: qpOASES / Drake / ROS are gone - the numerics run on the GPU
// behind qc_control(); ROS_ERROR becomes a replaceable log callback.
//
// With Armadillo installed (`__has_include(<armadillo>)`) the types are the
// reference's own arma::mat / arma::vec / arma::vec3, so a node written against
// the reference header compiles unchanged.  Without it (this image) a minimal
// column-major mat/vec with the same element accessors is provided.
#ifndef QC_BALANCE_CONTROLLER_HPP
#define QC_BALANCE_CONTROLLER_HPP

#include <cstdint>
#include <cstdio>
#include <functional>
#include <initializer_list>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "qc_balance.h"

#if defined(__has_include)
#if __has_include(<armadillo>) && !defined(QC_NO_ARMADILLO)
#include <armadillo>
#define QC_HAVE_ARMADILLO 1
#endif
#endif

namespace quadruped_controller
{
#ifdef QC_HAVE_ARMADILLO
using arma::mat;
using arma::vec;
using arma::vec3;
#else
/** Minimal dense column vector with Armadillo-style accessors. */
class vec
{
public:
  vec() = default;
  explicit vec(unsigned int n, double fill = 0.0) : d_(n, fill) {}
  vec(std::initializer_list<double> l) : d_(l) {}
  double& operator()(unsigned int i) { return d_.at(i); }
  const double& operator()(unsigned int i) const { return d_.at(i); }
  unsigned int size() const { return static_cast<unsigned int>(d_.size()); }
  unsigned int n_rows = 0;  // kept for source compatibility; use size()
  const double* memptr() const { return d_.data(); }

private:
  std::vector<double> d_;
};
typedef vec vec3;

/** Minimal dense matrix, column-major like arma::mat. */
class mat
{
public:
  mat() : n_rows(0), n_cols(0) {}
  mat(unsigned int r, unsigned int c, double fill = 0.0) : n_rows(r), n_cols(c), d_(static_cast<size_t>(r) * c, fill) {}
  mat(std::initializer_list<std::initializer_list<double>> rows) : n_rows(static_cast<unsigned int>(rows.size())), n_cols(0)
  {
    if (n_rows) n_cols = static_cast<unsigned int>(rows.begin()->size());
    d_.assign(static_cast<size_t>(n_rows) * n_cols, 0.0);
    unsigned int i = 0;
    for (const auto& r : rows)
    {
      unsigned int j = 0;
      for (double v : r) (*this)(i, j++) = v;
      i++;
    }
  }
  double& operator()(unsigned int i, unsigned int j) { return d_.at(static_cast<size_t>(j) * n_rows + i); }
  const double& operator()(unsigned int i, unsigned int j) const { return d_.at(static_cast<size_t>(j) * n_rows + i); }
  unsigned int n_rows, n_cols;

private:
  std::vector<double> d_;
};

inline mat eye(unsigned int r, unsigned int c)
{
  mat m(r, c);
  for (unsigned int i = 0; i < r && i < c; i++) m(i, i) = 1.0;
  return m;
}
#endif  // QC_HAVE_ARMADILLO

#ifndef QC_USE_REFERENCE_TYPES  // define it when the reference's own types.hpp / gait.hpp are also included
/** @brief Leg state in gait (types.hpp:91-95) */
enum LegState
{
  swing = 0,
  stance = 1
};
/** @brief map leg name to LegState and phase (types.hpp:100) */
typedef std::map<std::string, std::pair<LegState, double>> GaitMap;
/** @brief map leg name to foot position (types.hpp:108) */
typedef std::map<std::string, vec3> FootholdMap;
/** @brief map leg name to ground reaction forces [fx, fy, fz] (types.hpp:119) */
typedef std::map<std::string, vec3> ForceMap;

/** @brief all four legs in stance, phase 0 (gait.cpp:24-34) */
inline GaitMap make_stance_gait()
{
  GaitMap gait_map;
  gait_map.emplace("RL", std::make_pair(LegState::stance, 0.0));
  gait_map.emplace("FL", std::make_pair(LegState::stance, 0.0));
  gait_map.emplace("RR", std::make_pair(LegState::stance, 0.0));
  gait_map.emplace("FR", std::make_pair(LegState::stance, 0.0));
  return gait_map;
}

#endif  // QC_USE_REFERENCE_TYPES

typedef double real_t;  // qpOASES::real_t in the reference (balance_controller.hpp:29)

/** @brief Copy vector to array (balance_controller.hpp:38) */
inline void copy_to_real_t(const vec& source, real_t* target)
{
  for (unsigned int i = 0; i < source.size(); i++) target[i] = source(i);
}
/** @brief Copy matrix to ROW-MAJOR array (balance_controller.hpp:45, balance_controller.cpp:30-41) */
inline void copy_to_real_t(const mat& source, real_t* target)
{
  unsigned int count = 0;
  for (unsigned int i = 0; i < source.n_rows; i++)
    for (unsigned int j = 0; j < source.n_cols; j++) target[count++] = source(i, j);
}
/** @brief Copy array to vector (balance_controller.hpp:53) */
inline vec copy_from_real_t(const real_t* const source, unsigned int n_rows)
{
  vec target(n_rows);
  for (unsigned int i = 0; i < n_rows; i++) target(i) = source[i];
  return target;
}
/** @brief Print array contents to stdout (balance_controller.hpp:62) */
inline void print_real_t(const real_t* const array, unsigned int n_rows, unsigned int n_cols, const std::string& msg = "")
{
  std::cout << msg << "\n";
  unsigned int count = 0;
  for (unsigned int i = 0; i < n_rows; i++)
  {
    for (unsigned int j = 0; j < n_cols; j++) std::cout << array[count++] << " ";
    std::cout << std::endl;
  }
}

/** @brief Reactive optimal control strategy (GPU-backed drop-in) */
class BalanceController
{
public:
  /** Log sink replacing ROS_ERROR_STREAM_NAMED("Balance Controller", ...). */
  typedef std::function<void(const std::string&)> LogFn;

  /** Same parameter list as balance_controller.hpp:85-88; `device` selects the GPU. */
  BalanceController(double mu, double mass, double fzmin, double fzmax, const mat& Ib, const mat& S, const mat& W,
                    const vec& kff, const vec& kp_p, const vec& kd_p, const vec& kp_w, const vec& kd_w,
                    const std::vector<std::string>& leg_names, int device = 0)
    : leg_names_(leg_names), log_([](const std::string& m) { std::fprintf(stderr, "[Balance Controller] %s\n", m.c_str()); })
  {
    if (leg_names_.size() != 4) throw std::invalid_argument("BalanceController: leg_names must hold 4 names");
    if (Ib.n_rows != 3 || Ib.n_cols != 3 || S.n_rows != 6 || S.n_cols != 6 || W.n_rows != 12 || W.n_cols != 12 ||
        kff.size() != 6 || kp_p.size() != 3 || kd_p.size() != 3 || kp_w.size() != 3 || kd_w.size() != 3)
      throw std::invalid_argument("BalanceController: wrong matrix/vector size");
    qc_params p;
    p.mu = mu; p.mass = mass; p.fzmin = fzmin; p.fzmax = fzmax;
    copy_to_real_t(Ib, p.Ib);
    copy_to_real_t(S, p.S);
    copy_to_real_t(W, p.W);
    copy_to_real_t(kff, p.kff);
    copy_to_real_t(kp_p, p.kp_p);
    copy_to_real_t(kd_p, p.kd_p);
    copy_to_real_t(kp_w, p.kp_w);
    copy_to_real_t(kd_w, p.kd_w);
    p.max_iter = 200;  // nWSR_, balance_controller.cpp:85
    p.reserved = 0;
    // this translation unit and the shared library must agree on the structs that cross the C ABI
    if (QC_CHECK_ABI() != QC_OK) throw std::runtime_error(std::string("BalanceController: ") + qc_last_error());
    qc_handle* h = nullptr;
    if (qc_create(&p, device, &h) != QC_OK) throw std::runtime_error(std::string("BalanceController: ") + qc_last_error());
    handle_ = std::shared_ptr<qc_handle>(h, qc_destroy);
  }

  void setLogCallback(LogFn fn) { log_ = std::move(fn); }

  /** Same contract as balance_controller.hpp:104-107. */
  ForceMap control(const mat& Rwb, const mat& Rwb_d, const vec& x, const vec& xdot, const vec& w, const vec& x_d,
                   const vec& xdot_d, const vec& w_d, const FootholdMap& foot_map,
                   const GaitMap& gait_map = make_stance_gait()) const
  {
    ForceMap force_map;
    double a_Rwb[9], a_Rwbd[9], a_x[3], a_xdot[3], a_w[3], a_xd[3], a_xdotd[3], a_wd[3], feet[12], grf[12];
    std::uint8_t stance_flags[4];
    for (unsigned int i = 0; i < 4; i++)
    {
      const vec3& p = foot_map.at(leg_names_.at(i));  // throws std::out_of_range like the reference
      feet[3 * i] = p(0); feet[3 * i + 1] = p(1); feet[3 * i + 2] = p(2);
      stance_flags[i] = gait_map.at(leg_names_.at(i)).first == LegState::stance ? 1 : 0;
    }
    // Armadillo throws a size-mismatch std::logic_error when the reference multiplies wrongly sized arguments
    // (balance_controller.cpp:126-139); the same misuse must not run past the fixed-size records here.
    if (Rwb.n_rows != 3 || Rwb.n_cols != 3 || Rwb_d.n_rows != 3 || Rwb_d.n_cols != 3 || x.size() != 3 || xdot.size() != 3 ||
        w.size() != 3 || x_d.size() != 3 || xdot_d.size() != 3 || w_d.size() != 3)
      throw std::logic_error("BalanceController::control: rotations must be 3x3 and vectors of size 3");
    copy_to_real_t(Rwb, a_Rwb);
    copy_to_real_t(Rwb_d, a_Rwbd);
    copy_to_real_t(x, a_x);
    copy_to_real_t(xdot, a_xdot);
    copy_to_real_t(w, a_w);
    copy_to_real_t(x_d, a_xd);
    copy_to_real_t(xdot_d, a_xdotd);
    copy_to_real_t(w_d, a_wd);
    std::int32_t status = QC_NOT_PD;
    const int rc = qc_control(handle_.get(), a_Rwb, a_Rwbd, a_x, a_xdot, a_w, a_xd, a_xdotd, a_wd, feet, stance_flags, grf, &status);
    if (rc != QC_OK)
    {
      log_(std::string("qc_control failed: ") + qc_last_error());
      return force_map;
    }
    if (status != QC_SOLVED)
    {
      log_("Balance Controller QP Solver Failed");  // balance_controller.cpp:214
      return force_map;
    }
    for (unsigned int i = 0; i < 4; i++)
    {
      if (stance_flags[i])
      {
#ifdef QC_HAVE_ARMADILLO
        vec3 fb = { grf[3 * i], grf[3 * i + 1], grf[3 * i + 2] };
#else
        vec3 fb{ grf[3 * i], grf[3 * i + 1], grf[3 * i + 2] };
#endif
        force_map.emplace(leg_names_.at(i), fb);
      }
    }
    return force_map;
  }

private:
  std::shared_ptr<qc_handle> handle_;  // the solver state the reference keeps in `mutable SQProblem QPSolver_`
  std::vector<std::string> leg_names_;
  LogFn log_;
};
}  // namespace quadruped_controller
#endif
