// Exercises the C++ BalanceController adapter exactly the way the reference's
// commander node drives the original class (commander_node.cpp:289-338, 507-508).
// Prints "OK" lines that tests/test_gpu_adapter.py parses.  Needs a GPU.
#include <chrono>
#include <cmath>
#include <cstdio>
// the include line of commander_node.cpp:34, gait_visualizer_node.cpp:31 and test_node.cpp:17, resolved by -I <repo>/include
#include <quadruped_controller/balance_controller.hpp>

using namespace quadruped_controller;

int main()
{
  const std::vector<std::string> leg_names = { "RL", "FL", "RR", "FR" };
  mat Ib(3, 3), S(6, 6);
  const double ib[3] = { 0.011253, 0.036203, 0.042673 };
  const double sd[6] = { 1.0, 1.0, 1.0, 10.0, 10.0, 5.0 };
  for (int i = 0; i < 3; i++) Ib(i, i) = ib[i];
  for (int i = 0; i < 6; i++) S(i, i) = sd[i];
  mat W = eye(12, 12);
  for (int i = 0; i < 12; i++) W(i, i) = 1e-5;
  const vec kff = { 0.0, 0.0, 0.15, 0.0, 0.0, 0.0 };
  const vec kp_p = { 100.0, 100.0, 100.0 }, kd_p = { 50.0, 50.0, 50.0 };
  const vec kp_w = { 5000.0, 5000.0, 5000.0 }, kd_w = { 500.0, 500.0, 500.0 };
  const BalanceController balance_controller(0.8, 11.0, 10.0, 120.0, Ib, S, W, kff, kp_p, kd_p, kp_w, kd_w, leg_names);

  const mat Rwb = eye(3, 3), Rwb_d = eye(3, 3);
  const vec x = { 0.0, 0.0, 0.26 }, zero = { 0.0, 0.0, 0.0 };
  FootholdMap feet;
  feet.emplace("RL", vec3{ -0.196, 0.127, -0.26 });
  feet.emplace("FL", vec3{ 0.196, 0.127, -0.26 });
  feet.emplace("RR", vec3{ -0.196, -0.127, -0.26 });
  feet.emplace("FR", vec3{ 0.196, -0.127, -0.26 });

  int fails = 0;
  // KAT1: every foot (0,0,-17.5353311617)
  ForceMap fm = balance_controller.control(Rwb, Rwb_d, x, zero, zero, x, zero, zero, feet);
  if (fm.size() != 4) { std::printf("FAIL kat1 size %zu\n", fm.size()); fails++; }
  for (const auto& [name, f] : fm)
    if (std::fabs(f(2) + 17.5353311617) > 1e-8 || std::fabs(f(0)) > 1e-7 || std::fabs(f(1)) > 1e-7)
    { std::printf("FAIL kat1 %s %.12g %.12g %.12g\n", name.c_str(), f(0), f(1), f(2)); fails++; }
  if (!fails) std::printf("OK kat1\n");

  // KAT3: trot, RL+FR stance -> only those two legs in the map, 35.0705746471 N each
  GaitMap gait = make_stance_gait();
  gait.at("FL").first = LegState::swing;
  gait.at("RR").first = LegState::swing;
  fm = balance_controller.control(Rwb, Rwb_d, x, zero, zero, x, zero, zero, feet, gait);
  if (fm.size() == 2 && fm.count("RL") && fm.count("FR") && std::fabs(fm.at("RL")(2) + 35.0705746471) < 1e-8) std::printf("OK kat3\n");
  else { std::printf("FAIL kat3 size %zu\n", fm.size()); fails++; }

  // missing leg -> std::out_of_range, as std::map::at in the reference
  FootholdMap bad = feet;
  bad.erase("FR");
  try { balance_controller.control(Rwb, Rwb_d, x, zero, zero, x, zero, zero, bad); std::printf("FAIL out_of_range\n"); fails++; }
  catch (const std::out_of_range&) { std::printf("OK out_of_range\n"); }

  // wrongly sized arguments: an exception (Armadillo's size-mismatch logic_error in the reference), never an overrun
  const vec six = { 0.0, 0.0, 0.26, 0.0, 0.0, 0.0 };
  try { balance_controller.control(Rwb, Rwb_d, six, zero, zero, x, zero, zero, feet); std::printf("FAIL size_check\n"); fails++; }
  catch (const std::logic_error&) { std::printf("OK size_check\n"); }
  try { balance_controller.control(eye(4, 4), Rwb_d, x, zero, zero, x, zero, zero, feet); std::printf("FAIL size_check_mat\n"); fails++; }
  catch (const std::logic_error&) { std::printf("OK size_check_mat\n"); }

  // row-major helper (balance_controller.cpp:30-41)
  mat m = { { 1.0, 2.0 }, { 3.0, 4.0 } };
  real_t arr[4];
  copy_to_real_t(m, arr);
  if (arr[0] == 1.0 && arr[1] == 2.0 && arr[2] == 3.0 && arr[3] == 4.0) std::printf("OK copy_to_real_t\n");
  else { std::printf("FAIL copy_to_real_t\n"); fails++; }

  // config 1 of the benchmark set: one robot per call through the class a ROS node links against
  // (commander_node.cpp:507-508 does this at 300 Hz)
  for (int i = 0; i < 50; i++) fm = balance_controller.control(Rwb, Rwb_d, x, zero, zero, x, zero, zero, feet);
  const int reps = 2000;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) fm = balance_controller.control(Rwb, Rwb_d, x, zero, zero, x, zero, zero, feet);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  std::printf("latency_us %.2f (control() through the C++ adapter, %d calls, RL fz %.10f)\n", us, reps, fm.at("RL")(2));
  return fails;
}
