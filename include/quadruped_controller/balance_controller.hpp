// quadruped_controller/balance_controller.hpp - forwarding header at the reference's own include path.
//
// Every caller in the reference includes the class as
//   #include <quadruped_controller/balance_controller.hpp>
// (quadruped_controller/src/commander_node.cpp:34, src/gait_visualizer_node.cpp:31, src/test_node.cpp:17).
// With `-I <this repository>/include` AHEAD of the reference's own include directory that line resolves to this
// file, so those sources compile untouched against the GPU-backed class in qc_balance_controller.hpp.
//
// The reference's header also pulled two of its neighbours into every translation unit
// (balance_controller.hpp:13-14: math/rigid3d.hpp - commander_node.cpp:173,548 use math::Quaternion through it -
// and gait.hpp, which brings types.hpp).  When those are on the include path (the reference's tree is the caller)
// they are included from there, and the adapter then uses THEIR LegState / GaitMap / FootholdMap / ForceMap /
// make_stance_gait (QC_USE_REFERENCE_TYPES) instead of its own textually identical ones.  This repository ships no
// file of those names, so inside it (tests, a caller without the reference's tree) the adapter's own types apply.
// <qpOASES.hpp> (balance_controller.hpp:11) is what is NOT forwarded: nothing of it is needed any more.
#ifndef QC_FORWARD_BALANCE_CONTROLLER_HPP
#define QC_FORWARD_BALANCE_CONTROLLER_HPP

// the reference's own guard: a later include of its file by another spelling becomes a no-op instead of a redefinition
#ifndef BALANCE_CONTROLLER_HPP
#define BALANCE_CONTROLLER_HPP
#endif

#if defined(__has_include)
#if __has_include(<quadruped_controller/math/rigid3d.hpp>)
#include <quadruped_controller/math/rigid3d.hpp>
#endif
#if __has_include(<quadruped_controller/gait.hpp>)
#include <quadruped_controller/gait.hpp>
#ifndef QC_USE_REFERENCE_TYPES
#define QC_USE_REFERENCE_TYPES 1
#endif
#endif
#endif

#include "../qc_balance_controller.hpp"

#endif  // QC_FORWARD_BALANCE_CONTROLLER_HPP
