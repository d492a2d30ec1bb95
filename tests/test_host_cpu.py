"""CPU: host logic + the C-ABI library loads and exports what include/*.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_abi_exports_match_header(built):
    hdr = open(os.path.join(ROOT, "include", "qc_balance.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    inline = set(re.findall(r"static inline \w+ (qc_[a-z_]+)\s*\(", hdr))
    assert inline == {"qc_create"}  # ABI v6: the ABI guard travels with the constructor every C / C++ caller writes
    hdr = re.sub(r"static inline [^{]*\{.*?\n\}", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qc_[a-z_]+)\s*\(", hdr))
    assert {"qc_create_abi", "qc_destroy", "qc_control_batch", "qc_control_batch_host", "qc_control", "qc_last_error"} <= declared
    lib = ctypes.CDLL(os.path.join(ROOT, "quadruped_control_amd", "libqc_balance.so"))
    assert not hasattr(lib, "qc_create")  # a binary built against ABI <= 5 fails to load instead of running unguarded
    for name in declared:
        assert hasattr(lib, name), f"libqc_balance.so lacks {name}"
    from quadruped_control_amd import _lib

    assert set(_lib.EXPORTS) == declared
    assert _lib.load().qc_abi_version() == _lib.ABI_VERSION == 6
    m = re.search(r"#define QC_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "qc_balance.h")).read())
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_abi_check_refuses_a_caller_built_against_another_revision(built):
    """qc_batch_in has grown with every ABI revision and carries no size field, so an old caller
    against a new .so would have its struct read past the end.  qc_check_abi (called by _lib.load() and by the C++
    adapter's constructor) refuses the pair before any struct crosses the boundary; needs no device."""
    from quadruped_control_amd import _lib

    lib = _lib.load()  # (has already passed the check with this package's own structures)
    sizes = (ctypes.sizeof(_lib.QcParams), ctypes.sizeof(_lib.QcBatchIn), ctypes.sizeof(_lib.QcBatchOut))
    assert lib.qc_check_abi(_lib.ABI_VERSION, *sizes) == _lib.QC_OK
    assert lib.qc_check_abi(_lib.ABI_VERSION - 1, *sizes) == _lib.QC_ERR_ABI  # an ABI v5 caller
    assert "ABI v5" in _lib.last_error() and "ABI v6" in _lib.last_error()
    assert lib.qc_check_abi(_lib.ABI_VERSION, sizes[0], sizes[1] - 8, sizes[2]) == _lib.QC_ERR_ABI  # ABI v2's qc_batch_in (no gait_dt)
    assert lib.qc_check_abi(_lib.ABI_VERSION, sizes[0], sizes[1], sizes[2] - 8) == _lib.QC_ERR_ABI  # ABI v1's qc_batch_out (no joint_tau)
    # the guard is not advisory any more - the constructor itself refuses (no device needed to get that far)
    h = ctypes.c_void_p()
    p = _lib.QcParams()
    assert lib.qc_create_abi(ctypes.byref(p), 0, ctypes.byref(h), _lib.ABI_VERSION - 1, *sizes) == _lib.QC_ERR_ABI and not h.value
    assert lib.qc_create_abi(ctypes.byref(p), 0, ctypes.byref(h), _lib.ABI_VERSION, sizes[0], sizes[1] - 8, sizes[2]) == _lib.QC_ERR_ABI and not h.value
    assert "qc_batch_in" in _lib.last_error()
    # the adapter and the loader also call the device-free check
    assert "QC_CHECK_ABI()" in open(os.path.join(ROOT, "include", "qc_balance_controller.hpp")).read()
    assert "qc_check_abi(ABI_VERSION" in open(os.path.join(ROOT, "quadruped_control_amd", "_lib.py")).read()


def test_build_does_not_need_the_oracle(built, tmp_path, monkeypatch):
    """the product's build() must not fail because the test infrastructure is absent or broken -
    the oracle build is best-effort (a warning), the HIP library and the package import are not."""
    import importlib
    import warnings

    import __graft_entry__ as g

    importlib.reload(g)
    monkeypatch.setattr(g, "ORACLE_DIR", str(tmp_path / "no_oracle_here"))
    monkeypatch.setenv("QC_BUILD_REUSE", "1")  # (reuse the up-to-date in-tree library: this test is about the oracle leg)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        g.build()
    assert any("oracle" in str(x.message) for x in w), [str(x.message) for x in w]


def test_param_struct_layout_matches_c():
    from quadruped_control_amd import _lib

    assert ctypes.sizeof(_lib.QcParams) == 8 * (4 + 9 + 36 + 144 + 6 + 3 * 4) + 8
    assert ctypes.sizeof(_lib.QcBatchIn) == 18 * 8 and ctypes.sizeof(_lib.QcBatchOut) == 5 * 8
    assert ctypes.sizeof(_lib.QcKinematics) == 49 * 8
    assert ctypes.sizeof(_lib.QcSwingState) == 224
    assert ctypes.sizeof(_lib.QcLaunchInfo) == 4 * 4 + 4 * 8


def test_library_reads_no_environment():
    """Stray QC_* variables must not alter the controller: every override goes through qc_set_tuning."""
    src = open(os.path.join(ROOT, "quadruped_control_amd", "csrc", "qc_balance.hip")).read()
    assert "getenv" not in src


def test_no_gpu_fails_loudly(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import quadruped_control_amd as q

    with pytest.raises(RuntimeError, match="no HIP device"):
        q.BalanceController.from_params(q.cheetah_params())


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "quadruped_control_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle/", "") or f in ("balance_controller.py",), f
                assert "import oracle" not in txt and "from oracle" not in txt, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f)).read(), f


def test_forwarding_header_at_the_reference_include_path(built, tmp_path):
    """the reference's callers spell the class `#include <quadruped_controller/balance_controller.hpp>`
    (commander_node.cpp:34, gait_visualizer_node.cpp:31, test_node.cpp:17).  With -I <repo>/include that line must resolve to
    the GPU-backed adapter - same namespace, class, ctor and control() - without the caller editing anything.  Compile-only
    here (no GPU); tests/cpp/adapter_test.cpp includes the class the same way and runs on the GPU box."""
    import subprocess

    fwd = os.path.join(ROOT, "include", "quadruped_controller", "balance_controller.hpp")
    txt = open(fwd).read()
    assert "qc_balance_controller.hpp" in txt and "__has_include(<quadruped_controller/gait.hpp>)" in txt and "QC_USE_REFERENCE_TYPES" in txt
    assert os.listdir(os.path.dirname(fwd)) == ["balance_controller.hpp"]  # nothing here may shadow the caller's own gait.hpp / types.hpp
    src = tmp_path / "caller.cpp"
    src.write_text("""
#include <quadruped_controller/balance_controller.hpp>
#include <quadruped_controller/balance_controller.hpp>  // idempotent
using namespace quadruped_controller;
ForceMap tick(const BalanceController& bc, const mat& R, const vec& v, const FootholdMap& feet)
{ return bc.control(R, R, v, v, v, v, v, v, feet); }  // default gait_map = make_stance_gait(), balance_controller.hpp:107
ForceMap tick(const BalanceController& bc, const mat& R, const vec& v, const FootholdMap& feet, const GaitMap& gait)
{ return bc.control(R, R, v, v, v, v, v, v, feet, gait); }
real_t row_major[4];
void helpers(const mat& m, const vec& v) { copy_to_real_t(m, row_major); copy_to_real_t(v, row_major); (void)copy_from_real_t(row_major, 4); }
#ifndef BALANCE_CONTROLLER_HPP
#error "the reference's include guard is not defined: a second include of its own file would redefine the class"
#endif
""")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "#include <quadruped_controller/balance_controller.hpp>" in open(os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp")).read()


def test_forwarding_header_takes_the_callers_own_types(built, tmp_path):
    """the branch every real caller takes.  In a catkin build of the reference, `<quadruped_controller/gait.hpp>` (and
    through it types.hpp), `<quadruped_controller/math/rigid3d.hpp>` and `<armadillo>` are on the include path, so the forwarding header
    includes them and the adapter must use THEIR LegState / GaitMap / FootholdMap / ForceMap / make_stance_gait and arma's mat / vec / vec3
    (QC_USE_REFERENCE_TYPES, QC_HAVE_ARMADILLO) without redefining anything.  Neither the reference's tree nor Armadillo exists on the
    test boxes, so this test writes its OWN minimal stand-ins - declarations in the shapes of types.hpp:91-119, gait.hpp:21 and
    math/rigid3d.hpp:54, a dozen lines of arma-like vectors; none of it copied - into a temp dir placed BEHIND include/ on the path and
    compiles (syntax only) a caller shaped like commander_node.cpp:337-338, 507-512."""
    import subprocess

    inc = tmp_path / "their_tree"
    (inc / "quadruped_controller" / "math").mkdir(parents=True)
    (inc / "armadillo").write_text("""
#pragma once
#include <cstddef>
#include <initializer_list>
#include <vector>
#define QC_TEST_STANDIN_ARMADILLO 1
namespace arma {
typedef unsigned long long uword;
class vec {
public:
  vec() : n_rows(0), n_elem(0) {}
  explicit vec(uword n) : n_rows(n), n_elem(n), d_(n, 0.0) {}
  vec(std::initializer_list<double> l) : n_rows(l.size()), n_elem(l.size()), d_(l) {}
  double& operator()(uword i) { return d_[i]; }
  const double& operator()(uword i) const { return d_[i]; }
  uword size() const { return n_elem; }
  uword n_rows, n_elem;
private:
  std::vector<double> d_;
};
class vec3 : public vec {
public:
  vec3() : vec(3) {}
  vec3(std::initializer_list<double> l) : vec(l) {}
};
class mat {
public:
  mat() : n_rows(0), n_cols(0) {}
  mat(uword r, uword c) : n_rows(r), n_cols(c), d_(r * c, 0.0) {}
  double& operator()(uword i, uword j) { return d_[j * n_rows + i]; }
  const double& operator()(uword i, uword j) const { return d_[j * n_rows + i]; }
  uword n_rows, n_cols;
private:
  std::vector<double> d_;
};
typedef mat mat33;
}  // namespace arma
""")
    (inc / "quadruped_controller" / "types.hpp").write_text("""
#pragma once
#include <map>
#include <string>
#include <utility>
#include <armadillo>
#define QC_TEST_STANDIN_TYPES 1
namespace quadruped_controller {
using arma::mat;
using arma::vec;
using arma::vec3;
enum LegState { swing = 0, stance = 1 };
typedef std::map<std::string, std::pair<LegState, double>> GaitMap;
typedef std::map<std::string, vec3> FootholdMap;
typedef std::map<std::string, vec3> ForceMap;
typedef std::map<std::string, vec3> TorqueMap;
}  // namespace quadruped_controller
""")
    (inc / "quadruped_controller" / "gait.hpp").write_text("""
#pragma once
#include <quadruped_controller/types.hpp>
#define QC_TEST_STANDIN_GAIT 1
namespace quadruped_controller {
GaitMap make_stance_gait();  // declared only, as in the caller's tree (defined in its gait.cpp)
}
""")
    (inc / "quadruped_controller" / "math" / "rigid3d.hpp").write_text("""
#pragma once
#include <quadruped_controller/types.hpp>
#define QC_TEST_STANDIN_RIGID3D 1
namespace quadruped_controller { namespace math {
class Quaternion {
public:
  Quaternion() {}
  explicit Quaternion(const mat&) {}
  mat matrix() const { return mat(3, 3); }
};
} }
""")
    src = tmp_path / "commander_shaped.cpp"
    src.write_text("""
#include <string>
#include <type_traits>
#include <vector>
#include <quadruped_controller/balance_controller.hpp>
#if !defined(QC_USE_REFERENCE_TYPES) || !defined(QC_HAVE_ARMADILLO)
#error "the caller's gait.hpp / armadillo are on the path: the adapter must have selected them"
#endif
#if !defined(QC_TEST_STANDIN_TYPES) || !defined(QC_TEST_STANDIN_GAIT) || !defined(QC_TEST_STANDIN_RIGID3D) || !defined(QC_TEST_STANDIN_ARMADILLO)
#error "the forwarding header did not pull in the caller's own neighbours (balance_controller.hpp:13-14)"
#endif
using namespace quadruped_controller;
static_assert(std::is_same<mat, arma::mat>::value && std::is_same<vec3, arma::vec3>::value, "Armadillo's types at the boundary");
static_assert(std::is_same<ForceMap, std::map<std::string, arma::vec3>>::value, "the caller's ForceMap");
static_assert(std::is_same<GaitMap, std::map<std::string, std::pair<LegState, double>>>::value, "the caller's GaitMap and LegState");
static_assert(std::is_same<decltype(make_stance_gait()), GaitMap>::value, "the caller's make_stance_gait");
// commander_node.cpp:337-338 (construct once, const) and :507-512 (every tick; the result feeds the stance legs' torque map)
TorqueMap tick(const mat& Ib, const mat& S, const mat& W, const vec& k6, const vec& k3, const std::vector<std::string>& leg_names,
               const mat& Rwb, const vec3& x, const FootholdMap& foot_actual_map, const GaitMap& gait_map)
{
  const BalanceController balance_controller(0.8, 11.0, 10.0, 160.0, Ib, S, W, k6, k3, k3, k3, k3, leg_names);
  const math::Quaternion q(Rwb);  // (commander_node.cpp:173, 548 reach math:: through this header)
  const mat Rwb_d = q.matrix();
  const ForceMap force_map = balance_controller.control(Rwb, Rwb_d, x, x, x, x, x, x, foot_actual_map, gait_map);
  const ForceMap standing = balance_controller.control(Rwb, Rwb_d, x, x, x, x, x, x, foot_actual_map);  // default: make_stance_gait()
  TorqueMap torque_map;
  for (const auto& kv : force_map) torque_map.emplace(kv.first, kv.second);
  for (const auto& kv : standing) if (gait_map.at(kv.first).first == LegState::stance) torque_map.emplace(kv.first, kv.second);
  return torque_map;
}
""")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", str(inc), str(src)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # ... and the same caller WITHOUT their tree still compiles against the adapter's own types (the other branch, as before)
    src2 = tmp_path / "standalone.cpp"
    src2.write_text(src.read_text().split("using namespace quadruped_controller;")[0].split("#if !defined(QC_USE_REFERENCE_TYPES)")[0] +
                    "#ifdef QC_USE_REFERENCE_TYPES\n#error \"no reference tree on the path\"\n#endif\n")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src2)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_gait_rule():
    from quadruped_control_amd import LegState, leg_state_from_phase, make_stance_gait, stance_phase

    g = make_stance_gait()
    assert list(g) == ["RL", "FL", "RR", "FR"] and all(v == (LegState.stance, 0.0) for v in g.values())
    sp = stance_phase(0.18, 0.8)
    assert abs(sp - 0.8 / 0.98) < 1e-15
    ph = np.array([0.0, sp, sp + 1e-13, sp + 1e-9, 0.999, -1e-13])
    assert leg_state_from_phase(ph, sp).tolist() == [1, 1, 1, 0, 0, 1]


def test_workloads_deterministic_and_shardable():
    from quadruped_control_amd import workloads as W
    from quadruped_control_amd.sharding import shard_bounds

    full = W.config5(n=1000, start=0)
    parts = []
    for r in range(3):
        lo, hi = shard_bounds(1000, r, 3)
        parts.append(W.config5(n=hi - lo, start=lo))
    for k in full:
        np.testing.assert_array_equal(full[k], np.concatenate([p[k] for p in parts]))
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    # config 3 produces the 2/3/4-foot mix and rotations are orthonormal
    hist = np.bincount(full["stance"].sum(1), minlength=5)
    assert hist[2] > 0 and hist[3] > 0 and hist[4] > 0 and hist[:2].sum() == 0
    Rm = full["Rwb"].reshape(-1, 3, 3)
    np.testing.assert_allclose(Rm @ Rm.transpose(0, 2, 1), np.tile(np.eye(3), (1000, 1, 1)), atol=1e-12)
    t0, t1 = W.config4(16)
    assert np.all(t0["stance"] == 1) and not np.array_equal(t0["x"], t1["x"])
    c1 = W.config1()
    np.testing.assert_allclose(c1["feet"].reshape(4, 3)[:, 2], -0.26)


def test_bench_gpus_flag_is_not_ignored():
    """`bench.py --gpus N` used to be parsed and ignored.  Under a launcher that made a different world
    it must fail; without a launcher it must start N ranks itself - or say why it cannot (no GPUs here)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QC_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match WORLD_SIZE=1" in (r.stderr + r.stdout)
    import torch

    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "--gpus 2 but only" in (r.stderr + r.stdout)


def test_bench_parses_rccl_transport_log():
    """`ranks.rccl` of an N > 1 bench line is parsed from RCCL's own INIT / GRAPH log.  No multi-GPU box has ever run it, so
    the parser is held here against lines in the format RCCL 2.26 prints (the one-rank log of the 1-GPU box has the INIT and Pattern lines;
    the Channel / via lines follow NCCL's documented `Channel cc/n : a[dev] -> b[dev] [receive] via P2P/IPC[/read]` shape): eight ranks over
    peer-to-peer links, a mixed P2P + SHM box, an empty log, and text it has never seen (kept verbatim, classified "unknown")."""
    import bench

    pre = "gpu-node:4242:4311 [3] NCCL INFO "
    lines = [pre + "ncclCommInitRankConfig_impl comm 0x55d0 rank 3 nranks 8 cudaDev 3 nvmlDev 3 busId 9d000 commId 0x7a - Init START",
             pre + "Pattern 4, crossNic 0, nChannels 28, bw 48.000000/48.000000, type XGMI/PIX, sameChannels 1",
             pre + "Pattern 3, crossNic 0, nChannels 28, bw 48.000000/48.000000, type XGMI/PIX, sameChannels 1",
             pre + "Ring 00 : 2 -> 3 -> 4", pre + "Tree 0 : -1 -> 3 -> 4/-1/-1"]
    for c in range(28):
        lines.append(pre + "Channel %02d/0 : 3[3] -> 4[4] via P2P/IPC" % c)
        lines.append(pre + "Channel %02d/0 : 2[2] -> 3[3] [receive] via P2P/IPC/read" % c)
    lines.append(pre + "Connected all rings")
    r = bench.parse_rccl_log("\n".join(lines), 8)
    assert r["nranks"] == 8 and r["channels"] == 28 and r["via"] == {"P2P/IPC": 28, "P2P/IPC/read": 28}
    assert r["transport"] == "P2P (xGMI / peer access)" and r["graph"]["pattern4"] == {"nChannels": 28, "bw": 48.0, "type": "XGMI/PIX"}
    assert any("nranks 8" in l for l in r["lines"]) and any("via P2P/IPC" in l for l in r["lines"]) and len(r["lines"]) <= 24
    mixed = bench.parse_rccl_log("\n".join(lines + [pre + "Channel 00/0 : 3[3] -> 0[0] via SHM/direct/direct"]), 8)
    assert mixed["transport"] == "P2P/IPC+P2P/IPC/read+SHM/direct/direct" and mixed["via"]["SHM/direct/direct"] == 1
    assert bench.parse_rccl_log("", 8)["transport"] == "unknown (no channel lines in the log)" and bench.parse_rccl_log("", 1)["transport"] == "none (one rank)"
    odd = bench.parse_rccl_log(pre + "comm 0x1 rank 0 nranks 2 something new\n" + pre + "links are XGMI today", 2)
    assert odd["nranks"] == 2 and odd["channels"] is None and len(odd["lines"]) == 2 and odd["transport"].startswith("unknown")


def test_device_generator_matches_the_host_generator_on_cpu():
    """workloads_device (torch int64 splitmix64) against workloads (numpy uint64): the uniforms and everything decided
    by them (gait kind, phases, contact states) bit-identical, the rotation matrices / foot positions to the last ulp."""
    import torch

    from quadruped_control_amd import workloads as W
    from quadruped_control_amd import workloads_device as D

    idx = np.arange(7, 7 + 4096, dtype=np.uint64)
    for seed, stream in ((W.SEEDS[5], 0), (W.SEEDS[3], 21), (W.SEEDS[5] + 0x100, 28)):
        assert np.array_equal(W.uniform(seed, idx, stream), D.uniform(seed, torch.arange(7, 7 + 4096, dtype=torch.int64), stream).numpy())
    h = W.config5(3000, start=786432)
    d = D.config5(3000, start=786432, device="cpu")
    assert set(h) == set(d)
    for k in h:
        dd = d[k].numpy()
        assert dd.dtype == h[k].dtype and dd.shape == h[k].shape and dd.flags["C_CONTIGUOUS"]
        if k in ("Rwb", "feet"):
            assert np.max(np.abs(dd - h[k])) < 1e-15
        else:
            assert np.array_equal(dd, h[k]), k


def test_bench_tick_workloads_and_byte_counts():
    """bench.py's SURVEY 8(f) tick workloads: the algorithmic byte counts the roofline is priced with,
    the number of rotating sets of the cold-cache protocol, and the shape of the generated batches."""
    import bench

    assert bench.bytes_per_robot(False) == 488 == 48 * 8 + 4 + 12 * 8 + 4
    assert bench.bytes_per_robot(True) == 496
    assert bench.bytes_per_robot(False, True) == 584 == 488 + 96                      # + joint_tau
    assert bench.bytes_per_robot(False, "full-frozen") == 964 == (384 + 32 + 96 + 224) + (96 + 4 + 96 + 32)
    # the clocked complete tick: + gait_dt read + advanced phases written + 48 B per stance -> swing edge
    assert bench.bytes_per_robot(False, "full") == 964 + 8 + 32 == 1004
    assert abs(bench.bytes_per_robot(False, "full", edge_legs=4.0 / 294.0) - (1004 + 48 * 4.0 / 294.0)) < 1e-12
    assert bench.rotation_sets(4096, False) == 269 and bench.rotation_sets(65536, False) == 17
    assert bench.rotation_sets(65536, False, "full-frozen") == (512 << 20) // (964 * 65536) + 1 == 9 == bench.rotation_sets(65536, False, "full")
    assert bench.rotation_sets(262144, False, "full") == 3 == bench.rotation_sets(262144, False, "full-frozen") and bench.rotation_sets(2097152, False) == 1
    assert isinstance(bench.rotation_sets(262144, False, "full"), int)
    fused = bench.make_tick_batch(2, 64, 0, True)
    assert "feet" not in fused and fused["joint_q"].shape == (64, 12) and fused["stance"].shape == (64, 4)
    full = bench.make_tick_batch(3, 64, 128, "full", j=2)
    assert {"joint_q", "joint_qdot", "gait_phase"} <= set(full) and not {"stance", "swing_pos", "swing_vel", "feet"} & set(full)
    assert full["gait_phase"].shape == (64, 4) and (full["gait_phase"] >= 0).all() and (full["gait_phase"] < 1).all()
    assert full["gait_dt"].shape == (64,) and (full["gait_dt"] == 1.0 / 300.0).all()  # mit_cheetah_config.yaml:3
    frozen = bench.make_tick_batch(3, 64, 128, "full-frozen", j=2)
    assert "gait_dt" not in frozen and set(frozen) | {"gait_dt"} == set(full)
    np.testing.assert_array_equal(frozen["gait_phase"], full["gait_phase"])
    # rotation sets hold different robots of the same distribution; a shard is reproducible from its start index
    other = bench.make_tick_batch(3, 64, 128, "full", j=0)
    assert not np.array_equal(full["joint_q"], other["joint_q"])
    np.testing.assert_array_equal(bench.make_tick_batch(3, 32, 160, "full", j=2)["joint_q"], full["joint_q"][32:])
    assert "EPYC" in bench.cpu_model() or len(bench.cpu_model()) > 3
    assert bench.oracle_build_flags().startswith(("gcc", "cc")) and "-ffp-contract=off" in bench.oracle_build_flags()


def test_bench_counts_the_gait_edges_its_timed_region_contains(built):
    """the complete-tick entries advance the on-device gait clock inside the timed region and report the
    realised stance -> swing edge rate (each edge = one FootPlanner::singleFoot + trajectory reset on the device).  bench.py
    counts the edges by replaying the clock on a snapshot; here that replay is checked against the oracle's
    GaitScheduler::update + the package's contact rule, tick by tick, on sets that run different numbers of ticks."""
    import torch

    import bench
    from oracle import c_oracle as O
    from quadruped_control_amd import leg_state_from_phase, stance_phase

    n, per_set = 500, [7, 12, 0]
    rng = np.random.default_rng(3)
    ph0 = np.ascontiguousarray(np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + rng.uniform(0, 1, (3 * n, 1)), 1.0))
    kin = O.default_kinematics()
    assert (kin.t_swing, kin.t_stance) == (bench.GAIT_T_SWING, bench.GAIT_T_STANCE)
    duty = stance_phase(kin.t_swing, kin.t_stance)
    ref = ph0.copy()
    edges = robots_edge = 0
    for j, c in enumerate(per_set):
        ph = np.ascontiguousarray(ref[j * n:(j + 1) * n])
        st = leg_state_from_phase(ph, duty).astype(bool)
        for _ in range(c):
            O.gait_update(ph, np.full(n, bench.GAIT_DT), kin=kin)
            st_new = leg_state_from_phase(ph, duty).astype(bool)
            e = st & ~st_new
            edges += int(e.sum()); robots_edge += int(e.any(axis=1).sum())
            st = st_new
        ref[j * n:(j + 1) * n] = ph
    got = bench.count_gait_edges(torch.from_numpy(ph0), per_set, n, torch.from_numpy(ref))
    assert got[0] == edges > 0 and got[1] == 4 * n * sum(per_set) and got[2] == robots_edge and got[3] == n * sum(per_set)
    assert got[4] == 0.0  # the replayed clock is bit-equal to the oracle's
    # natural rate: one edge per leg per gait period
    assert abs(edges / (n * sum(per_set)) - 4 * bench.GAIT_DT / 0.98) < 0.004


def test_development_scripts_parse():
    """tools/, tests/stress_*.py, tests/golden/*.py and oracle/prototypes are run by hand on the GPU box, never by pytest: at least
    every one of them must still parse (a rename in the package would otherwise rot them silently)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "stress_*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "golden", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "prototypes", "*.py")) +
                   [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
    assert len(files) > 60
    for f in files:
        compile(open(f).read(), f, "exec")  # (syntax only: nothing is imported or run, no bytecode is written)


def test_xcd_aware_chunk_map_is_a_bijection():
    """balance_kernel's racing waves (4 or 8 robots each) share the cache lines their rows straddle with their neighbours; 16 robots are
    line-aligned in every array.  The kernel maps workgroup b (dispatched to XCD b % 8) so that the 16 / chunk waves of an aligned group run on
    ONE XCD (one L2 fetches the shared lines once) while the groups go round the XCDs:
        chunk = (slot / gw) * 8 gw + xcd * gw + slot % gw,  xcd = b % 8, slot = b / 8, gw = 16 / chunk   (blocks beyond the last full 8 gw: identity).
    The map must hit every chunk exactly once for any grid size, and keep every aligned group on one XCD."""
    src = open(os.path.join(ROOT, "quadruped_control_amd", "csrc", "qc_balance.hip")).read()
    assert "chunk_id = (slot / gw) * super + xcd * gw + (slot % gw);" in src and "constexpr unsigned gw = (unsigned)RACE, super = 8u * gw;" in src
    for chunk in (4, 8):
        gw = 16 // chunk
        sup = 8 * gw
        for nwg in list(range(1, 100)) + [255, 256, 257, 1023, 1024, 1025, 4096, 4097]:
            b = np.arange(nwg)
            full = (nwg // sup) * sup
            xcd, slot = b & 7, b >> 3
            c = np.where(b < full, (slot // gw) * sup + xcd * gw + (slot % gw), b)
            assert np.array_equal(np.sort(c), b), (chunk, nwg)
            owner = np.empty(nwg, dtype=int); owner[c] = xcd
            for g in range(full // gw):  # every aligned group of gw chunks on one XCD
                assert len(set(owner[g * gw:(g + 1) * gw])) == 1
