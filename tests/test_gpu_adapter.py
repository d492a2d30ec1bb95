"""-m gpu: the C++ BalanceController adapter (include/qc_balance_controller.hpp)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_cpp_adapter(built):
    exe = os.path.join(os.path.dirname(__file__), "cpp", "adapter_test")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    for tag in ("OK kat1", "OK kat3", "OK out_of_range", "OK size_check\n", "OK size_check_mat", "OK copy_to_real_t", "latency_us"):
        assert tag in r.stdout, r.stdout + r.stderr
    lat = float(r.stdout.split("latency_us")[1].split()[0])
    assert 1.0 < lat < 2000.0, r.stdout  # one robot per call, launch + in-place pinned record + synchronise


def test_cpp_batch_caller(built):
    """A C++ program on the bare C ABI (tests/cpp/batch_test.cpp): QC_CHECK_ABI(), the five KATs and 600 generated robots
    through qc_control_batch_host AND qc_control_batch on hipMalloc'ed arrays (its own stream), one robot per call through
    qc_control, and tick t's active set fed back as tick t+1's warm start (commander_node.cpp:337-338, 507-512 as a batch)."""
    exe = os.path.join(os.path.dirname(__file__), "cpp", "batch_test")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout, r.stdout
    for tag in ("OK abi", "OK kat1", "OK kat2", "OK kat3", "OK kat4", "OK kat5", "OK batch_host_rc", "OK batch_device_rc", "OK batch_all_solved",
                "OK batch_cone_and_swing", "OK batch_host_equals_device", "OK single_equals_batch", "OK warm_start_round_trip",
                "OK null_argument_refused", "PASSED"):
        assert tag in r.stdout, r.stdout + r.stderr
