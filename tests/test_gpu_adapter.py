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
