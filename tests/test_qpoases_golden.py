"""Parity pins from the LITERAL reference solver (optional fixture).

tests/golden/qpoases_golden.json exists only after oracle/qpoases_ref/run.py has run on a host that has qpOASES
itself (the reference's `SQProblem::init` -> `hotstart` sequence, balance_controller.cpp:165-216, on the oracle's
QP data).  When it is there, the C oracle - and on a GPU box the HIP path - must reproduce its forces; when it is
not, these tests skip: the oracle then stays "parity unpinned" for the QP."""
import json
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(__file__), "golden", "qpoases_golden.json")
FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")


def _cases():
    if not os.path.exists(FIX):
        pytest.skip("no qpOASES fixture: oracle/qpoases_ref/run.py has not run on a host with qpOASES")
    return json.load(open(FIX))["cases"]


def _params(c):
    from oracle import numpy_restatement as R

    P = R.cheetah_params(c["mu"])
    P["fzmin"], P["fzmax"] = c["fzmin"], c["fzmax"]
    return P


def _batch(c):
    b = {k: np.array(c["inputs"][k], dtype=np.float64) for k in FIELDS}
    b["stance"] = np.array(c["inputs"]["stance"], dtype=np.uint8)
    return b


def _world(b, grf):
    R = b["Rwb"].reshape(-1, 3, 3)
    return (-np.einsum("nij,nkj->nki", R, grf.reshape(-1, 4, 3))).reshape(-1, 12)


def test_oracle_matches_qpoases():
    from oracle import c_oracle as O

    for c in _cases():
        b = _batch(c)
        ok = np.array(c["status"]) == 0
        assert ok.mean() > 0.99, c["set"]
        grf, st, _ = O.control_batch(_params(c), b)
        fw, ref = _world(b, grf)[ok], np.array(c["f_world"])[ok]
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
        assert np.max(np.abs(fw - ref) / scale) < 1e-6, (c["set"], c["sequence"])


@pytest.mark.gpu
def test_gpu_matches_qpoases(built):
    import quadruped_control_amd as q

    for c in _cases():
        b = _batch(c)
        ok = np.array(c["status"]) == 0
        o = q.BalanceController.from_params(_params(c)).control_batch_host(b)
        fw, ref = _world(b, o["grf_body"])[ok], np.array(c["f_world"])[ok]
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
        assert (o["status"][ok] == 0).all()
        assert np.max(np.abs(fw - ref) / scale) < 1e-4, (c["set"], c["sequence"])  # north_star's bar
