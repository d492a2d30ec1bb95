import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree native artefacts exist (no-op when prebuilt)."""
    import __graft_entry__ as g

    g.build(force=False)
    return g


@pytest.fixture(autouse=True)
def _oracle_pinv_window_unvisited():
    """the oracle's swing-leg pseudo-inverse follows arma::pinv's own rank rule (the reference), the device its
    complete-pivoting rule (a documented deviation).  The oracle counts every swing leg on which the two would keep a different
    rank; no test may leave that count non-zero - a comparison made inside that window would not be evidence about the
    reference.  (Tests that plant such matrices on purpose reset the counter themselves.)"""
    yield
    mod = sys.modules.get("oracle.c_oracle")
    if mod is not None and getattr(mod, "_lib", None) is not None:  # only if this process has the checker loaded
        assert mod.pinv_rule_disagreements(reset=True) == 0, "a swing leg sat where arma::pinv's rank rule and the device's differ"
