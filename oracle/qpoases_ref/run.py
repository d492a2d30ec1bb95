"""TEST INFRASTRUCTURE (optional): literal-reference compare / timing hook.

Only does something on a host where qpOASES itself is installed, found the way the reference finds it
(cmake/FindqpOASES.cmake:34-45): $qpOASES_SOURCE_DIR/include/qpOASES.hpp and libqpOASES in
$qpOASES_BINARY_DIR/{lib,libs}.  Otherwise it prints "skipped" and exits 0 - nothing here emulates qpOASES.

  python oracle/qpoases_ref/run.py golden      -> tests/golden/qpoases_golden.json: the committed golden inputs and
                                                  2048 config-2 / config-3 robots solved by qpoases_ref.cpp (the
                                                  reference's init -> hotstart sequence on the oracle's H, g, C, lbC,
                                                  ubC), world-frame forces + status.  tests/test_qpoases_golden.py
                                                  then pins the oracle (and the GPU path) to it.
  python oracle/qpoases_ref/run.py time [n]    -> QPs/s of the literal sequence (nWSR 200, cputime 0.01) on n
                                                  config-2 robots, one thread; bench.py reports it as
                                                  cpu_baseline.kind = "reference" when the binary exists.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
EXE = os.path.join(REF, "qpoases_ref")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")


def find_qpoases():
    src, binr = os.environ.get("qpOASES_SOURCE_DIR"), os.environ.get("qpOASES_BINARY_DIR")
    if not src or not binr:
        return None
    inc = os.path.join(src, "include")
    if not os.path.exists(os.path.join(inc, "qpOASES.hpp")):
        return None
    for sub in ("lib", "libs", ""):
        for name in ("libqpOASES.so", "libqpOASES.a"):
            lib = os.path.join(binr, sub, name)
            if os.path.exists(lib):
                return inc, lib
    return None


def build():
    found = find_qpoases()
    if found is None:
        return None
    inc, lib = found
    os.makedirs(REF, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++11", "-D__AVOID_LA_NAMING_CONFLICTS__", "-I", inc, os.path.join(HERE, "qpoases_ref.cpp"), "-o", EXE]
    cmd += [lib] if lib.endswith(".a") else ["-L", os.path.dirname(lib), "-lqpOASES", "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.check_call(cmd)
    return EXE


def write_problems(P, batch, path):
    """QP data of every robot as the reference hands it to qpOASES (oracle_assemble = balance_controller.cpp:98-163, 294-330)."""
    import numpy as np

    from oracle import c_oracle as O

    n = batch["x"].shape[0]
    with open(path, "w") as f:
        f.write(f"{n}\n")
        for i in range(n):
            qp = O.assemble(P, *[batch[k][i] for k in FIELDS], batch["stance"][i])
            row = np.concatenate([qp["H"].reshape(-1), qp["g"], qp["C"].reshape(-1), qp["lb"], qp["ub"]])
            f.write(" ".join(repr(float(v)) for v in row) + "\n")


def run_exe(problems, results, *flags):
    subprocess.check_call([EXE, problems, results, *flags])
    rows, secs = [], None
    for line in open(results):
        t = line.split()
        if t[0] == "seconds_per_pass":
            secs = float(t[1])
        else:
            rows.append((int(t[0]), int(t[1]), float(t[2]), [float(v) for v in t[3:15]]))
    return rows, secs


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "golden"
    if build() is None:
        print("skipped: qpOASES not found (set qpOASES_SOURCE_DIR and qpOASES_BINARY_DIR as cmake/FindqpOASES.cmake expects)")
        return 0
    import numpy as np

    from oracle import numpy_restatement as R
    from quadruped_control_amd import workloads as W

    tmp = os.path.join(REF, "problems.txt")
    res = os.path.join(REF, "results.txt")
    if mode == "time":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
        P = R.cheetah_params(0.6)
        write_problems(P, W.config2(n), tmp)
        rows, secs = run_exe(tmp, res, "--repeat", "3")
        print(json.dumps({"value": n / secs, "unit": "QPs/s", "cores": 1, "kind": "reference",
                          "sample": f"{n} config-2 robots, one SQProblem: init then hotstart (nWSR 200, cputime 0.01 s), best-effort single pass",
                          "failed": sum(1 for r in rows if r[0] != 0)}))
        return 0
    cases = []
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "balance_golden.json")))
    groups = {}
    for c in gold["cases"]:
        groups.setdefault((c["mu"], c["fzmin"], c["fzmax"]), []).append(c)
    sets = []
    for (mu, fzmin, fzmax), cs in groups.items():
        P = R.cheetah_params(mu)
        P["fzmin"], P["fzmax"] = fzmin, fzmax
        b = {k: np.array([c[k] for c in cs], dtype=np.float64) for k in FIELDS}
        b["stance"] = np.array([c["stance"] for c in cs], dtype=np.uint8)
        sets.append((f"golden mu={mu} fz=[{fzmin},{fzmax}]", P, b))
    sets.append(("config2[0:2048]", R.cheetah_params(0.6), W.config2(2048)))
    sets.append(("config3[0:2048]", R.cheetah_params(0.6), W.config3(2048)))
    for name, P, b in sets:
        write_problems(P, b, tmp)
        for seq, flags in (("hotstart-sequence", ["--no-cputime-cap"]), ("fresh-init", ["--fresh", "--no-cputime-cap"])):
            rows, _ = run_exe(tmp, res, *flags)
            cases.append({"set": name, "sequence": seq, "mu": P["mu"], "fzmin": P["fzmin"], "fzmax": P["fzmax"],
                          "inputs": {k: b[k].tolist() for k in FIELDS + ("stance",)},
                          "status": [r[0] for r in rows], "nWSR": [r[1] for r in rows], "f_world": [r[3] for r in rows]})
    out = os.path.join(ROOT, "tests", "golden", "qpoases_golden.json")
    json.dump({"generator": "oracle/qpoases_ref/run.py golden", "qpoases": os.environ.get("qpOASES_SOURCE_DIR"), "cases": cases}, open(out, "w"))
    print("wrote", out, "with", len(cases), "case sets")
    return 0


if __name__ == "__main__":
    sys.exit(main())
