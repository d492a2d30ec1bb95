"""Fuzz campaign (run() is what tests/test_gpu_fuzz.py calls with a time budget; as a script it runs the long version): wild robot STATES through the reference's own parameters - rotations up to
pi, large velocities and offsets, arbitrary contact patterns (incl. none), feet far from nominal - GPU vs C oracle.
usage: python tests/stress_fuzz_states.py [batches=40] [robots=4096]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from oracle import c_oracle as O



def run(batches=40, n=4096, budget_s=None, min_batches=4):
    """Returns (worst relative error over solved robots, status mismatches, batches done)."""
    rng = np.random.default_rng(int(os.environ.get("QC_FUZZ_SEED", 4242)))  # QC_FUZZ_SEED: another campaign (the default is the one pytest runs)
    worst = 0.0; mism = 0; stat = np.zeros(4, int); t0 = time.time()
    for bi in range(batches):
        if budget_s is not None and bi >= min_batches and time.time() - t0 > budget_s: bi -= 1; break
        mu = float(rng.choice([0.3, 0.6, 0.8]))
        P = q.cheetah_params(mu)
        amp = float(rng.choice([0.3, 1.0, 3.0, np.pi * 0.999]))
        rv = rng.uniform(-1, 1, (n, 3)); rv *= (rng.uniform(0, amp, (n, 1)) / np.linalg.norm(rv, axis=1, keepdims=True))
        rvd = rng.uniform(-1, 1, (n, 3)); rvd *= (rng.uniform(0, amp, (n, 1)) / np.linalg.norm(rvd, axis=1, keepdims=True))
        R = W.rotvec_to_matrix(rv); Rd = W.rotvec_to_matrix(rvd) if bi % 2 else np.tile(np.eye(3), (n, 1, 1))
        x = np.array([0, 0, 0.26]) + rng.uniform(-0.5, 0.5, (n, 3))
        xd = np.array([0, 0, 0.26]) + rng.uniform(-0.2, 0.2, (n, 3))
        vs = float(rng.choice([0.3, 2.0, 10.0]))
        b = dict(Rwb=R.reshape(n, 9), Rwb_d=Rd.reshape(n, 9), x=x, xdot=rng.uniform(-vs, vs, (n, 3)), w=rng.uniform(-vs, vs, (n, 3)),
                 x_d=xd, xdot_d=rng.uniform(-1, 1, (n, 3)), w_d=rng.uniform(-1, 1, (n, 3)),
                 feet=(np.tile(np.array([[-0.196, 0.127, -0.26], [0.196, 0.127, -0.26], [-0.196, -0.127, -0.26], [0.196, -0.127, -0.26]]), (n, 1, 1))
                       + rng.uniform(-0.15, 0.15, (n, 4, 3))).reshape(n, 12),
                 stance=(rng.uniform(0, 1, (n, 4)) < rng.choice([0.3, 0.7, 1.0])).astype(np.uint8))
        b = {k: np.ascontiguousarray(v) for k, v in b.items()}
        ctl = q.BalanceController.from_params(P)
        o = ctl.control_batch_host(b, want_iterations=True)
        ref, st, it = O.control_batch(P, b, threads=16)
        mism += int((o["status"] != st).sum())
        stat += np.bincount(o["status"], minlength=4)[:4]
        okm = (o["status"] == 0) & (st == 0)
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
        e = (np.abs(o["grf_body"] - ref) / scale).max(axis=1)
        err = float(e[okm].max()) if okm.any() else 0.0
        err = err if err == err else float("inf")  # (a NaN force of a "solved" robot must not hide behind nan > tol == False)
        worst = max(worst, err)
        if err > 1e-6 or (o["status"] != st).any():
            i = int(np.argmax(np.where(okm, e, 0)))
            print("batch", bi, "amp %.2f vs %.1f mu %.1f: err %.2e (robot %d, stance %s), status mismatches %d; gpu iters max %d" %
                  (amp, vs, mu, err, i, b["stance"][i], int((o["status"] != st).sum()), o["iterations"].max()))
    print("%d batches x %d robots in %.0f s: worst rel err %.2e, status mismatches %d, gpu status histogram %s" % (bi + 1, n, time.time() - t0, worst, mism, stat))
    return worst, mism, bi + 1


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
