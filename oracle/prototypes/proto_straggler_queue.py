"""Round-3 design prototype (development tool, uses the oracle): what would a straggler QUEUE buy?  Robots of a
one-lane wave that still run after the clamp steps go to a list shared by K consecutive waves and are finished four
lanes per robot, 16 at a time, with lane groups refilled from the list as robots finish - instead of every wave
finishing its own <= 16 stragglers.  Counts wave-recalculations (1118 instructions on the one-lane body, 488 on the
4-lane body) on config 5's recalculation counts under the product's strategy (five clamp steps, drop the most negative
multiplier).  Result (65 536 robots): tail recalculations per wave 5.65 -> 2.3-3.8, -15 ... -22 % wave-instructions in
the solve for K = 4 ... 16 - and a consumer chain of 11-27 recalculations per list, which is what the built kernel
(tools/experiments/straggler_queue.patch, profiles/r03_queue_scan.log) then paid for at the end of the launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W
from oracle.prototypes.prototype_solver import assemble_batch
from oracle.prototypes.prototype_as import QP
from oracle.prototypes.proto_race_strategies_lib import solve_policy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P = R.cheetah_params(mu=0.6)
B = W.config5(n)
Q, c = assemble_batch(P, B)
base = np.array([solve_policy(QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"]), lambda it: it < 5, lambda it: "most")[1] for i in range(n)])
print("recalculations: mean %.2f max %d" % (base.mean(), base.max()))
Wv, C1, C4, FIRST = n // 64, 1118.0, 488.0, 5


def split(th):
    main, recs = 0, []
    for w in range(Wv):
        x, t = base[w * 64:(w + 1) * 64], 0
        while (x > t).any() and (t < FIRST or (x > t).sum() > th):
            t += 1
        main += t
        recs.append(x[x > t] - t)
    return main, recs


def consume(recs, K, refill=4):
    total, chains = 0, []
    for g in range(0, len(recs), K):
        lst, pos, slots, it = list(np.concatenate(recs[g:g + K])), 0, [], 0
        while len(slots) < 16 and pos < len(lst):
            slots.append(lst[pos]); pos += 1
        while slots:
            it += 1
            slots = [s - 1 for s in slots if s > 1]
            if pos < len(lst) and (16 - len(slots) >= refill or not slots):
                while len(slots) < 16 and pos < len(lst):
                    slots.append(lst[pos]); pos += 1
        total += it; chains.append(it)
    return total, np.array(chains)


m16, r16 = split(16)
cur = C1 * m16 + C4 * sum((r.max() if len(r) else 0) for r in r16)
print("product (own tail, hand-over at 16): main %.2f + tail %.2f recalculations per wave, %.0f instructions" % (m16 / Wv, sum((r.max() if len(r) else 0) for r in r16) / Wv, cur / Wv))
for th in (16, 24, 32):
    m, recs = split(th)
    for K in (1, 2, 4, 8, 16):
        it, ch = consume(recs, K)
        tot = C1 * m + C4 * it
        print("hand-over at %2d, K = %2d: main %.2f, %.1f records and %.2f tail recalculations per wave, consumer chain mean %.1f max %d -> %.0f instructions (%.1f %%)"
              % (th, K, m / Wv, sum(len(r) for r in recs) / Wv, it / Wv, ch.mean(), ch.max(), tot / Wv, 100 * tot / cur))
