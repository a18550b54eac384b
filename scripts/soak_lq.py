"""Randomised soak of the generic (LQ) path against the oracle: random dimensions nx <= 32,
nu <= 16, batch, horizon, dense or diagonal weights; a few fixed-work iterations end to end.
    python scripts/soak_lq.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from oracle import oracle as O
from scripts.lq_case import DT, classify, gpu, make_case



def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_traj = n_moved = 0
    while time.time() < t_end:
        cs = int(rng.integers(1, 2**31 - 1))
        c = make_case(cs)
        om, x0, u0, B, iters = c["om"], c["x0"], c["u0"], c["B"], c["iters"]
        g = gpu(c)
        c0 = g.init_traj(x0, u0)
        _, _, c_o = O.batch_rollout(om, x0, u0, DT)
        if np.max(np.abs(c0 - c_o) / np.maximum(np.abs(c_o), 1e-300)) > 1e-12:
            print("FAIL initial cost:", c["desc"])
            return 1
        g.iterate(iters)
        ro = O.batch_solve(om, x0, u0, DT, max_iters=iters, fixed_work=True)
        cost = g.cost()
        g.close()
        if not np.all(np.isfinite(cost)):
            print("FAIL non-finite cost:", c["desc"])
            return 1
        rel = np.abs(cost - ro["cost"]) / np.maximum(np.abs(ro["cost"]), 1e-300)
        for bb in np.flatnonzero(rel >= 1e-6):
            # a deviating trajectory must trace back to a clamp knife edge of one backward pass
            if not classify(c, int(bb)):
                print("FAIL cost parity:", c["desc"], "trajectory", bb, "rel", rel[bb], " (python scripts/lq_case.py %d %d)" % (cs, bb))
                return 1
            n_moved += 1
        n_cases += 1
        n_traj += B
    print("lq soak ok: %d cases, %d trajectories, %d that deviate from a clamp knife edge on (classified by scripts/lq_case.py), seed %d"
          % (n_cases, n_traj, n_moved, seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
