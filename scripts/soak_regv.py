"""Randomised soak of ILQR_FLAG_REGULARIZE_VXX on the generic backward kernel (k_backward_w3<.., REGV>; run on the GPU box).

Every case draws dimensions n <= 32, m <= 16, a batch, a horizon, limits and lambda > 0, and checks the teacher-forced backward pass of a
host-model handle (the backward pass is all it runs on the device) against the oracle's with the same switch (orc_set_fixes(4)) per knot
(tests/parity.check_backward: 1e-6; deviations must be clamp knife edges or fp64-conditioning-limited against the fp80 yardstick), and that
the flag changes the gains (it is another regularisation than lambda I on Quu).

    python scripts/soak_regv.py [seconds] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from oracle import oracle as O
from tests.parity import check_backward
from tests.util import mat

DT = 0.02


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_traj = n_same = 0
    O.set_fixes(4)
    try:
        while time.time() < t_end:
            n = int(rng.integers(2, 33))
            m = int(rng.integers(1, min(16, n) + 1)) if rng.random() < 0.8 else int(rng.integers(1, 17))
            B = int(rng.choice([1, 3, 8, 20]))
            T = int(rng.choice([1, 2, 5, 12, 30]))
            lim = float(rng.choice([0.1, 0.3, 1.0, 5.0]))
            lam = float(rng.choice([1e-3, 1.0, 10.0]))
            A = -np.eye(n) + 0.1 * rng.normal(size=(n, n)) / np.sqrt(n)
            Bm = rng.normal(size=(n, m)) / np.sqrt(n)
            om = O.Model("lq", lq=(A, Bm, np.eye(n), 0.1 * np.eye(m), np.eye(n)), u_lim=lim)
            x0 = rng.uniform(-1, 1, (B, n))
            u0 = rng.normal(size=(B, T, m)) * 0.5
            xs, us, cost = O.batch_rollout(om, x0, u0, DT)
            dv = O.batch_derivatives(om, xs, us, DT)
            k_prev = rng.normal(size=(B, T, m)) * 0.1
            desc = "n=%d m=%d B=%d T=%d lim=%g lam=%g seed=%d" % (n, m, B, T, lim, lam, seed)
            ro = O.batch_backward(om, us, dv, k_prev=k_prev, lam=lam)
            g = BatchILQR("host", B, T, DT, nx=n, nu=m, u_min=om.u_min, u_max=om.u_max, flags=capi.FLAG_REGULARIZE_VXX)
            g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
            g.set_derivatives(**{k: (dv[k] if k in ("cx", "cu") else mat(dv[k])) for k in dv})
            g.set_gains(k=k_prev, K=np.zeros((B, T, m, n)))
            g.set_lambda(lam, 1.0)
            div = np.asarray(g.backward_pass())
            k, K = g.gains()
            dV = g.dV()
            g.close()
            try:
                check_backward(O, om, us, dv, k_prev, lam, k, K, dV, div, ro, max_ties=max(1, B // 4), max_over10=max(1, B // 4))
            except AssertionError as e:
                print("FAIL against the oracle:", desc, str(e)[:400])
                return 1
            O.set_fixes(0)
            r0 = O.batch_backward(om, us, dv, k_prev=k_prev, lam=lam)
            O.set_fixes(4)
            if not np.abs(mat(r0["K"]) - mat(ro["K"])).max() > 1e-9 * max(1e-30, np.abs(mat(ro["K"])).max()):
                n_same += 1  # (every control clamped at every knot: both regularisations give K = 0)
            n_cases += 1
            n_traj += B
    finally:
        O.set_fixes(0)
    print("soak_regv: %d cases / %d trajectories clean against the oracle with the same switch (seed %d); %d cases where the two regularisations coincide (all gains zero)"
          % (n_cases, n_traj, seed, n_same))
    return 0


if __name__ == "__main__":
    sys.exit(main())
