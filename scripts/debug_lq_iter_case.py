"""Replays scripts/soak_lq_iter.py's random stream up to the case whose description is given and referees a whole-solve disagreement
between the fused route and ILQR_ROUTE_BACKWARD_W2 with the oracle's own solve.   python scripts/debug_lq_iter_case.py seed "n=6 m=2 B=5 T=60 lim=1 iters=3" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from oracle import oracle as O

DT = 0.02
seed, want = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(seed)
for case in range(100000):
    n = int(rng.choice([1, 2, 3, 5, 6, 8, 12, 15, 16, 17, 24, 31, 32]))
    m = int(rng.choice([1, 2, 3, 4, 7, 15, 16]))
    B = int(rng.choice([1, 5, 33, 100]))
    T = int(rng.choice([1, 2, 7, 30, 60]))
    lim = float(rng.choice([0.1, 0.3, 1.0]))
    iters = int(rng.choice([1, 3, 6]))
    A = -np.eye(n) + 0.3 * rng.normal(size=(n, n)) / np.sqrt(n)
    Bm = rng.normal(size=(n, m)) / np.sqrt(n)

    def spd(k, s):
        W = rng.normal(size=(k, k)) / np.sqrt(k)
        return s * (np.eye(k) + 0.25 * (W @ W.T))  # positive definite whatever the draw (an indefinite weight makes the problem unbounded and every rounding decisive)
    mats = (A, Bm, spd(n, 1.0), spd(m, 0.2), spd(n, 3.0))
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.2
    desc = "n=%d m=%d B=%d T=%d lim=%g iters=%d" % (n, m, B, T, lim, iters)
    if desc != want or case > int(sys.argv[3] if len(sys.argv) > 3 else 100000):
        continue
    om = O.Model("lq", lq=mats, u_lim=lim)
    ro = O.batch_solve(om, x0, u0, DT)
    res = []
    for route, name in ((0, "fused w3"), (capi.ROUTE_BACKWARD_W2, "w2")):
        g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_ANALYTIC_DERIVATIVES, route=route)
        g.init_traj(x0, u0)
        g.iterate(iters)
        c_mid = g.cost()
        g.generate_trajectory()
        st, it, al = g.status()
        res.append((name, c_mid, g.cost(), it, st))
        g.close()
    bad = np.abs(res[0][2] - res[1][2]).max() > 1e-6 * np.abs(res[1][2]).max()
    print("case", case, desc, "DIFFERENT" if bad else "same")
    if bad:
        print("oracle  end cost", ro["cost"], "iters", ro["iters"], "status", ro["status"])
        for name, c_mid, c_end, it, st in res:
            print("%-9s mid cost" % name, c_mid, "end cost", c_end, "iters", it, "status", st)
