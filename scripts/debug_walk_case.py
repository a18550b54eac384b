"""Reproduce one (batch size, sample index, iteration) case of a sampled walk on a single-trajectory handle (routes are
bit-identical and trajectories independent) and show where device and oracle part ways.
usage: python scripts/debug_walk_case.py B sample_index iteration [lim]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR
from oracle import oracle as O
from tests.util import acrobot_x0, mat
from tests import parity as P

B, si, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lim = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
T, DT = 499, 0.02
O.build()
rng = np.random.default_rng(99)
sel = np.unique(np.concatenate([[0, B - 1], rng.choice(B, size=64 - 2, replace=False)]))
b = sel[si]
x0 = acrobot_x0(B)[b:b + 1]
u0 = np.zeros((1, T, 1))
om = O.Model("acrobot", u_lim=lim)
g = BatchILQR("acrobot", 1, T, DT, u_min=-lim, u_max=lim)
g.init_traj(x0, u0)
g.iterate(it)
st = P.gpu_state(g)
nx = P.twin_iterate(O, om, P.PRECISIONS["f64"], x0, st, DT, False)
g.iterate(1)
gs = P.gpu_state(g)
ek, eK = P.gains_knot_errs(gs["k"], gs["K"], nx["k"], nx["K"], st["us"])
bad = np.flatnonzero((ek[0] > 1e-6) | (eK[0] > 1e-6))
print("global trajectory", b, "iteration", it, "lambda", st["lam"], "alpha dev/orc", gs["alpha"], nx["alpha"], "cost", gs["cost"], nx["cost"])
print("knots beyond 1e-6:", len(bad), "largest t:", bad.max() if len(bad) else None)
if len(bad):
    t = bad.max()
    for tt in range(min(T - 1, t + 2), max(-1, t - 4), -1):
        lo, hi = -lim - st["us"][0, tt, 0], lim - st["us"][0, tt, 0]
        print("t=%d us=%.6g lo=%.6g hi=%.6g  k dev %.17g orc %.17g  |K| dev %.6g orc %.6g  ek %.2e eK %.2e" % (
            tt, st["us"][0, tt, 0], lo, hi, gs["k"][0, tt, 0], nx["k"][0, tt, 0], np.abs(gs["K"][0, tt]).max(), np.abs(nx["K"][0, tt]).max(), ek[0, tt], eK[0, tt]))
    # the oracle's backward pass on the device's own records
    aux = g.clone()
    P.load_state(aux, x0, st)
    aux.compute_derivatives()
    d = aux.derivatives()
    derivs = {kk: np.asarray(v if kk in ("cx", "cu") else mat(v), dtype=np.float64) for kk, v in d.items()}
    r = O.batch_backward(om, st["us"], derivs, k_prev=st["k"], lam=st["lam"])
    print("oracle backward on device records: diverge", r["diverge"], "gain err vs device", P.gains_knot_err(gs["k"], gs["K"], r["k"], mat(r["K"]), st["us"]))
    ro = O.batch_derivatives(om, st["xs"], st["us"], DT)
    r2 = O.batch_backward(om, st["us"], ro, k_prev=st["k"], lam=st["lam"])
    print("oracle backward on oracle records: diverge", r2["diverge"], "gain err vs oracle iterate", P.gains_knot_err(nx["k"], nx["K"], r2["k"], mat(r2["K"]), st["us"]))
    print("device diverge/backpass lambda:", gs["lam"], nx["lam"], "dV dev", gs["dV"], "orc", nx["dV"])
    for name in derivs:
        a, bb = derivs[name], np.asarray(ro[name], dtype=np.float64)
        print("records", name, "max abs diff %.3e (scale %.3e)" % (np.abs(a - bb).max(), np.abs(bb).max()))
