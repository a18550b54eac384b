"""A device-driven walk (tests/parity.py) of a small free-running batch over a whole solve: every iteration of every trajectory
against the oracle, to the end.  usage: python scripts/long_walk.py [B] [iters] [lim] [scale]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR
from oracle import oracle as O
from tests.util import acrobot_x0
from tests import parity as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lim = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
T, DT = 499, 0.02
O.build()
x0 = acrobot_x0(B, scale=scale, seed=77)
u0 = np.zeros((B, T, 1))
om = O.Model("acrobot", u_lim=lim)
g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim)
try:
    r = P.walk_iterations(O, om, g, x0, u0, DT, iters, drive="gpu", verbose=False)
except AssertionError as e:
    print("UNEXPLAINED:", str(e)[:600])
    raise SystemExit(1)
tot = {kk: sum(p[kk] for p in r["per_iter"]) for kk in r["per_iter"][0] if kk != "iteration"}
print("walk of %d trajectories x up to %d iterations:" % (B, iters), tot, "worst cond ratio %.1f" % r["worst_cond_ratio"])
late = [p for p in r["per_iter"] if p["iteration"] >= 25]
print("iterations >= 25:", {kk: sum(p[kk] for p in late) for kk in tot})
st, it, al = g.status()
print("final status counts", {int(s): int((st == s).sum()) for s in np.unique(st)}, "median log10 cost %.3f" % np.median(np.log10(g.cost())))
ro = O.batch_solve(om, x0, u0, DT)
print("oracle free run: status", {int(s): int((ro["status"] == s).sum()) for s in np.unique(ro["status"])}, "median log10 cost %.3f" % np.median(np.log10(ro["cost"])))
