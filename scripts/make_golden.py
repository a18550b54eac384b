"""Generates tests/golden/*.npz (run in the build container, where /root/reference is mounted).

  ref_pieces.npz          outputs of the REAL reference code (oracle/_ref = src/boxqp.cpp,
                          finite_diff.h, acrobot.h, double_integrator.h compiled where they lie):
                          finite-difference arrays of random knot points and box-QP solutions.
  stages_<model>.npz      stage-by-stage outputs of the CPU oracle for a small batch (the oracle is
                          pinned to the reference by tests/test_oracle_*.py).
Only data (inputs + expected outputs) is written; no reference source is stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.util import acrobot_x0, integrator_x0  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
DT = 0.02


def ref_pieces():
    assert O.ref_available(), "needs /root/reference (build container)"
    rng = np.random.default_rng(2024)
    d = {}
    for mid, name, goal, nu in ((0, "acrobot", np.zeros(4), 1), (1, "integrator", np.array([1, .5, 0, 0.]), 2)):
        N = 48
        xs = rng.uniform(-1, 1, (N, 4)) * np.array([np.pi, np.pi, 3, 3])
        us = rng.uniform(-3, 3, (N, nu))
        keys = ("fx", "fu", "cx", "cu", "cxx", "cuu")
        acc = {k: [] for k in keys}
        accf = {k: [] for k in ("cx", "cxx", "cuu")}
        for x, u in zip(xs, us):
            r = O.ref_fd_knot(mid, goal, x, u, DT, 0)
            for k in keys:
                acc[k].append(r[k])
            rf = O.ref_fd_knot(mid, goal, x, u, DT, 1)
            for k in accf:
                accf[k].append(rf[k])
        d[name + "_x"], d[name + "_u"], d[name + "_goal"] = xs, us, goal
        for k in keys:
            d["%s_%s" % (name, k)] = np.array(acc[k])  # memory layout [col][row]
        for k in accf:
            d["%s_final_%s" % (name, k)] = np.array(accf[k])
    # box-QP golden vectors, m = 1 and 2
    for m in (1, 2):
        N = 256
        Q = np.zeros((N, m, m)); c = np.zeros((N, m)); x0 = np.zeros((N, m)); lo = np.zeros((N, m)); hi = np.zeros((N, m))
        xo = np.zeros((N, m)); vf = np.zeros((N, m), dtype=np.int32); res = np.zeros(N, dtype=np.int32)
        for i in range(N):
            A = rng.normal(size=(m, m))
            Q[i] = A @ A.T + (0.05 if i % 5 else -0.2) * np.eye(m)
            c[i] = rng.normal(size=m) * 2
            x0[i] = rng.normal(size=m)
            lo[i] = -rng.uniform(0.05, 1.5, size=m)
            hi[i] = rng.uniform(0.05, 1.5, size=m)
            r = O.ref_boxqp(Q[i], c[i], x0[i], lo[i], hi[i])
            xo[i], vf[i], res[i] = r["x_opt"], r["v_free"], r["result"]
        for k, v in (("Q", Q), ("c", c), ("x0", x0), ("lo", lo), ("hi", hi), ("x_opt", xo), ("v_free", vf), ("result", res)):
            d["qp%d_%s" % (m, k)] = v
    np.savez_compressed(os.path.join(OUT, "ref_pieces.npz"), **d)


def stages(name):
    B, T = 6, 24
    if name == "acrobot":
        om = O.Model("acrobot", u_lim=1.5)
        x0 = acrobot_x0(B, seed=99)
        goal, lim = np.zeros(4), 1.5
    else:
        goal, lim = np.array([1, .5, 0, 0.]), 0.5
        om = O.Model("integrator", goal=goal, u_lim=lim)
        x0 = integrator_x0(B, seed=99)
    u0 = np.random.default_rng(5).normal(size=(B, T, om.nu)) * 0.4
    xs, us, cost = O.batch_rollout(om, x0, u0, DT)
    dv = O.batch_derivatives(om, xs, us, DT)
    k_prev = np.random.default_rng(6).normal(size=(B, T, om.nu)) * 0.1
    bw = O.batch_backward(om, us, dv, k_prev=k_prev, lam=1.0)
    Kmat = np.swapaxes(bw["K"], -1, -2)
    cand = np.stack([O.batch_rollout(om, x0, us + a * bw["k"], DT, xs_nom=xs, K=Kmat)[2] for a in O.ALPHAS], axis=1)
    sol = O.batch_solve(om, x0, np.zeros((B, T, om.nu)), DT, max_iters=3)
    d = dict(x0=x0, u0=u0, goal=goal, lim=lim, dt=DT, xs=xs, us=us, cost=cost, k_prev=k_prev,
             k=bw["k"], K=bw["K"], dV=bw["dV"], diverge=bw["diverge"], cand_cost=cand,
             sol_cost=sol["cost"], sol_lam=sol["lam"], sol_iters=sol["iters"], sol_xs=sol["xs"])
    for kk in O.DERIV_NAMES:
        d["d_" + kk] = dv[kk]
    np.savez_compressed(os.path.join(OUT, "stages_%s.npz" % name), **d)


if __name__ == "__main__":
    ref_pieces()
    stages("acrobot")
    stages("integrator")
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
