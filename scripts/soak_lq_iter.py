"""Randomised soak of the generic path's whole ITERATIONS with exact derivatives (run on the GPU box): the fused route (k_backward_w3<.., LQF>:
no sweep kernel, no record array, cx / cu formed in the backward pass; the search kernel accepts) against round 2's route
(ILQR_ROUTE_BACKWARD_W2: k_analytic_lq + k_backward_w2 + k_accept) on random dimensions, limits and horizons -- costs, gains and the
trajectories after a few free-running iterations must agree to rounding (the two differ by summation order only; a clamp knife edge on one
side shows up as a larger deviation and is counted, bounded at 5 % of the trajectories) -- and whole solves must end with
the same statuses for all but those.

    python scripts/soak_lq_iter.py [seconds] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi

DT = 0.02


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_traj = n_moved = 0
    while time.time() < t_end:
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
        desc = "n=%d m=%d B=%d T=%d lim=%g iters=%d seed=%d" % (n, m, B, T, lim, iters, seed)
        outs = []
        for route in (0, capi.ROUTE_BACKWARD_W2):
            g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_ANALYTIC_DERIVATIVES, route=route)
            c0 = g.init_traj(x0, u0)
            g.iterate(iters)
            xs, us = g.trajectory()
            k, K = g.gains()
            mid = dict(c0=c0, cost=g.cost(), xs=xs, us=us, k=k, K=K, al=g.status()[2])
            g.generate_trajectory()
            mid.update(end_cost=g.cost(), end_status=g.status()[0])
            outs.append(mid)
            g.close()
        a, b = outs
        if not np.array_equal(a["c0"], b["c0"]):
            print("FAIL initial cost:", desc)
            return 1
        # A clamp knife edge (a control inside the 1e-4 band of a bound that one route reads as clamped) moves that control by <= 1e-4 and
        # everything downstream a little: such trajectories are COUNTED (cost or controls off by more than rounding) and bounded, the others
        # must agree to rounding in every array
        rel = np.abs(a["cost"] - b["cost"]) / np.maximum(np.abs(b["cost"]), 1e-300)
        du = np.abs(a["us"] - b["us"]).reshape(B, -1).max(axis=1) if T * m else np.zeros(B)
        dk = np.abs(a["k"] - b["k"]).reshape(B, -1).max(axis=1) if T * m else np.zeros(B)  # (a knife edge in the LAST backward pass has not reached us yet)
        # (... or shows only in K: a control within 1e-4 of a bound is clamped on one side -- its gain row exactly zero, ilqr_core.cpp:373-385 --
        #  and free on the other, with k the same to 1e-8)
        za = (np.abs(a["K"]).max(axis=-1) == 0) if T * m else np.zeros((B, 0, 0), dtype=bool)
        zb = (np.abs(b["K"]).max(axis=-1) == 0) if T * m else np.zeros((B, 0, 0), dtype=bool)
        row_flip = (za != zb).reshape(B, -1).any(axis=1) if T * m else np.zeros(B, dtype=bool)
        moved = (rel > 1e-8) | (du > 1e-7) | (dk > 1e-7 * max(1.0, np.abs(b["k"]).max())) | row_flip
        ok = ~moved
        for key in ("xs", "us", "k", "K"):
            scale = max(1.0, np.abs(b[key]).max())
            if ok.any() and not np.abs(a[key][ok] - b[key][ok]).max() <= 1e-6 * scale:
                print("FAIL", key, np.abs(a[key][ok] - b[key][ok]).max(), scale, desc)
                return 1
        if moved.any() and not rel[moved].max() < 1e-2:  # (its controls may sit anywhere in the box: with more controls than states many are equivalent)
            print("FAIL a moved trajectory is far off:", rel[moved].max(), du[moved].max(), desc)
            return 1
        # whole solves: the same end cost; a different STATUS at the same cost is a termination tie (the absolute stopping tests met an
        # iteration apart), a different cost must stay rare
        erel = np.abs(a["end_cost"] - b["end_cost"]) / np.maximum(np.abs(b["end_cost"]), 1e-300)
        if not np.all(np.isfinite(a["end_cost"])) or (ok.any() and (erel[ok] > 1e-4).mean() > 0.05):
            print("FAIL whole solves:", desc, (a["end_status"] != b["end_status"]).mean(), erel.max())
            return 1
        n_moved += int(moved.sum())
        n_cases += 1
        n_traj += B
        if n_moved > max(5, 0.05 * n_traj):
            print("FAIL too many trajectories moved by a tie:", n_moved, n_traj, desc)
            return 1
    print("soak_lq_iter ok: %d cases, %d trajectories, %d moved by a tie (> 1e-8 in cost after the free-running iterations), seed %d" % (n_cases, n_traj, n_moved, seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
