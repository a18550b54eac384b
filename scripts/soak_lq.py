"""Randomised soak of the generic backward kernels (run on the GPU box).

Every case draws dimensions n <= 32, m <= 16, a batch, a horizon, limits, lambda and (sometimes) a negative shift
of one diagonal entry of cuu (indefinite Quu: partial factors, stale factors, aborted passes), then checks
  * (through round 5: k_backward_w2 == round 1's LDS kernel k_backward_w, bit for bit; the LDS kernel was retired in ABI 5)
  * k_backward_w2 AND k_backward_w3 (the default: per-lane matrix-vector sums, the box-QP's inverse refined on the matrix cores where the
    free set is the previous knot's, the literal path elsewhere) against the oracle's backward_pass per knot (tests/parity.check_backward:
    1e-6, deviations must be clamp knife edges or fp64-conditioning-limited against the fp80 oracle); where the indefinite shift makes the
    ORACLE itself miss the fp80 answer by more than 100 % per knot, k_backward_w3 is held to the diverge flag only (max_unpinned);
  * the two against each other at 1e-9 on the unshifted cases.

    python scripts/soak_lq.py [seconds] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from oracle import oracle as O
from tests.parity import check_backward, first_gain_mismatch_is_knife_edge
from tests.util import mat

DT = 0.02


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_traj = n_ties = n_abort = n_blown = 0
    while time.time() < t_end:
        n = int(rng.integers(2, 33))
        m = int(rng.integers(1, min(16, n) + 1)) if rng.random() < 0.8 else int(rng.integers(1, 17))
        B = int(rng.choice([1, 3, 8, 20]))
        T = int(rng.choice([1, 2, 5, 12, 30]))
        lim = float(rng.choice([0.1, 0.3, 1.0, 5.0]))
        lam = float(rng.choice([0.0, 1e-3, 1.0]))
        A = -np.eye(n) + 0.1 * rng.normal(size=(n, n)) / np.sqrt(n)
        Bm = rng.normal(size=(n, m)) / np.sqrt(n)
        om = O.Model("lq", lq=(A, Bm, np.eye(n), 0.1 * np.eye(m), np.eye(n)), u_lim=lim)
        x0 = rng.uniform(-1, 1, (B, n))
        u0 = rng.normal(size=(B, T, m)) * 0.5
        xs, us, cost = O.batch_rollout(om, x0, u0, DT)
        dv = O.batch_derivatives(om, xs, us, DT)
        shifted = rng.random() < 0.3
        if shifted:
            s = np.zeros(m)
            s[int(rng.integers(0, m))] = -float(rng.choice([5.0, 60.0]))
            dv["cuu"] = dv["cuu"] + np.diag(s)[None, None]
            lam = 0.0
        k_prev = rng.normal(size=(B, T, m)) * 0.1
        desc = "n=%d m=%d B=%d T=%d lim=%g lam=%g shifted=%s seed=%d" % (n, m, B, T, lim, lam, shifted, seed)
        outs = []
        for route in (capi.ROUTE_BACKWARD_W2, 0):
            try:
                g = BatchILQR("host", B, T, DT, nx=n, nu=m, u_min=om.u_min, u_max=om.u_max, route=route)
                g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
                g.set_derivatives(**{k: (dv[k] if k in ("cx", "cu") else mat(dv[k])) for k in dv})
                g.set_gains(k=k_prev, K=np.zeros((B, T, m, n)))
                g.set_lambda(lam, 1.0)
                div = np.asarray(g.backward_pass())
                k, K = g.gains()
                outs.append(dict(div=div, k=k, K=K, dV=g.dV(), gnorm=g.gnorm()))
                g.close()
            finally:
                pass
        if not np.array_equal(outs[0]["div"], outs[1]["div"]):
            print("FAIL k_backward_w3 diverge knots:", desc, outs[0]["div"], outs[1]["div"])
            return 1
        if not shifted:  # the two register kernels against each other: 1e-9, or a clamp knife edge (a component inside the 1e-4 band of a bound
            # that one of them reads as clamped: tests/parity.py) -- bounded like the ties against the oracle
            lo_b, hi_b = om.u_min[None, None, :] - us, om.u_max[None, None, :] - us
            n_edge = 0
            for bb in range(B):
                close = all(np.abs(outs[0][key][bb] - outs[1][key][bb]).max() <= 1e-9 * max(1.0, np.abs(outs[0][key][bb]).max()) for key in ("k", "K", "dV", "gnorm"))
                if close:
                    continue
                if first_gain_mismatch_is_knife_edge(outs[1]["k"][bb], outs[1]["K"][bb], outs[0]["k"][bb], outs[0]["K"][bb], us[bb], lo_b[bb], hi_b[bb], 1e-9):
                    n_edge += 1
                    continue
                print("FAIL k_backward_w3 vs k_backward_w2 (not a clamp knife edge): trajectory", bb, np.abs(outs[0]["k"][bb] - outs[1]["k"][bb]).max(), desc)
                return 1
            if n_edge > max(1, B // 4):
                print("FAIL too many knife edges between the two kernels:", n_edge, desc)
                return 1
            n_ties += n_edge
        ro = O.batch_backward(om, us, dv, k_prev=k_prev, lam=lam)
        if (ro["diverge"] == 0).sum() == 0:  # every pass aborts in the oracle: the abort knots must agree
            if not np.array_equal(outs[0]["div"], ro["diverge"]):
                print("FAIL diverge knots:", desc, outs[0]["div"], ro["diverge"])
                return 1
            n_abort += B
            n_cases += 1
            n_traj += B
            continue
        # An indefinite Quu that the reference's unchecked factorisation lets through can blow the recursion up
        # (gains of 1e16, dV of 1e45: every rounding is amplified without bound and "parity" means nothing): those
        # trajectories are counted, not compared -- the two kernels above still had to agree on them bit for bit.
        sane = np.array([np.all(np.isfinite(ro["K"][b])) and np.abs(ro["K"][b]).max() < 1e6 and np.abs(ro["k"][b]).max() < 1e6
                         and np.abs(ro["dV"][b]).max() < 1e12 for b in range(B)])
        n_blown += int((~sane).sum())
        if sane.sum() == 0 or (ro["diverge"][sane] == 0).sum() == 0:
            n_cases += 1
            n_traj += B
            continue
        sub = lambda a: a[sane]
        ro_s = {kk: v[sane] for kk, v in ro.items()}
        try:
            r = check_backward(O, om, sub(us), {kk: v[sane] for kk, v in dv.items()}, sub(k_prev), lam, sub(outs[0]["k"]), sub(outs[0]["K"]),
                               sub(outs[0]["dV"]), sub(outs[0]["div"]), ro_s, max_ties=max(1, B // 4), max_over10=max(1, B // 16))
            r3 = check_backward(O, om, sub(us), {kk: v[sane] for kk, v in dv.items()}, sub(k_prev), lam, sub(outs[1]["k"]), sub(outs[1]["K"]),
                                sub(outs[1]["dV"]), sub(outs[1]["div"]), ro_s, max_ties=max(1, B // 4), max_over10=max(1, B // 16),
                                max_unpinned=(B if shifted else 0))
        except AssertionError as e:
            print("FAIL oracle parity:", desc, str(e)[:300])
            return 1
        n_ties += int(r3["ties"])
        n_ties += int(r["ties"])
        n_abort += int((outs[0]["div"] != 0).sum())
        n_cases += 1
        n_traj += B
    print("soak_lq ok: %d cases, %d trajectories, %d ties / conditioning-limited, %d aborted passes, %d blown up (not compared), seed %d"
          % (n_cases, n_traj, n_ties, n_abort, n_blown, seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
