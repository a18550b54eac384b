"""One case of scripts/soak_lq.py: construction from its seed, and the classification of a deviating
trajectory -- which iteration deviates first, and is the backward pass of that iteration, teacher-
forced on the GPU's own state, on a clamp knife edge?

    python scripts/lq_case.py <case-seed> <trajectory>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from oracle import oracle as O
from tests.test_gpu_lq_end_to_end import dense_mats, lq_mats
from tests.parity import first_gain_mismatch_is_knife_edge


def _is_clamp_knife_edge(k, K, ko, Ko, lo, hi):
    return first_gain_mismatch_is_knife_edge(k, K, ko, Ko, np.zeros_like(k), lo, hi)
from tests.util import mat

DT = 0.02


def make_case(cs):
    r = np.random.default_rng(cs)
    n = int(r.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32]))
    m = int(r.choice([1, 2, 3, 7, 8, 15, 16]))
    B = int(r.choice([1, 2, 5, 6, 11, 33]))
    T = int(r.choice([1, 2, 3, 9, 20]))
    if n >= 16 and B * T > 120:
        B = max(1, 120 // T)
    iters = int(r.integers(1, 4))
    lim = float(r.choice([0.3, 1.0, 10.0]))
    mats = (dense_mats if r.random() < 0.7 else lq_mats)(n, m, seed=int(r.integers(1, 1000)))
    x0 = r.uniform(-1, 1, (B, n))
    u0 = r.normal(size=(B, T, m)) * float(r.choice([0.0, 0.3]))
    return dict(n=n, m=m, B=B, T=T, iters=iters, lim=lim, mats=mats, x0=x0, u0=u0, om=O.Model("lq", lq=mats, u_lim=lim),
                desc="lq n=%d m=%d B=%d T=%d lim=%g iters=%d case-seed=%d" % (n, m, B, T, lim, iters, cs))


def gpu(c, flags=capi.FLAG_FIXED_WORK):
    return BatchILQR("lq", c["B"], c["T"], DT, u_min=-c["lim"], u_max=c["lim"], lq=c["mats"], flags=flags)


def classify(c, bb, verbose=False):
    """True if trajectory bb's deviation starts at a clamp knife edge of one backward pass."""
    om, x0, u0, T = c["om"], c["x0"], c["u0"], c["T"]
    first = None
    for it in range(1, c["iters"] + 1):
        g = gpu(c)
        g.init_traj(x0, u0)
        g.iterate(it)
        ro = O.batch_solve(om, x0, u0, DT, max_iters=it, fixed_work=True)
        rel = abs(g.cost()[bb] - ro["cost"][bb]) / abs(ro["cost"][bb])
        if verbose:
            print("after %d iterations: cost rel %.2e   lambda gpu %.6g oracle %.6g   alpha idx gpu %d" % (it, rel, g.lambdas()[0][bb], ro["lam"][bb], g.status()[2][bb]))
        g.close()
        if rel > 1e-9:
            first = it
            break
    if first is None:
        return True
    backs = {}
    for source in ("gpu", "oracle"):  # the state both sides are given: the GPU's after first-1 iterations, then the oracle's
        r = _classify_from(c, bb, first, source, verbose, backs)
        if r is not None:
            return r
    # Both sides agree from either state.  Then the two STATES (1e-13 apart after first-1 iterations)
    # sit on different sides of a branch: compare the oracle's backward pass from the one with the
    # oracle's backward pass from the other.
    if len(backs) == 2:
        (ka, Ka, lo, hi), (kb, Kb, _, _) = backs["gpu"], backs["oracle"]
        if np.abs(ka - kb).max() > 1e-6 * max(np.abs(kb).max(), 1e-300) or np.abs(Ka - Kb).max() > 1e-6 * max(np.abs(Kb).max(), 1e-300):
            v = bool(_is_clamp_knife_edge(ka, Ka, kb, Kb, lo, hi))
            if verbose:
                print(" oracle backward from the GPU's state vs from its own: max|k diff| %.3e max|K diff| %.3e knife edge: %s"
                      % (np.abs(ka - kb).max(), np.abs(Ka - Kb).max(), v))
            return v
    return False


def _classify_from(c, bb, first, source, verbose, backs):
    """None: both sides agree from this state; True / False: they differ, at a tie / not at a tie."""
    om, x0, u0, T = c["om"], c["x0"], c["u0"], c["T"]
    g = gpu(c)
    g.init_traj(x0, u0)
    g.iterate(first - 1)
    if source == "oracle" and first > 1:
        ro = O.batch_solve(om, x0, u0, DT, max_iters=first - 1, fixed_work=True)
        _, dlam = g.lambdas()
        g.set_trajectory(x0=x0, xs=ro["xs"], us=ro["us"], cost=ro["cost"])
        g.set_gains(k=ro["k"], K=ro["K"])
        g.set_lambda(ro["lam"], dlam)
    elif source == "oracle":  # the oracle's own initial rollout (differs from the GPU's in the last bit)
        xs_o, us_o, c_o = O.batch_rollout(om, x0, u0, DT)
        g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=c_o)
    if verbose:
        print(" state after %d iterations taken from: %s" % (first - 1, source))
    xs_g, us_g = g.trajectory()
    k_g, K_g = g.gains()
    lam_g, _ = g.lambdas()
    dv = O.batch_derivatives(om, xs_g, us_g, DT)
    rb = O.batch_backward(om, us_g, dv, k_prev=k_g, lam=lam_g)
    Ko = mat(rb["K"])[bb]
    lo, hi = om.u_min[None, :] - us_g[bb], om.u_max[None, :] - us_g[bb]
    backs[source] = (rb["k"][bb].copy(), Ko.copy(), lo, hi)
    verdicts = []
    for own in (False, True):  # the oracle's derivative records, then the GPU's own (FD noise ~1e-10 apart)
        if own:
            g.compute_derivatives()
        else:
            g.set_derivatives(**{kk: (dv[kk] if kk in ("cx", "cu") else mat(dv[kk])) for kk in dv})
        g.set_gains(k=k_g, K=K_g)  # (the box-QP of t = T-1 is warm-started from the stored k)
        div = g.backward_pass()
        if verbose:
            print("  diverge gpu %d oracle %d" % (div[bb], rb["diverge"][bb]))
        k2, K2 = g.gains()
        dk, dK = np.abs(k2[bb] - rb["k"][bb]).max(), np.abs(K2[bb] - Ko).max()
        if dk <= 1e-6 * max(np.abs(rb["k"][bb]).max(), 1e-300) and dK <= 1e-6 * max(np.abs(Ko).max(), 1e-300):
            v = None  # agrees (norm-wise 1e-6, the tolerance of the parity tests)
        else:
            v = bool(_is_clamp_knife_edge(k2[bb], K2[bb], rb["k"][bb], Ko, lo, hi))
        verdicts.append(v)
        if verbose:
            print("  iteration %d backward pass, %s derivatives: max|k diff| %.3e max|K diff| %.3e knife edge: %s"
                  % (first, "GPU's own" if own else "oracle's", dk, dK, "n/a (agrees)" if v is None else v))
    if all(v is None for v in verdicts):
        # The backward pass agrees: then the line search decided differently.  That is a tie when a
        # candidate's cost change is zero to rounding -- z = dcost / expected (ilqr_core.cpp:199-206)
        # has the sign of noise there, and the first alpha with z > 0 wins.
        costs = g.rollout_candidates()[bb]
        cost_s = g.cost()[bb]
        dcost = cost_s - costs
        tie = bool(np.any(np.abs(dcost) <= 1e-9 * abs(cost_s)))
        if verbose:
            print("  line search: cost_s %.15g, dcost per alpha %s -> tie: %s" % (cost_s, np.array2string(dcost, precision=3), tie))
        g.close()
        return True if tie else None
    g.close()
    return any(v for v in verdicts if v is not None) and not any(v is False for v in verdicts)


if __name__ == "__main__":
    case = make_case(int(sys.argv[1]))
    print(case["desc"])
    print("knife edge:", classify(case, int(sys.argv[2]), verbose=True))
