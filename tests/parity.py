"""Parity machinery shared by the GPU tests (test infrastructure).

* knot_relerr: PER-KNOT relative error (north_star: "K/k gains within 1e-6 relative"): every time step
  is measured against its own magnitude, not against the largest entry of the trajectory.
* walk_iterations: per-iteration teacher forcing.  Several iLQR iterations on chaotic dynamics amplify
  last-bit differences (the reference does that against itself between compiler flags),
  so an end-to-end comparison after N iterations cannot tell a bug from amplification.  Instead both
  sides start EVERY iteration from the same state (the oracle's), run one outer iteration, and every
  trajectory whose outcome differs must be PROVEN to sit on a tie:
    - a box-QP clamp-membership knife edge in the backward pass (gains differ first at a step where a
      control sits inside the 1e-4 approx_eq band of a bound, boxqp.h:61-64 / boxqp.cpp:65-71), or
    - a line-search tie (gains agree; a candidate's cost change is zero to rounding, so the sign of
      z = dcost / expected, ilqr_core.cpp:199-206, is noise), or
    - a termination tie (same trajectory and cost; dcost within rounding of tolFun, or lambda within
      rounding of lambdaMax: ilqr_core.cpp:257, :276).
  Anything else fails the test."""
import numpy as np

from tests.util import TOL, mat

# documented absolute floors of the per-knot scales:
#   k (feed-forward step): lives in the box [u_min - u, u_max - u]; the solver itself measures it against
#     |u| + 1 (get_gradient_norm, ilqr_core.cpp:405-412).  Floor 1e-3 (|u_t| + 1): a k_t below a thousandth
#     of that scale is compared absolutely (error <= 1e-9 (|u_t| + 1)).
#   K (feedback gain): floor 1e-3, i.e. an all-but-zero gain row is compared to 1e-9 absolute.
K_FLOOR = 1e-3


def knot_relerr(a, b, floor):
    """a, b [B][T][...]; floor scalar or [B][T]: err[b, t] = max|a - b| over the knot / max(max|b|, floor)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    B, T = a.shape[:2]
    da = np.abs(a - b).reshape(B, T, -1).max(axis=2)
    sb = np.abs(b).reshape(B, T, -1).max(axis=2)
    return da / np.maximum(sb, floor)


def gains_knot_errs(k, K, ko, Ko, us):
    """per-knot relative errors (ek, eK), each [B][T]."""
    ek = knot_relerr(k, ko, 1e-3 * (np.abs(us).max(axis=2) + 1.0))
    eK = knot_relerr(K, Ko, K_FLOOR)
    return ek, eK


def gains_knot_err(k, K, ko, Ko, us):
    """worst per-knot relative error of (k, K) per trajectory, [B]."""
    ek, eK = gains_knot_errs(k, K, ko, Ko, us)
    return np.maximum(ek, eK).max(axis=1)


def first_gain_mismatch_is_knife_edge(k, K, ko, Ko, us, lo, hi, tol=TOL):
    """One trajectory ([T][...]).  True if the first (largest-t) knot whose gains differ per-knot is a
    box-QP clamp tie: a component of k sits inside the 1e-4 band of a bound on one side or the other and
    the two k differ by no more than the band (see module docstring).  False when nothing differs."""
    ek, eK = gains_knot_errs(k[None], K[None], ko[None], Ko[None], us[None])
    bad = np.flatnonzero((ek[0] > tol) | (eK[0] > tol))
    if bad.size == 0:
        return False
    t = bad.max()
    if np.abs(k[t] - ko[t]).max() > 2e-4:
        return False
    band = np.minimum(np.abs(ko[t] - lo[t]), np.abs(ko[t] - hi[t]))
    band_g = np.minimum(np.abs(k[t] - lo[t]), np.abs(k[t] - hi[t]))
    return bool(np.any(band < 1.5e-4) or np.any(band_g < 1.5e-4))


def oracle_init_state(oracle, om, x0, u0, dt):
    xs, us, cost = oracle.batch_rollout(om, x0, u0, dt)
    B, T = u0.shape[:2]
    return dict(xs=xs, us=us, k=np.zeros((B, T, om.nu)), K=np.zeros((B, T, om.nu, om.nx)), cost=cost,
                lam=np.ones(B), dlam=np.ones(B))


def load_state(g, x0, st):
    g.set_trajectory(x0=x0, xs=st["xs"], us=st["us"], cost=st["cost"])
    g.set_gains(k=st["k"], K=st["K"])
    g.reset_state(warm=True)  # status / iteration count / flgChange restart
    g.set_lambda(st["lam"], st["dlam"])


def gpu_state(g):
    xs, us = g.trajectory()
    k, K = g.gains()
    lam, dlam = g.lambdas()
    st, it, al = g.status()
    return dict(xs=xs, us=us, k=k, K=K, cost=g.cost(), lam=lam, dlam=dlam, status=st, iters=it, alpha=al,
                gnorm=g.gnorm(), dV=g.dV())


def walk_iterations(oracle, om, g, x0, u0, dt, n_iters, fixed_work=False, tol=TOL, params=None, verbose=False):
    """See the module docstring.  `g` is a BatchILQR built for the same model / limits (NOT in fixed-work
    mode unless fixed_work).  Returns dict(checked, ties_backward, ties_search, ties_stop, worst) and raises
    AssertionError on the first unexplained deviation."""
    p = dict(tol_fun=1e-6, lambda_max=1e11)
    p.update(params or {})
    B, T = u0.shape[:2]
    st = oracle_init_state(oracle, om, x0, u0, dt)
    running = np.ones(B, dtype=bool)
    out = dict(checked=0, ties_backward=0, ties_search=0, ties_stop=0, worst_cost=0.0, worst_gain=0.0)
    for it in range(n_iters):
        if not running.any():
            break
        nx = oracle.batch_iterate_from(om, x0, st["xs"], st["us"], st["k"], st["K"], st["cost"], st["lam"], st["dlam"],
                                       dt, n_iters=1, fixed_work=fixed_work)
        load_state(g, x0, st)
        g.iterate(1)
        gs = gpu_state(g)
        lo = om.u_min[None, None, :] - st["us"]
        hi = om.u_max[None, None, :] - st["us"]
        eg = gains_knot_err(gs["k"], gs["K"], nx["k"], nx["K"], st["us"])
        ec = np.abs(gs["cost"] - nx["cost"]) / np.maximum(np.abs(nx["cost"]), 1e-300)
        for b in np.flatnonzero(running):
            out["checked"] += 1
            same_disc = gs["alpha"][b] == nx["alpha"][b] and gs["status"][b] == nx["status"][b]
            lam_ok = np.isclose(gs["lam"][b], nx["lam"][b], rtol=1e-12, atol=0) and np.isclose(gs["dlam"][b], nx["dlam"][b], rtol=1e-12)
            if same_disc and lam_ok and eg[b] < tol and ec[b] < tol:
                # get_gradient_norm (ilqr_core.cpp:405-412) and dV of the pass both sides agree on
                assert abs(gs["gnorm"][b] - nx["gnorm"][b]) <= 1e-9 * max(nx["gnorm"][b], 1e-300) + 1e-6 * tol, (gs["gnorm"][b], nx["gnorm"][b])
                assert np.allclose(gs["dV"][b], nx["dV"][b], rtol=tol, atol=tol * abs(st["cost"][b])), (gs["dV"][b], nx["dV"][b])
                out["worst_cost"] = max(out["worst_cost"], float(ec[b]))
                out["worst_gain"] = max(out["worst_gain"], float(eg[b]))
                continue
            where = "iteration %d trajectory %d: alpha %d/%d status %d/%d lambda %.6g/%.6g cost err %.2e gain err %.2e" % (
                it, b, gs["alpha"][b], nx["alpha"][b], gs["status"][b], nx["status"][b], gs["lam"][b], nx["lam"][b], ec[b], eg[b])
            if verbose:
                print("deviation:", where)
            if eg[b] >= tol:
                # the backward passes (incl. their lambda retries) differ: must start at a clamp knife edge
                assert first_gain_mismatch_is_knife_edge(gs["k"][b], gs["K"][b], nx["k"][b], nx["K"][b], st["us"][b], lo[b], hi[b], tol), \
                    "backward passes differ away from a clamp tie -- " + where
                out["ties_backward"] += 1
                continue
            # gains agree.  Line search or termination decided differently.
            if gs["alpha"][b] != nx["alpha"][b]:
                cc = _candidate_costs(g, x0, st, b)
                dcost = st["cost"][b] - cc
                a_lo = min(x for x in (gs["alpha"][b], nx["alpha"][b]) if x >= 0)
                # the earlier-accepted alpha (or every alpha when one side found none) has a cost change of rounding size
                cand = dcost[a_lo:] if min(gs["alpha"][b], nx["alpha"][b]) < 0 else dcost[a_lo:a_lo + 1]
                assert np.any(np.abs(cand) <= 1e-9 * abs(st["cost"][b])), "line searches differ away from a tie (dcost %s) -- %s" % (dcost, where)
                out["ties_search"] += 1
                continue
            if gs["status"][b] != nx["status"][b]:
                dcost = st["cost"][b] - nx["cost"][b]
                near_tolfun = abs(dcost - p["tol_fun"]) <= 1e-9 * abs(st["cost"][b])
                near_lmax = abs(nx["lam"][b] - p["lambda_max"]) <= 1e-9 * p["lambda_max"]
                near_grad = {int(gs["status"][b]), int(nx["status"][b])} == {0, 1} or 1 in (int(gs["status"][b]), int(nx["status"][b]))
                assert near_tolfun or near_lmax or (near_grad and abs(gs["gnorm"][b] - 1e-6) < 1e-9), "terminations differ away from a tie -- " + where
                out["ties_stop"] += 1
                continue
            raise AssertionError("same gains, alpha and status but cost / lambda differ -- " + where)
        running &= nx["status"] == 0
        st = {kk: nx[kk] for kk in ("xs", "us", "k", "K", "cost", "lam", "dlam")}
    return out


def _candidate_costs(g, x0, st, b):
    load_state(g, x0, st)
    g.compute_derivatives()
    g.backward_step()
    return g.rollout_candidates()[b]
