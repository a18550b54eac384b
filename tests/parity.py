"""Parity machinery shared by the GPU tests (test infrastructure).

* knot_relerr: PER-KNOT relative error (north_star: "K/k gains within 1e-6 relative"): every time step
  is measured against its own magnitude, not against the largest entry of the trajectory.
* conditioning: the backward recursion multiplies 499 Jacobians of an unstable system; the gains of early
  knots can be ill-conditioned functions of the records, and then NO fp64 evaluation order reproduces
  another to 1e-6 per knot (the reference does not reproduce itself across compiler flags there).  To tell
  that from a defect, the same step is also run in x87 extended precision (oracle flavour "f80", 64-bit
  mantissa): a trajectory whose worst per-knot gain error exceeds 1e-6 passes only if it is a proven tie
  (below) or the device is no further from the extended-precision answer than what the fp64 ORACLE
  ITSELF is, times a factor -- i.e. fp64 rounding, not the implementation, is what limits the agreement.
  The factor: 10 normally; up to 100 for a counted few (`cond_over10`, bounded by the tests): the two
  errors are single realisations of rounding noise and their ratio has a heavy tail.
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
    - or the INPUT of the backward pass: the device's finite differences round at the last bit differently from the oracle's; where
      Quu passes through zero that bit decides the sign of the gains.  Proven by running the oracle's backward pass on the device's
      own records (_explained_by_records).
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
    def on_edge(t):
        if np.abs(k[t] - ko[t]).max() > 2e-4:
            return False
        band = np.minimum(np.abs(ko[t] - lo[t]), np.abs(ko[t] - hi[t]))
        band_g = np.minimum(np.abs(k[t] - lo[t]), np.abs(k[t] - hi[t]))
        return bool(np.any(band < 1.5e-4) or np.any(band_g < 1.5e-4))

    t = bad.max()
    if on_edge(t):
        return True
    # The knots before the flip (larger t) may already sit a hair over the tolerance (conditioning: 1.0e-6 where 1e-6 is
    # asked) -- then the FIRST knot beyond tol is not the tie.  The tie itself is unmistakable: a gain row that is exactly
    # zero (clamped, boxqp.cpp:74-77 / ilqr_core.cpp:373-385) on one side and not on the other.  Accepted if that knot sits
    # on the band and every knot before it agrees to 10 tol.
    z = (np.abs(K).reshape(K.shape[0], K.shape[1], -1).max(axis=2) == 0) != (np.abs(Ko).reshape(Ko.shape[0], Ko.shape[1], -1).max(axis=2) == 0)
    flips = np.flatnonzero(z.any(axis=1))
    if flips.size == 0:
        return False
    tf = flips.max()
    before = np.arange(tf + 1, k.shape[0])
    return bool(on_edge(tf) and (before.size == 0 or max(ek[0][before].max(), eK[0][before].max()) < 10 * tol))


COND_FACTOR = 10.0   # device error vs extended precision: normally within this factor of the fp64 oracle's own
COND_HARD = 100.0    # ... and never beyond this one (errors are single realisations of rounding: the ratio of two
                     # has a heavy tail, so the tests bound how OFTEN it exceeds COND_FACTOR, and cap it here)


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def conditioning_verdict(k, K, k64, K64, k80, K80, us, tol=TOL):
    """Per trajectory: (e_dev, e_orc, ok) with e_* the worst per-knot relative error of the device's /
    the fp64 oracle's gains against the extended-precision gains, ok = fp64 rounding explains the
    device's deviation (see the module docstring)."""
    k80, K80 = _f64(k80), _f64(K80)
    e_dev = gains_knot_err(k, K, k80, K80, us)
    e_orc = gains_knot_err(k64, K64, k80, K80, us)
    ok = e_dev <= np.maximum(tol, COND_HARD * e_orc)
    return e_dev, e_orc, ok


# The product's fp32 mode (BASELINE configs[3]) is checked with the same machinery one precision down: the
# oracle's float build ("f32") is the twin the device must match, the fp64 oracle is the yardstick.
TOL32 = 1e-4   # float eps 6e-8 x O(100) operations per Riccati step x tens of steps of a recursion whose
               # condition grows with the horizon: what two correct fp32 evaluations agree to per knot
# what a closed-loop float rollout over T = 499 knots of an unstable system is good for (p90 of the ORACLE's own float rollouts
# against fp64 on the bench workload: 6e-4; scripts/f32_rollout_check.py)
ROLLOUT_NOISE = {"f32": 2e-3}
PRECISIONS = {
    # flavour the device is compared with, yardstick flavour, tolerance, relative cost change that counts as a tie
    # gtol: per-knot tolerance of the GAINS.  An fp32 handle's backward pass runs in double on float records, and
    # its twin's does too: the two see the same records almost always (the finite differences are taken in double and differ at
    # 1e-13 before they are rounded to float), so the gains agree far better than float rollouts do.
    "f64": dict(twin="f64", yard="f80", tol=TOL, gtol=TOL, tie_rel=1e-9),
    "f32": dict(twin="f32", yard="f64", tol=TOL32, gtol=1e-5, tie_rel=1e-5),
}


def _tw(om, name):
    return om if name == "f64" and om.flavour == "f64" else om.twin(name)


def backward_f80(oracle, om, us, derivs, k_prev, lam, yard="f80"):
    """oracle.batch_backward in the yardstick precision on the same inputs; K as [B][T][nu][nx]."""
    with oracle.flavour(yard):
        r = oracle.batch_backward(_tw(om, yard), us, derivs, k_prev=k_prev, lam=lam)
    return _f64(r["k"]), _f64(mat(r["K"])), r["diverge"]


def iterate_f80(oracle, om, x0, st, dt, fixed_work, sel, yard="f80"):
    """one outer iteration in the yardstick precision from the same state, trajectories `sel` only"""
    with oracle.flavour(yard):
        r = oracle.batch_iterate_from(_tw(om, yard), x0[sel], st["xs"][sel], st["us"][sel], st["k"][sel], st["K"][sel],
                                      st["cost"][sel], st["lam"][sel], st["dlam"][sel], dt, n_iters=1, fixed_work=fixed_work)
    return _f64(r["k"]), _f64(r["K"])


def twin_iterate(oracle, om, prec, x0, st, dt, fixed_work):
    """one outer iteration of the oracle flavour the device is compared with; everything back as float64"""
    with oracle.flavour(prec["twin"]):
        r = oracle.batch_iterate_from(_tw(om, prec["twin"]), x0, st["xs"], st["us"], st["k"], st["K"], st["cost"], st["lam"], st["dlam"],
                                      dt, n_iters=1, fixed_work=fixed_work)
    return {kk: (_f64(v) if v.dtype.kind == "f" else v) for kk, v in r.items()}


def check_backward(oracle, om, us, derivs, k_prev, lam, k, K, dV, div, ro, max_ties, max_over10, precision="f64", max_unpinned=0):
    """One teacher-forced backward pass: device outputs (k, K [B][T][nu][nx], dV, div) against the oracle's
    `ro` (batch_backward) for every trajectory the oracle completes: per-knot gains and dV within tol and
    the same diverge flag -- or fp64 rounding shown to be the limit (conditioning_verdict) -- or a proven
    clamp knife edge (at most max_ties of those).  Returns dict(good, ties, conditioned).
    max_unpinned (only the indefinite-Quu tests pass one): trajectories whose fp64 ORACLE is itself more than 100 % per knot away
    from the extended-precision answer -- a failed first pivot leaves R = Q and the "solve" amplifies rounding by 1e15, so no
    fp64 evaluation order determines these gains -- are compared by their diverge flag only and are NOT counted as good."""
    prec = PRECISIONS[precision]
    tol = prec["gtol"]  # (a teacher-forced backward pass: gains and dV; no float rollout is involved)
    ro = {kk: (_f64(v) if v.dtype.kind == "f" else v) for kk, v in ro.items()}
    us = _f64(us)
    Ko = mat(ro["K"])
    B = k.shape[0]
    lo, hi = om.u_min[None, None, :] - us, om.u_max[None, None, :] - us
    conv = ro["diverge"] == 0
    assert conv.sum() > 0
    eg = gains_knot_err(k, K, ro["k"], Ko, us)
    edv = np.abs(dV - ro["dV"]).max(axis=1) / np.maximum(np.abs(ro["dV"]).max(axis=1), 1e-300)
    good = (eg < tol) & (edv < tol) & (div == ro["diverge"])
    todo = np.flatnonzero(conv & ~good)
    ties = conditioned = over10 = unpinned = 0
    if todo.size:
        lam_b = np.broadcast_to(np.asarray(lam, dtype=np.float64), (B,))
        k80, K80, div80 = backward_f80(oracle, om, us[todo], {kk: v[todo] for kk, v in derivs.items()},
                                       None if k_prev is None else k_prev[todo], lam_b[todo], yard=prec["yard"])
        e_dev, e_orc, okc = conditioning_verdict(k[todo], K[todo], ro["k"][todo], Ko[todo], k80, K80, us[todo], tol)
        for i, b in enumerate(todo):
            if first_gain_mismatch_is_knife_edge(k[b], K[b], ro["k"][b], Ko[b], us[b], lo[b], hi[b], tol):
                ties += 1
                continue
            if max_unpinned and e_orc[i] > 1.0 and div[b] == ro["diverge"][b]:
                unpinned += 1
                continue
            assert okc[i] and div[b] == ro["diverge"][b] and edv[b] < max(tol, COND_HARD * e_orc[i]), \
                "trajectory %d: gain err %.2e dV err %.2e diverge %d/%d [vs yardstick: device %.2e, twin oracle %.2e]" % (
                    b, eg[b], edv[b], div[b], ro["diverge"][b], e_dev[i], e_orc[i])
            conditioned += 1
            over10 += int(e_dev[i] > max(tol, COND_FACTOR * e_orc[i]))
            good[b] = True
    assert ties <= max_ties, (ties, max_ties)
    assert unpinned <= max_unpinned, (unpinned, max_unpinned)
    assert np.array_equal(div[good | ~conv], ro["diverge"][good | ~conv])
    assert over10 <= max_over10, (over10, max_over10)  # every caller states its own bound
    return dict(good=conv & good, ties=ties, conditioned=conditioned, cond_over10=over10)


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


def walk_iterations(oracle, om, g, x0, u0, dt, n_iters, fixed_work=False, precision="f64", params=None, verbose=False, drive="oracle"):
    """See the module docstring.  `g` is a BatchILQR built for the same model / limits (in fixed-work mode
    iff fixed_work).  drive = "oracle": every iteration starts from the oracle's state on both sides;
    drive = "gpu": the device runs freely (init_traj, then iterate(1) again and again) and the ORACLE is
    given the device's state before every iteration -- i.e. every step of a free-running solve is checked.
    Returns dict(checked, ties_backward, ties_search, ties_stop, worst_*) and raises AssertionError on the
    first unexplained deviation."""
    p = dict(tol_fun=1e-6, lambda_max=1e11, tol_grad=1e-6)
    p.update(params or {})
    prec = PRECISIONS[precision]
    tol, gtol = prec["tol"], prec["gtol"]
    B, T = u0.shape[:2]
    aux = None
    if drive == "oracle":
        with oracle.flavour(prec["twin"]):
            st = oracle_init_state(oracle, _tw(om, prec["twin"]), x0, u0, dt)
        st = {kk: _f64(v) for kk, v in st.items()}
    else:
        g.init_traj(x0, u0)
        st = gpu_state(g)
    running = np.ones(B, dtype=bool)
    out = dict(checked=0, ties_backward=0, ties_search=0, ties_stop=0, conditioned=0, cond_over10=0, conditioned_branch=0, unresolved=0, worst_cond_ratio=0.0, tied=set(), worst_cost=0.0, worst_gain=0.0)
    out["per_iter"] = []
    for it in range(n_iters):
        if not running.any():
            break
        # what happened to this iteration's trajectories (profiles/parity_r05.json): every checked one lands in exactly one bin
        pit = dict(iteration=it, n=0, plain=0, knife_edge=0, plain_on_device_records=0, cond_le10=0, cond_le100=0, amplified=0, explained_by_records=0,
                   search_tie=0, stop_tie=0, unresolved=0)
        out["per_iter"].append(pit)
        nx = twin_iterate(oracle, om, prec, x0, st, dt, fixed_work)
        if drive == "oracle":
            load_state(g, x0, st)
        g.iterate(1)
        gs = gpu_state(g)
        lo = om.u_min[None, None, :] - st["us"]
        hi = om.u_max[None, None, :] - st["us"]
        eg = gains_knot_err(gs["k"], gs["K"], nx["k"], nx["K"], st["us"])
        ec = np.abs(gs["cost"] - nx["cost"]) / np.maximum(np.abs(nx["cost"]), 1e-300)
        g_status = np.where(gs["status"] == 4, 0, gs["status"])  # (max_iter is a property of the run, not of the step)
        cond_ok = None  # extended-precision verdicts for this iteration's deviating trajectories, computed on demand
        for b in np.flatnonzero(running):
            out["checked"] += 1
            pit["n"] += 1
            same_disc = gs["alpha"][b] == nx["alpha"][b] and g_status[b] == nx["status"][b]
            lam_ok = np.isclose(gs["lam"][b], nx["lam"][b], rtol=1e-12, atol=0) and np.isclose(gs["dlam"][b], nx["dlam"][b], rtol=1e-12)  # (double on both sides in both modes)
            if same_disc and lam_ok and eg[b] < gtol and ec[b] < tol:
                # get_gradient_norm (ilqr_core.cpp:405-412) and dV of the pass both sides agree on
                assert abs(gs["gnorm"][b] - nx["gnorm"][b]) <= tol * max(nx["gnorm"][b], 1e-3), (gs["gnorm"][b], nx["gnorm"][b])
                assert np.allclose(gs["dV"][b], nx["dV"][b], rtol=tol, atol=tol * abs(st["cost"][b])), (gs["dV"][b], nx["dV"][b])
                out["worst_cost"] = max(out["worst_cost"], float(ec[b]))
                out["worst_gain"] = max(out["worst_gain"], float(eg[b]))
                pit["plain"] += 1
                continue
            where = "iteration %d trajectory %d: alpha %d/%d status %d/%d lambda %.6g/%.6g cost err %.2e gain err %.2e" % (
                it, b, gs["alpha"][b], nx["alpha"][b], g_status[b], nx["status"][b], gs["lam"][b], nx["lam"][b], ec[b], eg[b])
            if verbose:
                print("deviation:", where)
            if eg[b] >= gtol:
                # the backward passes (incl. their lambda retries) differ: a clamp knife edge ...
                if first_gain_mismatch_is_knife_edge(gs["k"][b], gs["K"][b], nx["k"][b], nx["K"][b], st["us"][b], lo[b], hi[b], gtol):
                    out["ties_backward"] += 1
                    out["tied"].add(int(b))
                    pit["knife_edge"] += 1
                    continue
                # ... or the two backward passes were not given the same problem: the finite-difference records of the two sides
                # differ at the 1e-13 level (1e-16 of cancellation over 2 eps; different realisations of it on the device --
                # FMA contraction, x 1/(2 eps), another sincos -- and in the oracle), and the recursion amplifies record
                # noise a thousand times more than its own rounding.  Stage by stage the parity claim holds plainly: the
                # device's records agree with the oracle's to tol (checked here), and the ORACLE's backward pass on the
                # device's records reproduces the device's gains to the plain tolerance.
                if aux is None:
                    aux = g.clone()
                if _explained_by_records(oracle, om, prec, aux, x0, st, int(b), gs, gtol, plain_only=True):
                    pit["plain_on_device_records"] += 1
                    out["plain_on_device_records"] = out.get("plain_on_device_records", 0) + 1
                    if not (same_disc and lam_ok):  # the 1e-6-level gain difference then moved the line search / exit too
                        out["tied"].add(int(b))
                    continue
                # ... or fp64 rounding, not the implementation, limits the per-knot agreement
                if cond_ok is None:
                    sel = np.flatnonzero(running & (eg >= gtol))
                    k80, K80 = iterate_f80(oracle, om, x0, st, dt, fixed_work, sel, yard=prec["yard"])
                    e_dev, e_orc, okc = conditioning_verdict(gs["k"][sel], gs["K"][sel], nx["k"][sel], nx["K"][sel], k80, K80, st["us"][sel], gtol)
                    cond_ok = {int(bb): (bool(okc[i]), float(e_dev[i]), float(e_orc[i])) for i, bb in enumerate(sel)}
                okb, e_d, e_o = cond_ok[int(b)]
                where += " [vs yardstick: device %.2e, twin oracle %.2e]" % (e_d, e_o)
                if not okb and precision == "f32" and e_o >= 1e-2:
                    # float cannot resolve these gains at all: the float ORACLE is already > 1 % per knot away from
                    # the fp64 answer (long horizon, lambda -> 0: Quu = cuu + fu'Vxx fu cancels to a few float ulps)
                    out["unresolved"] += 1
                    out["tied"].add(int(b))
                    pit["unresolved"] += 1
                    continue
                if not okb:
                    # Last resort before failing: is it the INPUT of the backward pass, not the pass?  The device's finite
                    # differences round at the last bit differently from the oracle's (x 1/(2 eps) against / (2 eps), FMA
                    # contraction in the models); on a trajectory whose Quu passes through zero (lambda small, Vxx indefinite: Eigen's
                    # unchecked factor turns the sign of a rounding-noise Quu into gains of either sign) that last bit decides.
                    # Proof: the ORACLE's backward pass on the DEVICE's own records of this nominal must reproduce the device's gains.
                    if aux is None:
                        aux = g.clone()
                    if _explained_by_records(oracle, om, prec, aux, x0, st, int(b), gs, gtol):
                        out["explained_by_records"] = out.get("explained_by_records", 0) + 1
                        out["tied"].add(int(b))
                        pit["explained_by_records"] += 1
                        continue
                assert okb, "backward passes differ away from a clamp tie and beyond conditioning -- " + where
                out["conditioned"] += 1
                out["cond_over10"] += int(e_d > max(gtol, COND_FACTOR * e_o))
                pit["cond_le100" if e_d > max(gtol, COND_FACTOR * e_o) else "cond_le10"] += 1
                out["worst_cond_ratio"] = max(out["worst_cond_ratio"], e_d / max(e_o, 1e-300))
                if same_disc and lam_ok and ec[b] < max(tol, COND_HARD * e_o):
                    continue
                # the ill-conditioned gains then moved the line search / the exit as well: nothing more to compare
                out["conditioned_branch"] += 1
                out["tied"].add(int(b))
                continue
            # gains agree.  Line search or termination decided differently.
            if gs["alpha"][b] != nx["alpha"][b]:
                if aux is None:
                    aux = g.clone()
                cc = _candidate_costs(aux, x0, st, b)
                dcost = st["cost"][b] - cc
                a_lo = min(x for x in (gs["alpha"][b], nx["alpha"][b]) if x >= 0)
                # the earlier-accepted alpha (or every alpha when one side found none) has a cost change of rounding size
                cand = dcost[a_lo:] if min(gs["alpha"][b], nx["alpha"][b]) < 0 else dcost[a_lo:a_lo + 1]
                if precision == "f32" and not np.any(np.abs(cand) <= prec["tie_rel"] * abs(st["cost"][b])):
                    # Not a tie of rounding size -- but are this trajectory's FLOAT rollouts good for a decision at all?  The same gains
                    # (the device's, within tolerance of the twin's) rolled out by the float twin and by the fp64 yardstick, for every alpha
                    # up to the later-accepted one: if they disagree by more than what a float rollout over this horizon is good for
                    # (ROLLOUT_NOISE; overflow to inf / NaN included), the two searches were deciding on amplified rounding.
                    from oracle.oracle import ALPHAS
                    a_hi = max(int(gs["alpha"][b]), int(nx["alpha"][b]))
                    noisy = False
                    for a in range(a_hi + 1):
                        ut = np.asarray(st["us"][b:b + 1] + ALPHAS[a] * gs["k"][b:b + 1])
                        with oracle.flavour(prec["twin"]):
                            c_t = float(oracle.batch_rollout(_tw(om, prec["twin"]), x0[b:b + 1], ut, dt, xs_nom=st["xs"][b:b + 1], K=gs["K"][b:b + 1])[2][0])
                        with oracle.flavour(prec["yard"]):
                            c_y = float(oracle.batch_rollout(_tw(om, prec["yard"]), x0[b:b + 1], ut, dt, xs_nom=st["xs"][b:b + 1], K=gs["K"][b:b + 1])[2][0])
                        if not (np.isfinite(c_t) and np.isfinite(c_y)) or abs(c_t - c_y) > ROLLOUT_NOISE[precision] * abs(c_y):
                            noisy = True
                            break
                    if noisy:
                        out["amplified"] = out.get("amplified", 0) + 1
                        out["tied"].add(int(b))
                        pit["amplified"] += 1
                        continue
                assert np.any(np.abs(cand) <= prec["tie_rel"] * abs(st["cost"][b])), "line searches differ away from a tie (dcost %s) -- %s" % (dcost, where)
                out["ties_search"] += 1
                out["tied"].add(int(b))
                pit["search_tie"] += 1
                continue
            if g_status[b] != nx["status"][b]:
                dcost = st["cost"][b] - nx["cost"][b]
                near_tolfun = abs(dcost - p["tol_fun"]) <= prec["tie_rel"] * abs(st["cost"][b])
                near_lmax = abs(nx["lam"][b] - p["lambda_max"]) <= 1e-9 * p["lambda_max"]
                near_grad = 1 in (int(g_status[b]), int(nx["status"][b])) and abs(nx["gnorm"][b] - p["tol_grad"]) <= 10 * tol * p["tol_grad"]
                assert near_tolfun or near_lmax or near_grad, "terminations differ away from a tie -- " + where
                out["ties_stop"] += 1
                out["tied"].add(int(b))
                pit["stop_tie"] += 1
                continue
            # Same gains (within tol), alpha, status and lambda, but the new cost differs by more than tol: at the
            # BASELINE horizon (T = 499, swing-up scale) the closed-loop rollout amplifies a 1e-7 difference in the gains
            # beyond 1e-6 in the cost.  Proof that this is all it is: the oracle's OWN rollout of the device's gains
            # (same nominal, same alpha) must reproduce the device's cost, and its trajectory the device's.
            a = int(gs["alpha"][b])
            if lam_ok and a >= 0:
                from oracle.oracle import ALPHAS
                with oracle.flavour(prec["twin"]):
                    omt = _tw(om, prec["twin"])
                    xs_r, us_r, c_r = oracle.batch_rollout(omt, x0[b:b + 1], st["us"][b:b + 1] + ALPHAS[a] * gs["k"][b:b + 1], dt,
                                                           xs_nom=st["xs"][b:b + 1], K=gs["K"][b:b + 1])
                if not abs(gs["cost"][b] - float(c_r[0])) <= tol * abs(float(c_r[0])):
                    # two roundings of the SAME closed-loop rollout (same gains, same alpha) differ by more than tol: judge both
                    # against the rollout in the yardstick precision -- the device may be no further from it than 10 x the twin is
                    with oracle.flavour(prec["yard"]):
                        omy = _tw(om, prec["yard"])
                        yt = np.asarray(st["us"][b:b + 1] + ALPHAS[a] * gs["k"][b:b + 1])
                        xs_y, us_y, c_y = oracle.batch_rollout(omy, x0[b:b + 1], yt, dt, xs_nom=st["xs"][b:b + 1], K=gs["K"][b:b + 1])
                    c_y = float(c_y[0])
                    e_dev, e_twin = abs(gs["cost"][b] - c_y) / abs(c_y), abs(float(c_r[0]) - c_y) / abs(c_y)
                    # (single realisations of rounding noise: scripts/f32_rollout_check.py -- device and oracle float rollouts of the
                    #  same gains sit at the same distance from the fp64 rollout in distribution, median 2e-6, p90 6e-4 at T = 499,
                    #  and one is > 10 x the other in 4 % of the cases: those are counted with the > 10 x conditioning cases)
                    lax = max(COND_HARD * e_twin, ROLLOUT_NOISE.get(precision, 0.0))
                    assert e_dev <= max(tol, lax), "rollout of the device's own gains differs -- %s [vs yardstick: device %.2e, twin %.2e]" % (where, e_dev, e_twin)
                    out["cond_over10"] += int(e_dev > max(tol, COND_FACTOR * e_twin))
                out["amplified"] = out.get("amplified", 0) + 1
                out["worst_gain"] = max(out["worst_gain"], float(eg[b]))
                pit["amplified"] += 1
                continue
            raise AssertionError("same gains, alpha and status but cost / lambda differ -- " + where)
        if drive == "oracle":
            running &= nx["status"] == 0
            st = {kk: nx[kk] for kk in ("xs", "us", "k", "K", "cost", "lam", "dlam")}
        else:
            running &= gs["status"] == 0
            st = gs
    if aux is not None:
        aux.close()
    return out


def walk_both(oracle, om, g, x0, u0, dt, n_iters, **kw):
    """oracle-driven and device-driven walks; returns (trajectories that met a tie in either, the two results)."""
    ro = walk_iterations(oracle, om, g, x0, u0, dt, n_iters, drive="oracle", **kw)
    rg = walk_iterations(oracle, om, g, x0, u0, dt, n_iters, drive="gpu", **kw)
    return ro["tied"] | rg["tied"], ro, rg


AMPLIFIED = 100 * TOL


def assert_free_run(cost_dev, cost_orc, tied, what="", tol=TOL):
    """End-to-end comparison of two FREE-RUNNING solves (device vs oracle, each from its own state).  Every
    step of the device's run has been checked by the device-driven walk; what is left to verify here is
    that nothing accumulates: a trajectory without a tie agrees to 1e-6, or -- chaotic dynamics amplify
    last-bit differences from iteration to iteration, SURVEY.md 0.3 -- to 1e-4 for at most a few."""
    cost_orc = _f64(cost_orc)
    rel = np.abs(cost_dev - cost_orc) / np.maximum(np.abs(cost_orc), 1e-300)
    idx = np.flatnonzero(rel >= tol)
    free = [int(b) for b in idx if int(b) not in tied]
    assert len(free) <= max(1, len(rel) // 16), (what, free, rel[free])
    assert all(rel[b] < 100 * tol for b in free), (what, free, rel[free])
    return rel < tol


def _explained_by_records(oracle, om, prec, g, x0, st, b, gs, tol, plain_only=False):
    """The oracle's backward pass at the state's lambda, fed the DEVICE's derivative records of the state's nominal trajectory
    (computed by the stand-alone sweep kernel: the same finite-difference code the fused sweep runs), against the device's
    gains for trajectory b: True if it completes and agrees per knot to tol -- or to what the yardstick precision says fp
    conditioning allows on THESE records.  (A pass that needed a lambda retry is not explained here.)"""
    cache = getattr(g, "_records_of_state", None)
    if cache is None or cache[0] is not st:  # (one sweep per iteration of the walk, not one per trajectory)
        load_state(g, x0, st)
        g.compute_derivatives()
        g._records_of_state = cache = (st, g.derivatives())
    d = cache[1]
    derivs = {kk: _f64(v[b:b + 1] if kk in ("cx", "cu") else mat(v[b:b + 1])) for kk, v in d.items()}
    if plain_only:  # the records themselves against the oracle's, per knot, each block against its own scale over the trajectory
        with oracle.flavour(prec["twin"]):
            ro = oracle.batch_derivatives(_tw(om, prec["twin"]), st["xs"][b:b + 1], st["us"][b:b + 1], g.dt)
        for kk, dv in derivs.items():
            ov = _f64(ro[kk])
            if np.abs(dv - ov).max() > tol * max(np.abs(ov).max(), 1.0):
                return False
    us_b, kp_b, lam_b = st["us"][b:b + 1], st["k"][b:b + 1], st["lam"][b:b + 1]
    with oracle.flavour(prec["twin"]):
        r = oracle.batch_backward(_tw(om, prec["twin"]), us_b, derivs, k_prev=kp_b, lam=lam_b)
    if int(r["diverge"][0]) != 0:
        return False
    ko, Ko = _f64(r["k"]), _f64(mat(r["K"]))
    if gains_knot_err(gs["k"][b:b + 1], gs["K"][b:b + 1], ko, Ko, us_b)[0] < tol:
        return True
    if plain_only:
        return False
    k80, K80, div80 = backward_f80(oracle, om, us_b, derivs, kp_b, lam_b, yard=prec["yard"])
    e_dev, e_orc, okc = conditioning_verdict(gs["k"][b:b + 1], gs["K"][b:b + 1], ko, Ko, k80, K80, us_b, tol)
    return bool(okc[0])


def _candidate_costs(g, x0, st, b):
    load_state(g, x0, st)
    g.compute_derivatives()
    g.backward_step()
    return g.rollout_candidates()[b]


def publish(name, r, **meta):
    """Adds the per-iteration bins of a walk to the tracked statistics file (copied to profiles/parity_r06.json from the GPU
    run): which share of the checked trajectory-iterations met the plain 1e-6 criterion and which went through which proof."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("ILQR_PARITY_JSON", os.path.join(root, "gpurun_out", "parity_r06.json"))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    doc = json.load(open(path)) if os.path.exists(path) else {}
    tot = {kk: sum(p[kk] for p in r["per_iter"]) for kk in r["per_iter"][0] if kk != "iteration"} if r["per_iter"] else {}
    doc[name] = dict(meta, tolerance=PRECISIONS[meta.get("precision", "f64")]["tol"], gain_tolerance=PRECISIONS[meta.get("precision", "f64")]["gtol"], checked=r["checked"], totals=tot,
                     plain_fraction=(tot["plain"] / max(tot["n"], 1)) if tot else None,
                     plain_or_plain_on_device_records_fraction=((tot["plain"] + tot["plain_on_device_records"]) / max(tot["n"], 1)) if tot else None,
                     worst_cond_ratio=r["worst_cond_ratio"], worst_gain_err_plain=r["worst_gain"], worst_cost_err_plain=r["worst_cost"],
                     per_iteration=r["per_iter"])
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    return doc[name]


def assert_walk(r, n_iters, min_plain_it0=0.95, tied_div=16, over10_div=24, min_plain=None, max_on_records=0.30, max_amplified=None):
    """The bounds every full-size walk is held to: every sampled trajectory checked in every iteration it ran; at iteration 0
    (lambda = 1) at least 95 % of them within the PLAIN tolerance -- end to end, or stage by stage (records to tol, and the
    oracle's backward pass on the device's records to tol: the two sides' finite differences are different realisations of
    1e-13 cancellation noise, which the T = 499 recursion amplifies beyond 1e-6 for a good half of full-scale problems);
    proven ties and > 10 x conditioning cases counted and bounded."""
    p0 = r["per_iter"][0]
    assert p0["n"] == len(r["sel"]) and p0["plain"] + p0["plain_on_device_records"] >= min_plain_it0 * p0["n"], p0
    assert sum(p["n"] for p in r["per_iter"]) == r["checked"] and r["checked"] >= len(r["sel"]) * min(n_iters, 3), r["checked"]
    assert r["cond_over10"] <= max(2, r["checked"] // over10_div), r["cond_over10"]
    ties = sum(p["knife_edge"] + p["search_tie"] + p["stop_tie"] for p in r["per_iter"])
    assert ties <= max(2, r["checked"] // tied_div), ties
    # The strict END-TO-END share over the whole walk, kept apart from the stage-by-stage one: it may not fall below the baseline recorded
    # for this configuration (profiles/parity_r05.json minus a sampling margin; the caller states it), and the share that needs the
    # "plain on the device's records" argument is bounded -- that bin must not quietly absorb a regression of the kernels.
    n_all = max(1, sum(p["n"] for p in r["per_iter"]))
    plain = sum(p["plain"] for p in r["per_iter"]) / n_all
    on_records = sum(p["plain_on_device_records"] for p in r["per_iter"]) / n_all
    if min_plain is not None:
        assert plain >= min_plain, ("end-to-end plain share below the recorded baseline", plain, min_plain)
    assert on_records <= max_on_records, ("plain-on-device-records share", on_records, max_on_records)
    # "amplified": same gains and alpha, a T-step closed-loop rollout that amplified rounding (proven per case by the twin's own rollout of the
    # device's gains).  It has a bound of its own, like the tie bins: a kernel regression that perturbs rollouts must not be absorbed here.
    # Recorded (profiles/parity_r05.json / r06): fp64 walks 0 - 1.4 % of the checked trajectory-iterations, fp32 walks 6 - 8 %.
    amplified = sum(p.get("amplified", 0) for p in r["per_iter"])
    cap = max_amplified if max_amplified is not None else max(2, r["checked"] // 50)  # (fp32 callers pass checked // 8)
    assert amplified <= cap, ("rollout-amplification bin", amplified, cap)


class Sampled:
    """A window onto trajectories `sel` of a FULL-SIZE handle, with the interface walk_iterations(drive="gpu") uses.
    The big batch runs freely on the device (init_traj, then iterate(1) again and again, all B trajectories); the
    oracle is given the sampled trajectories' state before every iteration and must reproduce their next state --
    i.e. the kernels are compared with the oracle AT the BASELINE size (grid, tiling, occupancy, route selection of
    the full batch), for as many trajectories as the oracle walks in seconds."""

    def __init__(self, g, sel, x0_full, u0_full):
        self.g, self.sel, self.x0_full, self.u0_full = g, np.asarray(sel), x0_full, u0_full

    def init_traj(self, x0, u0):
        assert np.array_equal(x0, self.x0_full[self.sel])
        return self.g.init_traj(self.x0_full, self.u0_full)[self.sel]

    def iterate(self, n=1):
        self.g.iterate(n)

    def trajectory(self):
        xs, us = self.g.trajectory()
        return xs[self.sel], us[self.sel]

    def gains(self):
        k, K = self.g.gains()
        return k[self.sel], K[self.sel]

    def lambdas(self):
        lam, dlam = self.g.lambdas()
        return lam[self.sel], dlam[self.sel]

    def status(self):
        return tuple(a[self.sel] for a in self.g.status())

    def cost(self):
        return self.g.cost()[self.sel]

    def gnorm(self):
        return self.g.gnorm()[self.sel]

    def dV(self):
        return self.g.dV()[self.sel]

    def clone(self):
        from ilqr_amd import BatchILQR
        kw = dict(self.g._ctor)
        kw["B"] = len(self.sel)
        return BatchILQR(**kw)


def sampled_walk(oracle, om, g, x0, u0, dt, n_iters, n_sample=64, seed=99, **kw):
    """Device-driven walk (see walk_iterations) of n_sample random trajectories of a full-size batch that runs freely
    on `g`.  Always includes the first and last trajectory (first / last tile, possibly ragged)."""
    B = x0.shape[0]
    rng = np.random.default_rng(seed)
    sel = np.unique(np.concatenate([[0, B - 1], rng.choice(B, size=min(n_sample, B) - 2, replace=False)]))
    r = walk_iterations(oracle, om, Sampled(g, sel, x0, u0), x0[sel], u0[sel], dt, n_iters, drive="gpu", **kw)
    r["sel"] = sel
    return r
