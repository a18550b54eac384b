"""Whole iLQR iterations on the device against the oracle, with per-iteration teacher forcing and a
PROOF OF TIE for every trajectory that takes another branch (tests/parity.py): no "most of the batch
agrees" thresholds.  Also: get_gradient_norm and the gradient-norm exit (status 1) on both sides."""
import numpy as np
import pytest

from tests.parity import gains_knot_err, walk_iterations
from tests.util import TOL, acrobot_x0, integrator_x0

pytestmark = pytest.mark.gpu
DT = 0.02


def build(oracle, name, B, T, lim, params=None):
    from ilqr_amd import BatchILQR
    if name == "acrobot":
        om = oracle.Model("acrobot", u_lim=lim)
        g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, params=params)
    else:
        goal = [1.0, 0.5, 0.0, 0.0]
        om = oracle.Model("integrator", goal=goal, u_lim=lim)
        g = BatchILQR("integrator", B, T, DT, u_min=-lim, u_max=lim, goal=goal, params=params)
    return om, g


def assert_rare(r):
    """Proven ties stay rare (a tie needs a control within 1e-4 of a bound with a noise-sized gradient, or a
    cost change of rounding size), and so do steps whose per-knot gain agreement is limited by fp64
    conditioning by more than 10x the fp64 oracle's own distance from extended precision."""
    ties = r["ties_backward"] + r["ties_search"] + r["ties_stop"]
    assert ties <= max(2, r["checked"] // 20), r
    assert r["cond_over10"] <= max(1, r["checked"] // 50), r


CASES = [
    # name, B, T, limit, x0 scale, iterations
    ("acrobot", 64, 120, 1.5, 1.0, 8),     # the bench workload's regime: clamps active, chaotic
    ("acrobot", 48, 200, 5.0, 0.3, 6),
    ("acrobot", 32, 499, 5.0, 0.01, 5),    # the small-scale stable regime at the headline horizon
    ("integrator", 33, 99, 0.5, 1.0, 16),  # runs into lambda growth and the cost-change exit
    ("integrator", 19, 7, 2.0, 1.0, 4),    # T < candidate chunk
]


@pytest.mark.parametrize("name,B,T,lim,scale,iters", CASES)
def test_iterations_teacher_forced(oracle, name, B, T, lim, scale, iters):
    om, g = build(oracle, name, B, T, lim)
    x0 = acrobot_x0(B, scale=scale) if name == "acrobot" else integrator_x0(B)
    u0 = np.zeros((B, T, om.nu))
    r = walk_iterations(oracle, om, g, x0, u0, DT, iters)
    g.close()
    print(name, r)
    assert r["checked"] >= B * min(iters, 2)
    assert_rare(r)


def test_iterations_teacher_forced_fixed_work(oracle):
    """bench mode (ILQR_FLAG_FIXED_WORK): same walk, terminations disabled on both sides, across the two
    routes of ilqr_iterate (fused sweep+backward kernel and two kernels)."""
    from ilqr_amd import BatchILQR, capi
    B, T, lim = 40, 131, 1.5
    x0 = acrobot_x0(B, seed=3)
    for extra in (0, capi.FLAG_UNFUSED):
        om = oracle.Model("acrobot", u_lim=lim)
        g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, flags=capi.FLAG_FIXED_WORK | extra)
        r = walk_iterations(oracle, om, g, x0, np.zeros((B, T, 1)), DT, 6, fixed_work=True)
        g.close()
        assert r["checked"] == 6 * B, r
        assert_rare(r)


def test_gradient_norm_exit(oracle):
    """status 1 ("gradient norm < tolGrad", ilqr_core.cpp:153-159) on both sides.  With the reference's
    tolGrad = 1e-6 the cost-change exit always fires first on the shipped models, so the test loosens
    tolGrad (an ilqr_params field here, a compile-time constant there): the exit then needs lambda < 1e-5,
    i.e. seven accepted iterations, and the gradient norm below the threshold at the start of one."""
    from ilqr_amd import BatchILQR
    B, T, lim = 32, 100, 5.0
    tol_grad = 5e-2
    params = dict(tol_grad=tol_grad)
    om, g = build(oracle, "acrobot", B, T, lim, params=params)
    x0 = acrobot_x0(B, scale=0.3, seed=5)
    u0 = np.zeros((B, T, 1))
    oracle.set_params(tol_grad=tol_grad)
    try:
        r = walk_iterations(oracle, om, g, x0, u0, DT, 12, params=params)
        # free-running on both sides as well: same exits
        g.init_traj(x0, u0)
        g.generate_trajectory()
        ro = oracle.batch_solve(om, x0, u0, DT)
    finally:
        oracle.set_params()
    st, it, al = g.status()
    gn = g.gnorm()
    lam, _ = g.lambdas()
    g.close()
    assert (ro["status"] == 1).sum() >= B // 4, np.bincount(ro["status"])
    both = (st == 1) & (ro["status"] == 1)
    assert both.sum() >= (ro["status"] == 1).sum() * 3 // 4
    assert np.array_equal(it[both], ro["iters"][both])
    assert np.all(gn[st == 1] < tol_grad) and np.all(lam[st == 1] < 1e-5)
    assert_rare(r)


def test_gnorm_matches_oracle(oracle):
    """get_gradient_norm (ilqr_core.cpp:405-412) after one STEP 2 on identical state, both backward kernels."""
    from ilqr_amd import BatchILQR, capi
    B, T, lim = 50, 90, 1.5
    om = oracle.Model("acrobot", u_lim=lim)
    x0 = acrobot_x0(B, seed=9)
    rng = np.random.default_rng(1)
    u0 = rng.normal(size=(B, T, 1))
    xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
    nx = oracle.batch_iterate_from(om, x0, xs, us, np.zeros((B, T, 1)), np.zeros((B, T, 1, 4)), cost, 1.0, 1.0, DT,
                                   n_iters=1, fixed_work=True)
    for flags in (0, capi.FLAG_BACKWARD_THREAD_PER_TRAJ):
        g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, flags=flags)
        g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
        g.compute_derivatives()
        g.backward_step()
        k, K = g.gains()
        gn = g.gnorm()
        g.close()
        ok = gains_knot_err(k, K, nx["k"], nx["K"], us) < TOL
        assert ok.mean() > 0.9
        assert np.allclose(gn[ok], nx["gnorm"][ok], rtol=1e-9, atol=0), np.abs(gn[ok] / nx["gnorm"][ok] - 1).max()
