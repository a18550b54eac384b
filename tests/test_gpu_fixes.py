"""ILQR_FLAG_REFERENCE_FIXES (opt-in): the two things the reference's own comments call for --
the clamped rollout (src/ilqr_core.cpp:327-329, "This is the right way") and a box-QP that reports a failed
Cholesky factorisation (src/boxqp.cpp:85-88 ignores info()) -- against the oracle with the same switch
(orc_set_fixes), and the property they exist for: the controls of a solve respect their limits."""
import numpy as np
import pytest

from tests.parity import check_backward, walk_iterations
from tests.util import acrobot_x0, integrator_x0, mat

pytestmark = pytest.mark.gpu
DT = 0.02


@pytest.fixture
def fixed_oracle(oracle):
    oracle.set_fixes(3)
    yield oracle
    oracle.set_fixes(0)


def build(oracle, name, B, T, lim, flags):
    from ilqr_amd import BatchILQR
    if name == "acrobot":
        return oracle.Model("acrobot", u_lim=lim), BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, flags=flags), acrobot_x0(B, scale=0.5)
    goal = [1.0, 0.5, 0.0, 0.0]
    return (oracle.Model("integrator", goal=goal, u_lim=lim), BatchILQR("integrator", B, T, DT, u_min=-lim, u_max=lim, goal=goal, flags=flags),
            integrator_x0(B))


@pytest.mark.parametrize("name,B,T,lim,iters", [("acrobot", 48, 150, 1.5, 6), ("integrator", 33, 60, 0.5, 6)])
def test_iterations_with_the_fixes_match_the_oracle(fixed_oracle, name, B, T, lim, iters):
    from ilqr_amd import capi
    oracle = fixed_oracle
    om, g, x0 = build(oracle, name, B, T, lim, capi.FLAG_REFERENCE_FIXES)
    u0 = np.zeros((B, T, om.nu))
    r = walk_iterations(oracle, om, g, x0, u0, DT, iters)
    print("fixes walk", name, {k: v for k, v in r.items() if k != "tied"})
    assert r["checked"] >= 2 * B and len(r["tied"]) <= max(2, r["checked"] // 10), r
    # the point of the clamped rollout: the solve's controls stay inside their box ...
    g.init_traj(x0, u0)
    g.iterate(iters)
    _, us = g.trajectory()
    assert np.abs(us).max() <= lim
    g.close()
    # ... which the reference's own rollout does not guarantee (README.md:9, "control-limited part not working")
    oracle.set_fixes(0)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=iters)
    oracle.set_fixes(3)
    if name == "acrobot":
        assert np.abs(ro["us"]).max() > lim


@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_indefinite_quu_is_reported_as_divergence(fixed_oracle, name):
    """Quu made indefinite at a few knots (cuu shifted negative, lambda = 0), free controls: with the fix the box-QP
    returns -1 there and backward_pass returns that knot (ilqr_core.cpp:371) -- same knot on both sides; without the
    fix both sides carry on with Eigen's partial factor (covered by the default tests)."""
    from ilqr_amd import capi
    oracle = fixed_oracle
    B, T, lim = 24, 40, 50.0   # wide limits: nothing is clamped, the factorisation is reached
    om, g, x0 = build(oracle, name, B, T, lim, capi.FLAG_REFERENCE_FIXES)
    rng = np.random.default_rng(5)
    u0 = rng.normal(size=(B, T, om.nu)) * 0.2
    xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
    do = oracle.batch_derivatives(om, xs, us, DT)
    bad_t = rng.integers(3, T - 3, size=B)
    for b in range(0, B, 2):  # every other trajectory gets one indefinite knot
        do["cuu"][b, bad_t[b]] -= 5.0 * np.eye(om.nu)
    k_prev = np.zeros((B, T, om.nu))
    ro = oracle.batch_backward(om, us, do, k_prev=k_prev, lam=0.0)
    assert (ro["diverge"][0::2] == bad_t[0::2]).mean() > 0.7 and np.all(ro["diverge"][1::2] == 0)
    g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    g.set_gains(k=k_prev, K=np.zeros((B, T, om.nu, om.nx)))
    g.set_lambda(0.0, 1.0)
    div = g.backward_pass()
    assert np.array_equal(div, ro["diverge"])
    k, K = g.gains()
    check_backward(oracle, om, us, do, k_prev, 0.0, k, K, g.dV(), div, ro, max_ties=2, max_over10=1)
    # STEP 2 as a whole: the retry loop raises lambda until the pass goes through (ilqr_core.cpp:136-150)
    g.backward_step()
    lam, _ = g.lambdas()
    assert np.all(lam[0::2][ro["diverge"][0::2] > 0] > 0) and np.all(lam[1::2] == 0)
    g.close()


def test_the_flag_is_off_by_default_and_rejected_where_not_implemented():
    from ilqr_amd import BatchILQR, capi
    from ilqr_amd.capi import ILQRError
    # (a host-evaluated model takes the flag since ABI 5: the failed-factorisation exit is the device's, the clamped rollout the caller's --
    #  tests/test_cpp_facade.py::test_user_twin_with_its_own_dimensions_against_host_virtuals runs the facade's)
    BatchILQR("host", 2, 5, DT, nx=3, nu=2, u_min=[-1, -1], u_max=[1, 1], flags=capi.FLAG_REFERENCE_FIXES).close()
    from tests.test_gpu_lq_end_to_end import dense_mats
    with pytest.raises(ILQRError, match="REGULARIZE_VXX"):  # (the generic path has it in k_backward_w3 only)
        BatchILQR("lq", 2, 5, DT, lq=dense_mats(6, 3), u_min=-1.0, u_max=1.0, flags=capi.FLAG_REGULARIZE_VXX, route=capi.ROUTE_BACKWARD_W2)


@pytest.mark.parametrize("n,m,B,T,lim,route", [(6, 3, 30, 40, 0.15, 0), (32, 16, 5, 20, 0.1, 0), (6, 3, 30, 40, 0.15, "thread")])
def test_generic_path_iterations_with_the_fixes_match_the_oracle(fixed_oracle, n, m, B, T, lim, route):
    """ILQR_FLAG_REFERENCE_FIXES on the generic path (LQ twin: matrix-core rollouts k_rollout_lq and, forced, the thread-per-rollout
    kernel every user twin runs): clamped rollouts + a box-QP that reports a failed factorisation, iteration by iteration against the
    oracle with the same switch; the controls of the solve stay inside their box."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    oracle = fixed_oracle
    mats = dense_mats(n, m)
    om = oracle.Model("lq", lq=mats, u_lim=lim)
    g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_REFERENCE_FIXES,
                  route=capi.ROUTE_LQ_THREAD_ROLLOUT if route == "thread" else 0)
    rng = np.random.default_rng(21)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.3   # beyond the limits: the init rollout clamps too
    c0 = g.init_traj(x0, u0)
    xs_o, us_o, c_o = oracle.batch_rollout(om, x0, u0, DT)
    assert np.abs(us_o).max() <= lim and np.max(np.abs(c0 - c_o) / np.abs(c_o)) < 1e-12
    r = walk_iterations(oracle, om, g, x0, u0, DT, 5)
    assert r["checked"] >= 2 * B and len(r["tied"]) <= max(2, r["checked"] // 10), r
    g.init_traj(x0, u0)
    g.iterate(5)
    _, us = g.trajectory()
    assert np.abs(us).max() <= lim
    g.close()


def test_generic_path_indefinite_quu_is_reported_as_divergence(fixed_oracle):
    """As test_indefinite_quu_is_reported_as_divergence, through the generic backward kernel's own box-QP (k_backward_w3: the matrix-core
    refinement never accepts an indefinite block, the literal path it falls back to returns -1 with the fix)."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    oracle = fixed_oracle
    n, m, B, T, lim = 6, 3, 24, 30, 50.0
    mats = dense_mats(n, m)
    om = oracle.Model("lq", lq=mats, u_lim=lim)
    g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_REFERENCE_FIXES)
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.2
    xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
    do = oracle.batch_derivatives(om, xs, us, DT)
    bad_t = rng.integers(3, T - 3, size=B)
    for b in range(0, B, 2):
        do["cuu"][b, bad_t[b]] -= 5.0 * np.eye(m)
    k_prev = np.zeros((B, T, m))
    ro = oracle.batch_backward(om, us, do, k_prev=k_prev, lam=0.0)
    assert (ro["diverge"][0::2] == bad_t[0::2]).mean() > 0.7 and np.all(ro["diverge"][1::2] == 0)
    g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    g.set_gains(k=k_prev, K=np.zeros((B, T, m, n)))
    g.set_lambda(0.0, 1.0)
    div = g.backward_pass()
    assert np.array_equal(div, ro["diverge"])
    g.close()


@pytest.mark.parametrize("name,B,T,lim", [("acrobot", 40, 120, 5.0), ("integrator", 33, 60, 0.5)])
def test_vxx_regularisation_matches_the_oracle(oracle, name, B, T, lim):
    """ILQR_FLAG_REGULARIZE_VXX: lambda on Vxx' ([Tassa 2012] eq. 10) instead of on Quu, both sides (orc_set_fixes(4)):
    teacher-forced backward passes at lambda in {1, 10} and whole iterations."""
    from ilqr_amd import capi
    oracle.set_fixes(4)
    try:
        om, g, x0 = build(oracle, name, B, T, lim, capi.FLAG_REGULARIZE_VXX)
        rng = np.random.default_rng(3)
        u0 = rng.normal(size=(B, T, om.nu)) * 0.3
        xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
        do = oracle.batch_derivatives(om, xs, us, DT)
        for lam in (1.0, 10.0):
            ro = oracle.batch_backward(om, us, do, lam=lam)
            g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
            g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
            g.set_gains(k=np.zeros((B, T, om.nu)), K=np.zeros((B, T, om.nu, om.nx)))
            g.set_lambda(lam, 1.0)
            div = g.backward_pass()
            k, K = g.gains()
            check_backward(oracle, om, us, do, np.zeros((B, T, om.nu)), lam, k, K, g.dV(), div, ro, max_ties=max(1, B // 16), max_over10=max(1, B // 50))
            # and it IS a different regularisation: the gains differ from the reference's lambda I on Quu
            oracle.set_fixes(0)
            r0 = oracle.batch_backward(om, us, do, lam=lam)
            oracle.set_fixes(4)
            assert np.abs(mat(r0["K"]) - mat(ro["K"])).max() > 1e-3 * np.abs(mat(ro["K"])).max()
        r = walk_iterations(oracle, om, g, x0, np.zeros((B, T, om.nu)), DT, 4)
        assert r["checked"] >= 2 * B and len(r["tied"]) <= max(2, r["checked"] // 10), r
        g.close()
    finally:
        oracle.set_fixes(0)


@pytest.mark.parametrize("n,m,B,T,lim", [(6, 3, 30, 40, 0.15), (32, 16, 5, 20, 0.1), (5, 2, 20, 30, 0.15), (20, 7, 6, 25, 0.1)])
def test_generic_path_vxx_regularisation_matches_the_oracle(oracle, n, m, B, T, lim):
    """ILQR_FLAG_REGULARIZE_VXX on the generic path (k_backward_w3<.., REGV>: fu'fu and fu'fx as two more transposed products on the
    matrix cores): teacher-forced backward passes at lambda in {1, 10} against the oracle with orc_set_fixes(4) -- one 16 x 16 tile and
    two, the scalar box-QPs of m <= 2, sizes that are not multiples of a tile -- and whole iterations."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    oracle.set_fixes(4)
    try:
        mats = dense_mats(n, m)
        om = oracle.Model("lq", lq=mats, u_lim=lim)
        g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_REGULARIZE_VXX)
        rng = np.random.default_rng(8)
        x0 = rng.uniform(-1, 1, (B, n))
        u0 = rng.normal(size=(B, T, m)) * 0.1
        xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
        do = oracle.batch_derivatives(om, xs, us, DT)
        for lam in (1.0, 10.0):
            ro = oracle.batch_backward(om, us, do, lam=lam)
            g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
            g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
            g.set_gains(k=np.zeros((B, T, m)), K=np.zeros((B, T, m, n)))
            g.set_lambda(lam, 1.0)
            div = g.backward_pass()
            k, K = g.gains()
            check_backward(oracle, om, us, do, np.zeros((B, T, m)), lam, k, K, g.dV(), div, ro, max_ties=max(1, B // 8), max_over10=max(1, B // 10))
            oracle.set_fixes(0)  # ... and it is another regularisation than lambda I on Quu
            r0 = oracle.batch_backward(om, us, do, lam=lam)
            oracle.set_fixes(4)
            assert np.abs(mat(r0["K"]) - mat(ro["K"])).max() > 1e-3 * np.abs(mat(ro["K"])).max()
        r = walk_iterations(oracle, om, g, x0, u0, DT, 4)
        assert r["checked"] >= 2 * B and len(r["tied"]) <= max(2, r["checked"] // 10), r
        g.close()
    finally:
        oracle.set_fixes(0)


def test_generic_path_vxx_regularisation_with_exact_derivatives(oracle):
    """With ILQR_FLAG_ANALYTIC_DERIVATIVES the LQ handle leaves its record-free route under the flag (per-knot cx / cu records + the
    constant blocks): the backward pass over the handle's own sweep against the oracle's over the same records."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    n, m, B, T, lim = 32, 16, 5, 20, 0.1
    oracle.set_fixes(4)
    try:
        mats = dense_mats(n, m)
        om = oracle.Model("lq", lq=mats, u_lim=lim)
        g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=capi.FLAG_REGULARIZE_VXX | capi.FLAG_ANALYTIC_DERIVATIVES)
        rng = np.random.default_rng(10)
        x0 = rng.uniform(-1, 1, (B, n))
        u0 = rng.normal(size=(B, T, m)) * 0.1
        g.init_traj(x0, u0)
        g.compute_derivatives()
        g.set_lambda(3.0, 1.0)
        div = g.backward_pass()
        k, K = g.gains()
        d = g.derivatives()
        _, us = g.trajectory()
        do = {kk: np.ascontiguousarray(vv if kk in ("cx", "cu") else np.swapaxes(vv, -1, -2)) for kk, vv in d.items()}
        ro = oracle.batch_backward(om, us, do, lam=3.0)
        check_backward(oracle, om, us, do, np.zeros((B, T, m)), 3.0, k, K, g.dV(), div, ro, max_ties=1, max_over10=1)
        g.iterate(3)
        assert np.all(np.isfinite(g.cost()))
        g.close()
    finally:
        oracle.set_fixes(0)


def test_host_model_backward_pass_with_vxx_regularisation(oracle):
    """A host-evaluated model's handle runs only the backward pass on the device: ILQR_FLAG_REGULARIZE_VXX applies to it as to any model."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    n, m, B, T, lim = 7, 3, 12, 30, 0.2
    oracle.set_fixes(4)
    try:
        mats = dense_mats(n, m)
        om = oracle.Model("lq", lq=mats, u_lim=lim)
        g = BatchILQR("host", B, T, DT, nx=n, nu=m, u_min=[-lim] * m, u_max=[lim] * m, flags=capi.FLAG_REGULARIZE_VXX)
        rng = np.random.default_rng(9)
        x0 = rng.uniform(-1, 1, (B, n))
        u0 = rng.normal(size=(B, T, m)) * 0.1
        xs, us, cost = oracle.batch_rollout(om, x0, u0, DT)
        do = oracle.batch_derivatives(om, xs, us, DT)
        ro = oracle.batch_backward(om, us, do, lam=2.0)
        g.set_trajectory(x0=x0, xs=xs, us=us, cost=cost)
        g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
        g.set_gains(k=np.zeros((B, T, m)), K=np.zeros((B, T, m, n)))
        g.set_lambda(2.0, 1.0)
        div = g.backward_pass()
        k, K = g.gains()
        check_backward(oracle, om, us, do, np.zeros((B, T, m)), 2.0, k, K, g.dV(), div, ro, max_ties=2, max_over10=2)
        g.close()
    finally:
        oracle.set_fixes(0)
