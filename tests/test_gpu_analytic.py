"""ILQR_FLAG_ANALYTIC_DERIVATIVES (opt-in): the device models' exact derivatives
against the finite-difference sweep (which is what the reference computes and what the parity
tests pin).  The two must agree to the finite differences' own error: truncation O(eps^2 f''') on the
first derivatives, rounding noise ~1e-16 f / eps^2 on the second ones."""
import numpy as np
import pytest

from tests.util import acrobot_x0, integrator_x0

pytestmark = pytest.mark.gpu
DT = 0.02


def _records(model, flags, x0, u0, **kw):
    from ilqr_amd import BatchILQR
    B, T = u0.shape[0], u0.shape[1]
    g = BatchILQR(model, B, T, DT, flags=flags, **kw)
    g.init_traj(x0, u0)
    g.compute_derivatives()
    d = g.derivatives()
    g.close()
    return d


def _cmp(fd, an, T, tol_first, tol_second):
    for name in ("fx", "fu", "cx", "cu"):
        scale = max(np.abs(fd[name]).max(), 1.0)
        assert np.abs(fd[name] - an[name]).max() <= tol_first * scale, name
    for name in ("cxx", "cuu"):
        scale = max(np.abs(fd[name]).max(), 1.0)
        assert np.abs(fd[name] - an[name]).max() <= tol_second * scale, name
    # cxu[T] is the reference's own "wrong" formula (pure rounding noise, never consumed); analytic: 0
    scale = max(np.abs(fd["cxu"]).max(), 1.0)
    assert np.abs(fd["cxu"][:, :T] - an["cxu"][:, :T]).max() <= tol_second * scale
    assert np.all(an["fx"][:, T] == 0) and np.all(an["fu"][:, T] == 0) and np.all(an["cu"][:, T] == 0)


def test_acrobot_exact_derivatives_match_finite_differences():
    from ilqr_amd import capi
    B, T = 40, 60
    x0 = acrobot_x0(B, seed=3)
    u0 = np.random.default_rng(4).normal(size=(B, T, 1)) * 2.0
    fd = _records("acrobot", 0, x0, u0, u_min=-5.0, u_max=5.0)
    an = _records("acrobot", capi.FLAG_ANALYTIC_DERIVATIVES, x0, u0, u_min=-5.0, u_max=5.0)
    # SURVEY.md 8c anchor values hold for the exact derivatives too
    _cmp(fd, an, T, 2e-6, 1e-5)
    z = _records("acrobot", capi.FLAG_ANALYTIC_DERIVATIVES, np.zeros((1, 4)), np.zeros((1, 3, 1)))
    assert np.allclose(z["fx"][0, 0], [[1, 0, .02, 0], [0, 1, 0, .02], [-0.140143, 0.0280285, 1, 0], [0.112114, -0.140143, 0, 1]], atol=2e-6)
    assert np.allclose(z["fu"][0, 0].ravel(), [0, 0, -0.0171429, 0.0457143], atol=1e-7)
    assert abs(z["cuu"][0, 0, 0, 0] - 0.02) < 1e-15 and np.allclose(z["cxx"][0, 3], 800 * np.eye(4))


def test_integrator_exact_derivatives_match_finite_differences():
    from ilqr_amd import capi
    B, T = 20, 30
    x0 = integrator_x0(B)
    u0 = np.random.default_rng(5).normal(size=(B, T, 2)) * 0.3
    kw = dict(goal=[1.0, 0.5, 0.0, 0.0])
    fd = _records("integrator", 0, x0, u0, **kw)
    an = _records("integrator", capi.FLAG_ANALYTIC_DERIVATIVES, x0, u0, **kw)
    _cmp(fd, an, T, 1e-9, 1e-6)  # linear dynamics, quadratic costs: only rounding separates the two


@pytest.mark.parametrize("n,m", [(32, 16), (6, 3), (5, 2), (31, 16)])  # (odd nx: 8-byte stores in k_analytic_lq)
def test_lq_exact_derivatives_match_finite_differences(n, m):
    from ilqr_amd import capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    B, T = 5, 8
    mats = dense_mats(n, m)
    rng = np.random.default_rng(6)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.3
    kw = dict(lq=mats, u_min=-1.0, u_max=1.0)
    fd = _records("lq", 0, x0, u0, **kw)
    an = _records("lq", capi.FLAG_ANALYTIC_DERIVATIVES, x0, u0, **kw)
    _cmp(fd, an, T, 1e-9, 1e-6)


def test_solves_with_exact_derivatives_reach_the_same_optimum():
    """The double integrator is linear-quadratic: with exact derivatives the solve takes the same
    path up to rounding; the acrobot (x0 = 0 canonical problem) must still descend."""
    from ilqr_amd import BatchILQR, capi
    B, T = 24, 99
    x0 = integrator_x0(B)
    u0 = np.zeros((B, T, 2))
    costs = []
    for fl in (0, capi.FLAG_ANALYTIC_DERIVATIVES):
        g = BatchILQR("integrator", B, T, DT, goal=[1.0, 0.5, 0.0, 0.0], flags=fl)
        g.generate_trajectory(x0, u0)
        assert g.count_running() == 0
        costs.append(g.cost())
        g.close()
    assert np.max(np.abs(costs[0] - costs[1]) / costs[0]) < 1e-3
    g = BatchILQR("acrobot", 2, 499, DT, flags=capi.FLAG_ANALYTIC_DERIVATIVES)
    c0 = g.init_traj(np.zeros((2, 4)), np.zeros((2, 499, 1)))
    g.generate_trajectory()
    c = g.cost()
    assert np.all(c < 0.01 * c0) and abs(c[0] - c[1]) == 0  # FD solve: 5.40 from 3947.6
    g.close()


@pytest.mark.parametrize("n,m", [(32, 16), (5, 2)])
def test_lq_partial_records_equal_whole_records(n, m, monkeypatch):
    """With exact derivatives the LQ model's record is constant up to cx, cu.  Three routes:
      * ILQR_ROUTE_BACKWARD_W2: the sweep writes the matrices once (const_rec) and per knot only cx, cu; k_backward_w2 reads the shared copy;
      * ... | ILQR_ROUTE_FULL_RECORDS: whole per-knot records -- the same solve BIT FOR BIT, the same records from the getter (which fills
        the matrices in on demand), also after ilqr_set_derivatives replaced them;
      * the default (k_backward_w3, fused): no sweep and no record array; the backward pass forms cx = cxx x_t, cu = cuu u_t itself, in
        another summation order: the same solve to rounding.  Its getter computes the records of the CURRENT nominal on demand (the other
        two return what the last sweep left, one accepted step behind, as the reference's members are), so cx / cu are compared with
        cxx x / cuu u of the returned trajectory instead."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    B, T = 9, 17
    mats = dense_mats(n, m)
    out = []
    for route in (capi.ROUTE_BACKWARD_W2, capi.ROUTE_BACKWARD_W2 | capi.ROUTE_FULL_RECORDS, 0):
        rng = np.random.default_rng(16)
        x0 = rng.uniform(-1, 1, (B, n))
        u0 = rng.normal(size=(B, T, m)) * 0.3
        g = BatchILQR("lq", B, T, DT, u_min=-0.4, u_max=0.4, lq=mats, flags=capi.FLAG_ANALYTIC_DERIVATIVES, route=route)
        assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("backward")) == (b"k_backward_w3" if route == 0 else b"k_backward_w2")
        g.init_traj(x0, u0)
        assert all(np.all(vv == 0) for vv in g.derivatives().values())  # what init_traj leaves (ilqr_core.cpp:39-45), on every route
        g.iterate(3)
        d = g.derivatives()
        xs, us = g.trajectory()
        k, K = g.gains()
        # caller-provided records replace the model's: the backward pass must read them knot by knot
        d2 = {kk: np.array(vv) for kk, vv in d.items()}
        d2["fx"] = d2["fx"] * (1 + 0.01 * rng.standard_normal((B, T + 1, 1, 1)))
        g.set_derivatives(**d2)
        g.set_lambda(1.0, 1.0)
        g.backward_pass()
        k2, K2 = g.gains()
        out.append(dict(xs=xs, us=us, k=k, K=K, cost=g.cost(), k2=k2, K2=K2, **{"d_" + kk: vv for kk, vv in d.items()}))
        g.close()
    for key in out[0]:
        assert np.array_equal(out[0][key], out[1][key], equal_nan=True), key
    for o in out:
        assert not np.array_equal(o["K2"], o["K"])
    f = out[2]
    for key in ("xs", "us", "k", "K", "cost"):
        scale = max(1.0, np.abs(out[0][key]).max())
        assert np.abs(out[0][key] - f[key]).max() <= 1e-9 * scale, (key, np.abs(out[0][key] - f[key]).max())
    for key in ("d_fx", "d_fu", "d_cxx", "d_cxu", "d_cuu"):  # the constant blocks: the same on every route
        assert np.array_equal(out[0][key], f[key]), key
    assert np.allclose(f["d_cx"], np.einsum("btij,btj->bti", f["d_cxx"], f["xs"]), rtol=1e-12, atol=1e-13)
    assert np.allclose(f["d_cu"][:, :T], np.einsum("btij,btj->bti", f["d_cuu"][:, :T], f["us"]), rtol=1e-12, atol=1e-13)
