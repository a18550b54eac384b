"""The oracle against the REAL reference code (oracle/_ref/libref_ilqr.so = the reference's
src/boxqp.cpp, finite_diff.h and model headers compiled where they lie).  Container-only: the
shim needs /root/reference to build, so these tests skip on the GPU box."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def O(oracle):
    if not oracle.ref_available():
        pytest.skip("reference shim not built (no /root/reference here)")
    return oracle


def test_ref_kats(O):
    """test/test_finite_diff.cpp:13-32 on the reference's own operators."""
    assert abs(O.ref().ref_fd_scalar_negquad(-1.0) - 2) < 1e-6
    import ctypes as C
    out = np.zeros(2)
    x = np.array([1.0, 1.0])
    O.ref().ref_fd_grad_quadvec(x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.allclose(out, [2, 10], atol=1e-6)


@pytest.mark.parametrize("m", [1, 2])
def test_boxqp_exact_small(O, m):
    """m = 1, 2 (the shipped models' control dimensions): identical result code, free set and
    solution on every case, PD or not."""
    rng = np.random.default_rng(10 + m)
    for trial in range(4000):
        A = rng.normal(size=(m, m))
        Q = A @ A.T + (0.05 if trial % 4 else -0.3) * np.eye(m)
        c = rng.normal(size=m) * 2
        x0 = rng.normal(size=m)
        lo = -rng.uniform(0.05, 1.5, size=m)
        hi = rng.uniform(0.05, 1.5, size=m)
        a, b = O.boxqp(Q, c, x0, lo, hi), O.ref_boxqp(Q, c, x0, lo, hi)
        if m == 2 and a["result"] != b["result"]:
            # 2 <-> 4 at a converged point is a rounding-level tie (see test below)
            assert {a["result"], b["result"]} == {2, 4}
        else:
            assert a["result"] == b["result"]
        assert np.array_equal(a["v_free"], b["v_free"])
        assert np.allclose(a["x_opt"], b["x_opt"], rtol=1e-10, atol=1e-13, equal_nan=True)
        if a["R_free"].size:
            assert np.allclose(a["R_free"], b["R_free"], rtol=1e-12, atol=1e-14, equal_nan=True)


def test_boxqp_random_pd(O):
    """Positive-definite Q, m up to 6.  The oracle forms R^-1 R^-T by a triangular solve where
    the reference uses two PartialPivLU inverses: identical up to rounding,
    so >= 99.8 % of cases must agree on code, free set and x; the rest are knife-edge ties."""
    rng = np.random.default_rng(0)
    n_ok = n = 0
    for trial in range(6000):
        m = int(rng.integers(1, 7))
        A = rng.normal(size=(m, m))
        Q = A @ A.T + 0.05 * np.eye(m)
        c = rng.normal(size=m) * 2
        x0 = rng.normal(size=m)
        lo = -rng.uniform(0.05, 1.5, size=m)
        hi = rng.uniform(0.05, 1.5, size=m)
        a, b = O.boxqp(Q, c, x0, lo, hi), O.ref_boxqp(Q, c, x0, lo, hi)
        n += 1
        same = (a["result"] == b["result"] and np.array_equal(a["v_free"], b["v_free"]) and
                np.allclose(a["x_opt"], b["x_opt"], rtol=1e-9, atol=1e-12))
        n_ok += same
        assert a["result"] >= 1 and b["result"] >= 1
    assert n_ok >= 0.998 * n, (n_ok, n)


def test_line_search_and_helpers(O):
    rng = np.random.default_rng(1)
    for _ in range(3000):
        m = int(rng.integers(1, 6))
        A = rng.normal(size=(m, m))
        Q = A @ A.T + 0.1 * np.eye(m)
        c, x0, d = rng.normal(size=m), rng.normal(size=m), rng.normal(size=m)
        lo, hi = -rng.uniform(0.1, 2, size=m), rng.uniform(0.1, 2, size=m)
        a, b = O.line_search(x0, d, Q, c, lo, hi), O.ref_line_search(x0, d, Q, c, lo, hi)
        assert a["failed"] == b["failed"] and a["n_steps"] == b["n_steps"]
        assert np.allclose(a["x_opt"], b["x_opt"], rtol=1e-12, equal_nan=True)
        assert np.isclose(a["v_opt"], b["v_opt"], rtol=1e-12, equal_nan=True)
        assert np.isclose(O.quad_cost(Q, c, x0), O.ref_quad_cost(Q, c, x0), rtol=1e-13, atol=1e-13)
        assert np.array_equal(O.clamp_to_limits(x0, lo, hi), O.ref_clamp(x0, lo, hi))


@pytest.mark.parametrize("mid,name,goal", [(0, "acrobot", None), (1, "integrator", [1, .5, 0, 0])])
def test_models_and_fd_bit_exact(O, mid, name, goal):
    """dynamics / Euler step / cost / final_cost and every finite-difference array of a knot
    point equal the reference's BIT FOR BIT (same operation order, same libm, no FMA)."""
    rng = np.random.default_rng(2)
    m = O.Model(name, goal=goal)
    dt = 0.02
    for _ in range(200):
        x = rng.uniform(-1, 1, 4) * np.array([np.pi, np.pi, 3, 3])
        u = rng.uniform(-5, 5, m.nu)
        dx, x1, c, f = O.ref_model_eval(mid, goal, x, u, dt)
        assert np.array_equal(dx, m.dynamics(x, u))
        assert np.array_equal(x1, m.integrate(x, u, dt))
        assert c == m.cost(x, u) and f == m.final_cost(x)
        d = O.batch_derivatives(m, np.stack([x, x])[None], u[None, None], dt)
        for is_final in (0, 1):
            r = O.ref_fd_knot(mid, goal, x, u, dt, is_final)
            for kname in ("fx", "fu", "cx", "cu", "cxx", "cuu"):
                if is_final and kname in ("fx", "fu"):
                    continue
                assert np.array_equal(d[kname][0, is_final], r[kname]), (kname, is_final)
