"""Known answers of the reference's own unit tests, checked against the CPU oracle.

Each test names the reference test it restates (file:line under /root/reference/test).
"""
import numpy as np
import pytest


def test_clamp(oracle):  # test_boxqp.cpp:16-24
    out = oracle.clamp_to_limits([20.0, -50.0, 1.0], [-10.0] * 3, [5.0] * 3)
    assert np.allclose(out, [5.0, -10.0, 1.0], rtol=0, atol=1e-6)


def test_quad_cost(oracle):  # test_boxqp.cpp:38-48
    v = oracle.quad_cost([[0.25, 0.0], [0.0, 0.6]], [-15.0, 1.0], [0.35, 0.7])
    assert abs(v - (-4.3876875)) < 1e-6
    assert v == pytest.approx(-4.3876875000000002, rel=1e-15)


H2 = [[2.0, 0.0], [0.0, 2.0]]


def test_line_search_easy(oracle):  # test_boxqp.cpp:50-70
    r = oracle.line_search([2.0, 2.0], [-1.0, -1.0], H2, [0.0, 0.0], [-10.0] * 2, [10.0] * 2)
    assert not r["failed"]
    assert np.allclose(r["x_opt"], [1.0, 1.0], atol=1e-6)
    assert abs(r["v_opt"] - 2) < 1e-6


def test_line_search_wrong_direction(oracle):  # test_boxqp.cpp:72-85
    r = oracle.line_search([2.0, 2.0], [1.0, 1.0], H2, [0.0, 0.0], [-10.0] * 2, [10.0] * 2)
    assert r["failed"]


def test_line_search_hits_limits(oracle):  # test_boxqp.cpp:87-102
    r = oracle.line_search([2.0, 2.0], [-1.0, -1.0], H2, [0.0, 0.0], [1.5, 1.5], [10.0] * 2)
    assert not r["failed"]
    assert np.allclose(r["x_opt"], [1.5, 1.5], atol=1e-6)
    assert abs(r["v_opt"] - 4.5) < 1e-6


def test_boxqp1(oracle):  # test_boxqp.cpp:112-128 (+ oracle output recorded in SURVEY.md 8c)
    r = oracle.boxqp(H2, [0.0, 0.0], [2.0, 2.0], [-10.0] * 2, [10.0] * 2)
    assert np.allclose(r["x_opt"], [0.0, 0.0], atol=1e-6)
    assert r["result"] == 5
    assert np.allclose(r["R_free"], np.sqrt(2) * np.eye(2), atol=1e-12)


def test_boxqp2(oracle):  # test_boxqp.cpp:130-154
    r = oracle.boxqp(H2, [0.0, 0.0], [2.0, 2.0], [1.5, 1.5], [10.0] * 2)
    assert r["result"] == 6
    assert np.allclose(r["x_opt"], [1.5, 1.5], atol=1e-6)
    assert list(r["v_free"]) == [0, 0]


def test_boxqp3(oracle):  # test_boxqp.cpp:156-180
    r = oracle.boxqp([[3.001, 0], [0, 3.001]], [0.201, 0.201], [0.0, 0.0], [-0.6] * 2, [0.4] * 2)
    assert r["result"] == 5
    assert np.allclose(r["x_opt"], [-0.0669777, -0.0669777], atol=1e-6)
    assert r["x_opt"][0] == pytest.approx(-0.066977674108630481, rel=1e-13)
    assert list(r["v_free"]) == [1, 1]
    assert np.allclose(r["R_free"], 1.7323394586512193 * np.eye(2), atol=1e-12)


def test_boxqp4_3d_one_clamp(oracle):  # test_boxqp.cpp:184-202 (prints only; SURVEY.md 8c values)
    H = np.eye(3)
    H[1, 1] = 5.0
    r = oracle.boxqp(H, np.zeros(3), [0.5, 0.5, 1.0], [0.2, -1.0, -1.0], [1.0] * 3)
    assert r["result"] == 5
    assert np.allclose(r["x_opt"], [0.2, 0.0, 0.0], atol=1e-12)
    assert list(r["v_free"]) == [0, 1, 1]
    assert np.allclose(r["R_free"], np.diag([2.2360679774997898, 1.0]), atol=1e-12)


def test_boxqp_nonpd_1d_partial_factor(oracle):  # LLT failure is never checked
    r = oracle.boxqp([[-2.0]], [1.0], [0.0], [-5.0], [5.0])
    assert r["result"] == 2
    assert r["x_opt"][0] == pytest.approx(-0.25)
    assert r["R_free"][0, 0] == -2.0


def test_double_integrator_model(oracle):  # test_dynamicsmodels.cpp:32-60
    m = oracle.Model("integrator", goal=[1.0, 1.0, 0.0, 0.0])
    x, u = [0.0, 0.0, 0.5, 0.1], [1.0, -1.0]
    dx = m.dynamics(x, u)
    assert np.allclose(dx, [0.5, 0.1, 1.0, -1.0], atol=1e-6)
    assert np.allclose(m.integrate(x, u, 0.05), np.array(x) + 0.05 * dx, atol=1e-6)
    assert abs(m.cost([0.1, 0.1, 0.5, 0.1], [0.1, -1.0]) - 2.682) < 1e-3


def test_forward_pass_first_step(oracle):  # test_ilqr_forward_pass.cpp:52-81 (T=9, dt=.05)
    m = oracle.Model("integrator", goal=[1.0, 1.0, 0.0, 0.0])
    s = oracle.Solver(m, 9, 0.05)
    s.init_traj(np.zeros(4), np.full((9, 2), 0.1))
    assert np.allclose(s.xs[1], [0.0, 0.0, 0.005, 0.005], atol=1e-3)
    # the commented-out expectations of the same test (:59-60) hold too
    assert np.allclose(s.xs[9], [0.0113, 0.0113, 0.05, 0.05], atol=1e-2)


def test_derivatives_double_integrator(oracle):  # test_ilqr_derivatives.cpp:38-94 (commented out there)
    m = oracle.Model("integrator", goal=[1.0, 1.0, 0.0, 0.0])
    dt = 0.05
    s = oracle.Solver(m, 9, dt)
    s.init_traj(np.zeros(4), np.full((9, 2), 0.1))
    s.compute_derivatives()
    fx_exp = np.eye(4)
    fx_exp[0, 2] = fx_exp[1, 3] = dt
    fu_exp = np.zeros((4, 2))
    fu_exp[2, 0] = fu_exp[3, 1] = dt
    assert np.allclose(s.mat("fx")[0], fx_exp, atol=1e-2)
    assert np.allclose(s.mat("fu")[0], fu_exp, atol=1e-2)
    assert np.allclose(s.vecs("cx")[0], [-2, -2, 0, 0], atol=1e-2)
    assert np.allclose(s.vecs("cu")[0], [0.2, 0.2], atol=1e-2)
    assert np.allclose(s.mat("cxx")[0], np.diag([2, 2, 0.4, 0.4]), atol=1e-2)
    assert np.allclose(s.mat("cxu")[0], 0, atol=1e-2)
    assert np.allclose(s.mat("cuu")[0], 2 * np.eye(2), atol=1e-2)
    # fx[T], fu[T] are never written (derivatives.cpp:19), cu[T] = 0 (:50-51)
    assert np.all(s.mat("fx")[9] == 0) and np.all(s.mat("fu")[9] == 0) and np.all(s.vecs("cu")[9] == 0)
