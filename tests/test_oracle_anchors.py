"""Golden anchors recorded from the reference binary in SURVEY.md section 8(c).

class iLQR of the reference cannot be compiled in this image (include/ilqr.h:12 needs a gtest
header that is absent and may not be stubbed), so forward_pass / backward_pass / the outer loop
of the oracle are pinned by these observed outputs of `./run_iLQR acrobot|integrator`.
"""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def acrobot_first_pass(oracle):
    m = oracle.Model("acrobot")
    s = oracle.Solver(m, 499, 0.02)
    c0 = s.init_traj(np.zeros(4), np.zeros((499, 1)))
    s.compute_derivatives()
    div = s.backward_pass()
    return s, c0, div


def test_acrobot_initial_cost(acrobot_first_pass):
    _, c0, _ = acrobot_first_pass
    assert c0 == pytest.approx(3947.6089000000002, rel=1e-15)


def test_acrobot_first_backward_pass(acrobot_first_pass):
    s, _, div = acrobot_first_pass
    assert div == 0
    assert s.dV[0] == pytest.approx(-3452.217975208916, rel=1e-12)
    assert s.dV[1] == pytest.approx(65.646637761198917, rel=1e-12)
    assert s.k[0, 0] == pytest.approx(1.70299, rel=5e-6)
    assert np.allclose(s.K[0][0], [0.172117, 0.0522323, -0.331642, -0.286526], rtol=6e-6)
    assert s.k[249, 0] == pytest.approx(-1.70681, rel=5e-6)
    assert s.k[498, 0] == 0.0
    assert np.allclose(s.K[498][0], [-2.05749, 1.88238, 4.68554, -12.4948], rtol=6e-6)


def test_acrobot_first_derivatives(acrobot_first_pass):
    s, _, _ = acrobot_first_pass
    fx0 = [[1, 0, .02, 0], [0, 1, 0, .02], [-0.140143, 0.0280285, 1, 0], [0.112114, -0.140143, 0, 1]]
    assert np.allclose(s.mat("fx")[0], fx0, rtol=6e-6, atol=1e-12)
    assert np.allclose(s.mat("fu")[0].ravel(), [0, 0, -0.0171429, 0.0457143], rtol=6e-6, atol=1e-12)
    assert np.allclose(s.mat("cxx")[499], 800 * np.eye(4), rtol=1e-8, atol=1e-5)
    assert s.mat("cuu")[0][0, 0] == pytest.approx(0.02, rel=1e-9)
    assert np.all(s.mat("fx")[499] == 0)


def test_acrobot_canonical_solve(oracle):
    """`./run_iLQR acrobot`: 100 iterations (hits maxIter), final cost 5.39788253688, lambda 0,
    max|u| 2.22044; one backward pass per iteration, ~3.84 rollouts per iteration."""
    m = oracle.Model("acrobot")
    s = oracle.Solver(m, 499, 0.02)
    st, log = s.generate_trajectory(np.zeros(4), np.zeros((499, 1)), log=True)
    assert oracle.STATUS[st] == "max_iter" and s.iters == 100
    assert s.cost == pytest.approx(5.39788253688, rel=1e-9)
    assert s.lam == 0.0
    assert np.abs(s.us).max() == pytest.approx(2.22044, rel=1e-5)
    assert np.allclose(log[:3], [2.87e3, 2.66e3, 1.96e3], rtol=2e-3)  # the per-iteration cost table of SURVEY 8c: its first three rows ...
    assert log[-1] == pytest.approx(5.4, rel=1e-3) and np.all(np.diff(log) <= 0)  # ... its last, and what every row in between obeys
    assert s.s.contents.n_backward == 100
    assert s.s.contents.n_rollouts == 385  # 1 initial + 384 line-search rollouts (3.84/iter)


def test_integrator_first_backward_pass(oracle):
    m = oracle.Model("integrator", goal=[1, .5, 0, 0])
    s = oracle.Solver(m, 99, 0.02)
    c0 = s.init_traj([-1, 0, 0, -.2], np.zeros((99, 2)))
    assert c0 == pytest.approx(494.1509440000001, rel=1e-15)
    s.compute_derivatives()
    assert s.backward_pass() == 0
    assert s.dV[0] == pytest.approx(-168.85165147790619, rel=1e-11)
    assert s.dV[1] == pytest.approx(31.258604939734756, rel=1e-11)
    assert np.all(s.k[0] == [0.5, 0.5]) and np.all(s.K[0] == 0)  # both controls clamped


def test_integrator_canonical_solve(oracle):
    """`./run_iLQR integrator`: the loop ends at iteration index 14 (15 iterations) at cost 356
    after ten NO-STEP iterations that drive lambda to 10^9.2.

    What happens AT iteration 14 is decided by rounding noise: k ~ 1e-12 there, so the
    line search compares costs that differ by +-1 ulp of 356 (5.7e-14).  The reference binary
    happened to see dcost > 0 ("cost change < tolFun"); the oracle's rollouts round the other
    way on all 11 alphas and leave through "lambda > lambdaMax" in the same iteration.  Both
    are the same trajectory and cost; only iteration count, cost and the lambda history are
    asserted."""
    m = oracle.Model("integrator", goal=[1, .5, 0, 0])
    s = oracle.Solver(m, 99, 0.02)
    st, log = s.generate_trajectory([-1, 0, 0, -.2], np.zeros((99, 2)), log=True)
    assert oracle.STATUS[st] in ("converged_cost", "lambda_max")
    assert s.iters == 15
    assert round(s.cost) == 356
    assert s.cost == pytest.approx(356.168506469842, rel=1e-12)
    no_step = np.flatnonzero(np.diff(log) == 0)
    assert len(no_step) >= 10 and no_step[0] == 3  # iterations 4..13 are NO STEP


def test_integrator_lambda_history(oracle):
    """lambda after the tenth NO-STEP iteration is 10^9.2."""
    import ctypes as C
    m = oracle.Model("integrator", goal=[1, .5, 0, 0])
    s = oracle.Solver(m, 99, 0.02)
    s.init_traj([-1, 0, 0, -.2], np.zeros((99, 2)))
    flg = C.c_int(1)
    for _ in range(14):
        assert oracle.lib().orc_iterate_once(m.ref, s.s, C.byref(flg), 0) == 0
    assert np.log10(s.lam) == pytest.approx(9.2, abs=0.05)
