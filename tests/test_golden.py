"""Committed golden vectors (tests/golden/, made by scripts/make_golden.py in the build container).

CPU: the oracle reproduces (a) the REAL reference's outputs for finite differences and box-QP,
(b) its own recorded stage outputs (regression).  GPU: the HIP path reproduces both through the
C ABI -- these run on the GPU box, where /root/reference does not exist."""
import os

import numpy as np
import pytest

from tests.util import TOL, mat, relerr, relerr_abs

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = 0.02


def _model(oracle, name, goal, lim=None):
    return oracle.Model(name, goal=goal if name != "acrobot" else None, u_lim=lim)


@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_oracle_matches_reference_fd_vectors(oracle, name):
    d = np.load(os.path.join(G, "ref_pieces.npz"))
    om = _model(oracle, name, d[name + "_goal"])
    xs, us = d[name + "_x"], d[name + "_u"]
    # one transition per knot: xs[b] = (x, x) so that knot 0 is a running knot and knot 1 a final one
    dv = oracle.batch_derivatives(om, np.stack([xs, xs], axis=1), us[:, None, :], DT)
    for k in ("fx", "fu", "cx", "cu", "cxx", "cuu"):
        assert np.array_equal(dv[k][:, 0], d["%s_%s" % (name, k)]), k  # bit-exact vs the real reference
    for k in ("cx", "cxx", "cuu"):
        assert np.array_equal(dv[k][:, 1], d["%s_final_%s" % (name, k)]), k


@pytest.mark.parametrize("m", [1, 2])
def test_oracle_matches_reference_boxqp_vectors(oracle, m):
    d = np.load(os.path.join(G, "ref_pieces.npz"))
    n_tie = 0
    for i in range(len(d["qp%d_result" % m])):
        r = oracle.boxqp(d["qp%d_Q" % m][i], d["qp%d_c" % m][i], d["qp%d_x0" % m][i], d["qp%d_lo" % m][i], d["qp%d_hi" % m][i])
        if r["result"] != d["qp%d_result" % m][i]:
            assert {int(r["result"]), int(d["qp%d_result" % m][i])} == {2, 4}  # converged-point tie
            n_tie += 1
        assert np.array_equal(r["v_free"], d["qp%d_v_free" % m][i])
        assert np.allclose(r["x_opt"], d["qp%d_x_opt" % m][i], rtol=1e-10, atol=1e-13)
    assert n_tie <= 3


@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_oracle_stage_regression(oracle, name):
    d = np.load(os.path.join(G, "stages_%s.npz" % name))
    om = _model(oracle, name, d["goal"], float(d["lim"]))
    xs, us, cost = oracle.batch_rollout(om, d["x0"], d["u0"], DT)
    assert np.array_equal(xs, d["xs"]) and np.array_equal(cost, d["cost"])
    dv = oracle.batch_derivatives(om, xs, us, DT)
    for kk in oracle.DERIV_NAMES:
        assert np.array_equal(dv[kk], d["d_" + kk])
    bw = oracle.batch_backward(om, us, dv, k_prev=d["k_prev"], lam=1.0)
    assert np.array_equal(bw["k"], d["k"]) and np.array_equal(bw["K"], d["K"]) and np.array_equal(bw["dV"], d["dV"])


def _gpu(name, d, B, T):
    from ilqr_amd import BatchILQR
    lim = float(d["lim"]) if "lim" in d else None
    kw = dict(u_min=-lim, u_max=lim) if lim else {}
    if name == "acrobot":
        return BatchILQR("acrobot", B, T, DT, **kw)
    return BatchILQR("integrator", B, T, DT, goal=d["goal"], **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_gpu_matches_reference_fd_vectors(name):
    """HIP finite differences against the REAL reference's arrays for the same knots."""
    d = np.load(os.path.join(G, "ref_pieces.npz"))
    xs, us = d[name + "_x"], d[name + "_u"]
    B = len(xs)
    g = _gpu(name, {"goal": d[name + "_goal"]}, B, 1)
    g.set_trajectory(x0=xs, xs=np.stack([xs, xs], axis=1), us=us[:, None, :], cost=np.zeros(B))
    g.compute_derivatives()
    dv = g.derivatives()
    for k in ("fx", "fu", "cxx", "cuu"):
        assert relerr_abs(dv[k][:, 0], mat(d["%s_%s" % (name, k)]), 1e-2) < TOL, k
    for k in ("cx", "cu"):
        assert relerr_abs(dv[k][:, 0], d["%s_%s" % (name, k)], 1e-2) < TOL, k
    assert relerr_abs(dv["cxx"][:, 1], mat(d[name + "_final_cxx"]), 1e-2) < TOL
    assert relerr_abs(dv["cx"][:, 1], d[name + "_final_cx"], 1e-2) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_gpu_matches_golden_stages(oracle, name):
    d = np.load(os.path.join(G, "stages_%s.npz" % name))
    B, T = d["u0"].shape[:2]
    g = _gpu(name, d, B, T)
    cost = g.init_traj(d["x0"], d["u0"])
    xs, us = g.trajectory()
    assert relerr(xs, d["xs"]) < TOL and np.allclose(cost, d["cost"], rtol=TOL)
    g.set_trajectory(x0=d["x0"], xs=d["xs"], us=d["us"], cost=d["cost"])
    g.set_derivatives(**{k: (d["d_" + k] if k in ("cx", "cu") else mat(d["d_" + k])) for k in ("fx", "fu", "cx", "cu", "cxx", "cxu", "cuu")})
    g.set_gains(k=d["k_prev"], K=np.zeros((B, T, d["k"].shape[2], 4)))
    g.set_lambda(1.0, 1.0)
    div = g.backward_pass()
    k, K = g.gains()
    assert np.array_equal(div, d["diverge"])
    assert relerr(k, d["k"]) < TOL and relerr(K, mat(d["K"])) < TOL and relerr(g.dV(), d["dV"]) < TOL
    cc = g.rollout_candidates()
    fin = np.isfinite(d["cand_cost"]) & (np.abs(d["cand_cost"]) < 1e12)
    assert np.allclose(cc[fin], d["cand_cost"][fin], rtol=TOL)
    # three full iterations from zero controls
    g.init_traj(d["x0"], np.zeros_like(d["u0"]))
    g.iterate(3)
    st, it, al = g.status()
    cost3 = g.cost()
    # every step of that run checked against the oracle; a trajectory that left the golden solution did so at
    # a proven tie (tests/parity.py)
    from tests.parity import assert_free_run, walk_both
    om = _model(oracle, name, d["goal"], float(d["lim"]))
    tied, r_o, r_g = walk_both(oracle, om, g, d["x0"], np.zeros_like(d["u0"]), DT, 3)
    ok = assert_free_run(cost3, d["sol_cost"], tied, name)
    assert np.array_equal(it[ok], d["sol_iters"][ok])
