"""A user's device twin that is neither small nor linear-quadratic (examples/user_model_pendulum_chain.hpp: n = 16, m = 4, trigonometric
dynamics, non-quadratic cost) on the generic kernels -- the path any twin of n > 8 takes, with finite differences point by point through the
model's own functions (src/derivatives.cpp:15-144) -- against its oracle twin (oracle/orc_models.inc: chain_*): rollout, every block of the
derivative records, and whole iterations walked by both drives."""
import numpy as np
import pytest

from tests.util import TOL, mat, relerr

pytestmark = pytest.mark.gpu
DT = 0.02
NL = 8
PARAMS = np.array([9.81, 0.1, 2.0, 10.0, 1.0, 0.1, 50.0, 0.0])


@pytest.fixture(scope="module")
def chain_lib():
    from ilqr_amd import _build
    import os
    if not os.path.exists(_build.USER_CHAIN_LIB) and not os.path.exists(_build.HIPCC):
        pytest.skip("the pendulum-chain build is missing and there is no hipcc to make it")
    return _build.build_user(_build.USER_CHAIN_HEADER, _build.USER_CHAIN_LIB)


def chain_x0(B, seed=3):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(-1, 1, (B, NL)), rng.uniform(-1, 1, (B, NL)) * 0.5], axis=1)


def make(oracle, chain_lib, B, T, lim, **kw):
    from ilqr_amd import BatchILQR, capi
    g = BatchILQR("user", B, T, DT, u_min=-lim, u_max=lim, lib=chain_lib, nx=2 * NL, nu=NL // 2, user_params=PARAMS, **kw)
    names = {s: g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index(s)) for s in ("derivatives", "backward", "rollout")}
    assert names == {"derivatives": b"k_derivatives_g", "backward": b"k_backward_w3", "rollout": b"k_rollout_g"}, names
    return oracle.Model("chain", chain=(NL, PARAMS), u_lim=lim), g


def test_rollout_and_records_match_the_oracle_twin(oracle, chain_lib):
    B, T, lim = 23, 50, 2.0
    om, g = make(oracle, chain_lib, B, T, lim)
    x0 = chain_x0(B)
    u0 = np.random.default_rng(1).normal(size=(B, T, NL // 2)) * 0.5
    cost = g.init_traj(x0, u0)
    xs, us = g.trajectory()
    xs_o, us_o, c_o = oracle.batch_rollout(om, x0, u0, DT)
    assert relerr(xs, xs_o) < 1e-12 and np.array_equal(us, u0) and np.max(np.abs(cost - c_o) / np.abs(c_o)) < 1e-12
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=c_o)
    g.compute_derivatives()
    d = g.derivatives()
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    for key in ("fx", "fu", "cx", "cu"):  # first differences: 1e-16 f / eps of rounding noise
        ref = do[key] if key in ("cx", "cu") else mat(do[key])
        assert np.abs(d[key] - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), key
    for key in ("cxx", "cxu", "cuu"):  # second differences of a cost of O(100): 1e-16 x 100 / 4 eps^2 = 2.5e-9 per entry
        ref = mat(do[key])
        assert np.abs(d[key] - ref).max() <= TOL * max(1.0, np.abs(ref).max()), (key, np.abs(d[key] - ref).max())
    assert np.abs(d["cxx"]).max() > 1.0 and np.abs(d["fx"] - np.eye(2 * NL)[None, None]).max() > 0.01  # (not a trivial model)
    g.close()


@pytest.mark.parametrize("lim,iters", [(2.0, 6), (0.5, 5)])
def test_iterations_walked_against_the_oracle_twin(oracle, chain_lib, lim, iters):
    """Whole iterations (finite differences -> k_backward_w3 with the masked 4 x 4 box-QP -> 11 rollouts -> accept), both drives; limits
    +-0.5 keep most controls clamped."""
    from tests.parity import walk_iterations
    B, T = 21, 80
    om, g = make(oracle, chain_lib, B, T, lim)
    x0 = chain_x0(B, seed=5)
    u0 = np.zeros((B, T, NL // 2))
    for drive in ("oracle", "gpu"):
        r = walk_iterations(oracle, om, g, x0, u0, DT, iters, drive=drive)
        print("chain walk", drive, lim, {kk: v for kk, v in r.items() if kk not in ("per_iter", "tied")})
        assert r["checked"] >= B * min(iters, 3), r["checked"]
        ties = r["ties_backward"] + r["ties_search"] + r["ties_stop"]
        assert ties <= max(2, r["checked"] // 10), r
        assert r["cond_over10"] <= max(1, r["checked"] // 20) and r["unresolved"] == 0, r
    g.generate_trajectory()
    assert g.count_running() == 0 and np.all(np.isfinite(g.cost()))
    g.close()


def test_full_solve_in_distribution(oracle, chain_lib):
    """Free-running solves on both sides.  This model ends its solves in a plateau (lambda growing to lambdaMax through no-step iterations, or a
    cost change just under tolFun): WHICH iteration trips the exit is a tie of the kind the walks above prove one by one, so the exits agree
    for most trajectories, not all (recorded: 75 %) -- and the cost a solve ends at does not depend on it."""
    B, T, lim = 32, 120, 2.0
    om, g = make(oracle, chain_lib, B, T, lim)
    x0 = chain_x0(B, seed=8)
    u0 = np.zeros((B, T, NL // 2))
    g.init_traj(x0, u0)
    g.generate_trajectory()
    st, it, al = g.status()
    ro = oracle.batch_solve(om, x0, u0, DT)
    same = (st == ro["status"]) & (it == ro["iters"])
    rel = np.abs(g.cost() - ro["cost"]) / np.abs(ro["cost"])
    print("chain full solves: same exit %.2f, cost rel err median %.2e max %.2e" % (same.mean(), np.median(rel), rel.max()))
    assert same.mean() >= 0.6, (same.mean(), st, ro["status"], it, ro["iters"])
    assert np.abs(it - ro["iters"]).max() <= 2
    assert np.median(rel) < 1e-6 and (rel < 1e-4).mean() >= 0.9, rel
    g.close()
