"""k_backward_h (ILQR_FLAG_BACKWARD_LANE_GROUP: sixteen lanes per trajectory, backward_hex.hpp; an opt-in experiment for
the acrobot) against the oracle's backward_pass with the criteria of the four-lane kernel, and against that kernel:
sums run in a different order, so the two agree to rounding, not bit for bit."""
import numpy as np
import pytest

from tests.parity import check_backward
from tests.test_gpu_parity import DT, make, u_init
from tests.util import mat

pytestmark = pytest.mark.gpu


def _pass(oracle, flags, B, T, lim, lam, dtype="f64", late=0):
    from ilqr_amd import capi
    om, g, x0 = make(oracle, "acrobot", B, T, lim, flags=capi.FLAG_UNFUSED | flags, dtype=dtype)
    if late:
        ro0 = oracle.batch_solve(om, x0, np.zeros((B, T, 1)), DT, max_iters=late, fixed_work=True)
        xs_o, us_o, cost_o, lam, k_prev = ro0["xs"], ro0["us"], ro0["cost"], ro0["lam"], ro0["k"]
    else:
        xs_o, us_o, cost_o = oracle.batch_rollout(om, x0, u_init(B, T, 1, scale=1.0), DT)
        k_prev = u_init(B, T, 1, seed=11, scale=0.2)
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    ro = oracle.batch_backward(om, us_o, do, k_prev=k_prev, lam=lam)
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=cost_o)
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    g.set_gains(k=k_prev, K=np.zeros((B, T, 1, 4)))
    g.set_lambda(lam, 1.0)
    div = g.backward_pass()
    k, K = g.gains()
    out = dict(k=k, K=K, dV=g.dV(), div=np.asarray(div), gnorm=g.gnorm())
    g.close()
    return om, us_o, do, k_prev, lam, ro, out


@pytest.mark.parametrize("lam", [1.0, 1e-3, 0.0])
@pytest.mark.parametrize("B,T,lim", [(48, 60, 5.0), (37, 131, 1.5), (3, 1, 0.5)])
def test_hex_backward_teacher_forced(oracle, B, T, lim, lam):
    from ilqr_amd import capi
    om, us_o, do, k_prev, lam_, ro, h = _pass(oracle, capi.FLAG_BACKWARD_LANE_GROUP, B, T, lim, lam)
    check_backward(oracle, om, us_o, do, k_prev, lam_, h["k"], h["K"], h["dV"], h["div"], ro, max_ties=max(1, B // 16))
    _, _, _, _, _, _, q = _pass(oracle, 0, B, T, lim, lam)
    # against the four-lane kernel: same divergence knots; gains to rounding for all but the (rare) trajectories
    # where a last-bit difference meets a tie or a badly conditioned pass (those are covered by the oracle check)
    assert np.array_equal(h["div"], q["div"])
    den = np.maximum(np.abs(q["K"]).reshape(B, -1).max(axis=1), 1e-300)
    err = np.abs(h["K"] - q["K"]).reshape(B, -1).max(axis=1) / den
    assert np.median(err) < 1e-10 and (err < 1e-6).mean() >= 0.9, (np.median(err), (err < 1e-6).mean())
    assert np.allclose(h["gnorm"][err < 1e-6], q["gnorm"][err < 1e-6], rtol=1e-6)


def test_hex_backward_late_in_a_solve(oracle):
    """lambda = 0, Quu up to 1e14, QPs that leave the two-iteration fast path (qp1_continue) -- at the bench's horizon."""
    from ilqr_amd import capi
    B = 32
    om, us_o, do, k_prev, lam, ro, h = _pass(oracle, capi.FLAG_BACKWARD_LANE_GROUP, B, 499, 1.5, None, late=30)
    # (sums in tree / rotated order are a little noisier than the reference's left-to-right ones where Quu is 1e14: a few
    #  more trajectories than for the four-lane kernel sit between 10x and 100x the oracle's own distance to fp80)
    r = check_backward(oracle, om, us_o, do, k_prev, lam, h["k"], h["K"], h["dV"], h["div"], ro, max_ties=B // 8, max_over10=B // 8)
    print("hex, late in a solve:", {kk: v for kk, v in r.items() if kk != "good"})


def test_hex_backward_fp32(oracle):
    from ilqr_amd import capi
    B, T = 32, 60
    _, _, _, _, _, _, h = _pass(oracle, capi.FLAG_BACKWARD_LANE_GROUP, B, T, 5.0, 1.0, dtype="f32")
    _, _, _, _, _, _, q = _pass(oracle, 0, B, T, 5.0, 1.0, dtype="f32")
    assert np.array_equal(h["div"], q["div"])
    den = np.maximum(np.abs(q["K"]).reshape(B, -1).max(axis=1), 1e-300)
    err = np.abs(h["K"] - q["K"]).reshape(B, -1).max(axis=1) / den
    assert np.median(err) < 1e-3, np.median(err)  # (float rounding through a 60-step recursion)


def test_hex_fused_sweep_matches_the_default_route(oracle):
    """ILQR_AMD_HEX=1 (experiment): the fused sweep + backward pass with FOUR 16-lane backward wavefronts per tile and the
    tile-wide pass votes of RingGateH, per-stage launches.  One iteration from the same start agrees with the default
    route to rounding (gains per trajectory, lambda schedule, accepted step); a solve to termination ends with the
    same statuses for all but a few trajectories (last-bit differences meet ties over many iterations) and never
    costs more than marginally."""
    import os
    from ilqr_amd import BatchILQR, capi
    B, T = 100, 60  # (7 tiles, the last one partly filled)
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]) * 0.5
    outs = {}
    for label in ("quad", "hex"):
        if label == "hex":
            os.environ["ILQR_AMD_HEX"] = "1"
        try:
            g = BatchILQR("acrobot", B, T, DT, flags=capi.FLAG_STAGED, u_min=-1.5, u_max=1.5, params=dict(max_iter=40))
            g.init_traj(x0, np.zeros((B, T, 1)))
            g.iterate(1)
            k, K = g.gains()
            lam, _ = g.lambdas()
            st, it, al = g.status()
            first = dict(K=K.copy(), lam=lam.copy(), al=np.asarray(al).copy(), cost=g.cost().copy())
            g.generate_trajectory()
            st, it, al = g.status()
            outs[label] = dict(first=first, cost=g.cost(), st=np.asarray(st), it=np.asarray(it))
            g.close()
        finally:
            os.environ.pop("ILQR_AMD_HEX", None)
    q, h = outs["quad"], outs["hex"]
    den = np.maximum(np.abs(q["first"]["K"]).reshape(B, -1).max(axis=1), 1e-300)
    err = np.abs(h["first"]["K"] - q["first"]["K"]).reshape(B, -1).max(axis=1) / den
    assert np.median(err) < 1e-10 and (err < 1e-6).mean() >= 0.95, (np.median(err), (err < 1e-6).mean())
    same = err < 1e-6
    assert np.array_equal(h["first"]["al"][same], q["first"]["al"][same]) and np.array_equal(h["first"]["lam"][same], q["first"]["lam"][same])
    assert np.allclose(h["first"]["cost"][same], q["first"]["cost"][same], rtol=1e-9)
    assert np.all(np.isfinite(h["cost"])) and np.all(h["st"] != 0)
    assert (h["st"] == q["st"]).mean() >= 0.9
    assert np.median(h["cost"] / q["cost"]) < 1.001
