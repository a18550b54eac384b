"""GPU parity tests: the HIP path (through the C ABI of include/ilqr_amd.h) against the CPU
oracle on identical seeded inputs, stage by stage with teacher forcing, plus
end-to-end solves on the cases that are numerically stable.  Tolerance: 1e-6 relative (fp64),
the figure BASELINE.json's north_star states; discrete outputs (diverge flags, accepted alpha,
status, iteration counts) must match exactly."""
import numpy as np
import pytest

from tests.parity import sampled_walk, assert_free_run, assert_walk, check_backward, gains_knot_err, publish, walk_both, walk_iterations
from tests.util import TOL, acrobot_x0, integrator_x0, mat, relerr, relerr_abs

pytestmark = pytest.mark.gpu

DT = 0.02
CASES = [
    # name, model kwargs (oracle), B, T, limit
    ("acrobot", 70, 60, 5.0),
    ("acrobot", 70, 60, 1.5),
    ("integrator", 33, 40, 0.5),
]


def make(oracle, name, B, T, lim, **kw):
    from ilqr_amd import BatchILQR
    if name == "acrobot":
        om = oracle.Model("acrobot", u_lim=lim)
        g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, **kw)
        x0 = acrobot_x0(B)
    else:
        goal = [1.0, 0.5, 0.0, 0.0]
        om = oracle.Model("integrator", goal=goal, u_lim=lim)
        g = BatchILQR("integrator", B, T, DT, u_min=-lim, u_max=lim, goal=goal, **kw)
        x0 = integrator_x0(B)
    return om, g, x0


def u_init(B, T, nu, seed=7, scale=0.3):
    return np.random.default_rng(seed).normal(size=(B, T, nu)) * scale


@pytest.mark.parametrize("name,B,T,lim", CASES)
def test_init_rollout(oracle, name, B, T, lim):
    om, g, x0 = make(oracle, name, B, T, lim)
    u0 = u_init(B, T, om.nu)
    cost = g.init_traj(x0, u0)
    xs, us = g.trajectory()
    xs_o, us_o, cost_o = oracle.batch_rollout(om, x0, u0, DT)
    assert np.array_equal(us, u0)  # open loop: controls are stored untouched (ilqr_core.cpp:323)
    assert relerr(xs, xs_o) < TOL
    assert np.max(np.abs(cost - cost_o) / np.abs(cost_o)) < TOL
    lam, dlam = g.lambdas()
    assert np.all(lam == 1) and np.all(dlam == 1)


@pytest.mark.parametrize("name,B,T,lim", CASES)
def test_derivatives_teacher_forced(oracle, name, B, T, lim):
    om, g, x0 = make(oracle, name, B, T, lim)
    u0 = u_init(B, T, om.nu)
    xs_o, us_o, cost_o = oracle.batch_rollout(om, x0, u0, DT)
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=cost_o)
    g.compute_derivatives()
    d = g.derivatives()
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    for k in ("fx", "fu", "cx", "cu", "cxx", "cuu"):
        ref = do[k] if k in ("cx", "cu") else mat(do[k])
        # floor: entries that are mathematically zero carry +-1e-9-class finite-difference noise
        assert relerr_abs(d[k], ref, 1e-2) < TOL, k
    # cxu is ~0 for both shipped models (pure FD noise, <= 1e-8 absolute) and cxu[T] is unused
    assert np.abs(d["cxu"] - mat(do["cxu"])).max() < 1e-6
    assert np.all(d["fx"][:, T] == 0) and np.all(d["fu"][:, T] == 0) and np.all(d["cu"][:, T] == 0)


def _per_traj_err(a, b):
    B = a.shape[0]
    return np.abs(a - b).reshape(B, -1).max(axis=1) / np.maximum(np.abs(b).reshape(B, -1).max(axis=1), 1e-300)


@pytest.mark.parametrize("lam", [1.0, 1e-3, 0.0])
@pytest.mark.parametrize("name,B,T,lim", CASES)
def test_backward_teacher_forced(oracle, name, B, T, lim, lam):
    om, g, x0 = make(oracle, name, B, T, lim)
    u0 = u_init(B, T, om.nu, scale=1.0)
    xs_o, us_o, cost_o = oracle.batch_rollout(om, x0, u0, DT)
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    k_prev = u_init(B, T, om.nu, seed=11, scale=0.2)
    ro = oracle.batch_backward(om, us_o, do, k_prev=k_prev, lam=lam)
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=cost_o)
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    g.set_gains(k=k_prev, K=np.zeros((B, T, om.nu, om.nx)))
    g.set_lambda(lam, 1.0)
    div = g.backward_pass()
    k, K = g.gains()
    dV = g.dV()
    Ko = mat(ro["K"])
    lo, hi = om.u_min[None, None, :] - us_o, om.u_max[None, None, :] - us_o
    conv = ro["diverge"] == 0
    # per-knot gains, dV, diverge flags; deviations must be fp64 conditioning or proven clamp ties (tests/parity.py)
    r = check_backward(oracle, om, us_o, do, k_prev, lam, k, K, dV, div, ro, max_ties=max(1, B // 16), max_over10=max(1, B // 50))
    good = r["good"]
    # k stays inside the box it was solved for
    ok = conv & good
    assert np.all(k[ok] >= lo[ok] - 1e-12) and np.all(k[ok] <= hi[ok] + 1e-12)
    if lim < 5:
        clamped = (np.abs(k - lo) < 1e-9) | (np.abs(k - hi) < 1e-9)
        assert clamped[ok].mean() > 0.02  # the active-limit path is exercised
        # a clamped control has a zero feedback row (src/ilqr_core.cpp:376-385)
        assert np.all(K[ok][clamped[ok]] == 0)


@pytest.mark.parametrize("iters", [25, 45])
def test_backward_teacher_forced_late_in_a_solve(oracle, iters):
    """The state a solve is in after 25 / 45 iterations (lambda = 0 for most trajectories, Quu up to
    1e14, box-QPs whose search direction is rounding noise, controls on the bounds): the oracle's
    trajectories, gains and lambdas go in, one backward pass comes out.  This is where the scalar
    QP leaves its two-iteration fast path (qp1_continue) and where the exact shortcuts for line
    searches that can only run out apply."""
    name, B, T, lim = "acrobot", 48, 499, 1.5  # the bench workload's horizon and limits
    om, g, x0 = make(oracle, name, B, T, lim)
    ro0 = oracle.batch_solve(om, x0, np.zeros((B, T, 1)), DT, max_iters=iters, fixed_work=True)
    xs_o, us_o, lam = ro0["xs"], ro0["us"], ro0["lam"]
    k_prev = ro0["k"]
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    ro = oracle.batch_backward(om, us_o, do, k_prev=k_prev, lam=lam)
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=ro0["cost"])
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    g.set_gains(k=k_prev, K=np.zeros((B, T, 1, 4)))
    g.set_lambda(lam, 1.0)
    div = g.backward_pass()
    k, K = g.gains()
    dV = g.dV()
    conv = ro["diverge"] == 0
    assert conv.mean() > 0.5 and (lam == 0).mean() > 0.2, (conv.mean(), (lam == 0).mean())
    r = check_backward(oracle, om, us_o, do, k_prev, lam, k, K, dV, div, ro, max_ties=B // 8, max_over10=max(1, B // 50))
    print("late in a solve:", {kk: v for kk, v in r.items() if kk != "good"})


@pytest.mark.parametrize("name,B,T,lim", CASES)
def test_rollout_candidates_teacher_forced(oracle, name, B, T, lim):
    om, g, x0 = make(oracle, name, B, T, lim)
    u0 = u_init(B, T, om.nu)
    xs_o, us_o, cost_o = oracle.batch_rollout(om, x0, u0, DT)
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    ro = oracle.batch_backward(om, us_o, do, lam=1.0)
    Kmat = mat(ro["K"])
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=cost_o)
    g.set_gains(k=ro["k"], K=Kmat)
    cost_c = g.rollout_candidates()
    from ilqr_amd import ALPHAS
    for a in (0, 3, 10):
        xs_a, us_a, c_a = oracle.batch_rollout(om, x0, us_o + ALPHAS[a] * ro["k"], DT, xs_nom=xs_o, K=Kmat)
        xg, ug = g.candidate(a)
        fin = np.isfinite(c_a) & (np.abs(xs_a).reshape(B, -1).max(axis=1) < 1e6)
        assert fin.sum() > B // 2
        assert relerr(xg[fin], xs_a[fin]) < TOL
        assert relerr(ug[fin], us_a[fin]) < TOL
        assert np.max(np.abs(cost_c[fin, a] - c_a[fin]) / np.abs(c_a[fin])) < TOL


@pytest.mark.parametrize("name,B,T,lim", CASES)
def test_one_iteration_from_same_state(oracle, name, B, T, lim):
    """init -> derivatives -> backward (+lambda retry) -> line search -> accept, all on device, against
    the oracle's iterate_once from the same state: same accepted alpha, lambda schedule, gains per knot
    and new cost for EVERY trajectory, or a proven tie (tests/parity.py)."""
    om, g, x0 = make(oracle, name, B, T, lim)
    u0 = np.zeros((B, T, om.nu))
    r = walk_iterations(oracle, om, g, x0, u0, DT, 1)
    assert r["checked"] == B and len(r["tied"]) <= max(1, B // 32), r
    # and from the device's own init_traj (the rollouts differ in the last bit)
    g.init_traj(x0, u0)
    g.iterate(1)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=1)
    st, it, al = g.status()
    assert np.array_equal(it, ro["iters"])
    same = np.isclose(g.cost(), ro["cost"], rtol=TOL)
    assert same.sum() >= B - max(1, B // 32) - len(r["tied"])
    lam, dlam = g.lambdas()
    assert np.allclose(lam[same], ro["lam"][same], rtol=1e-12)
    xs, us = g.trajectory()
    assert relerr(xs[same], ro["xs"][same]) < TOL


def test_acrobot_canonical_solve(oracle):
    """`./run_iLQR acrobot` (x0 = 0, T = 499): 100 iterations, cost 5.397882537."""
    from ilqr_amd import BatchILQR
    B, T = 3, 499
    g = BatchILQR("acrobot", B, T, DT)
    x0, u0 = np.zeros((B, 4)), np.zeros((B, T, 1))
    c0 = g.init_traj(x0, u0)
    assert np.allclose(c0, 3947.6089000000002, rtol=1e-12)
    g.generate_trajectory()
    st, it, al = g.status()
    assert np.all(st == 4) and np.all(it == 100)
    cost = g.cost()
    assert np.allclose(cost, 5.39788253688, rtol=TOL)
    assert np.all(cost == cost[0])  # identical problems -> bit-identical results
    om = oracle.Model("acrobot")
    ro = oracle.batch_solve(om, x0[:1], u0[:1], DT)
    k, K = g.gains()
    xs, us = g.trajectory()
    assert relerr(xs[:1], ro["xs"]) < 1e-5 and relerr(us[:1], ro["us"]) < 1e-4
    assert np.all(g.lambdas()[0] == 0.0)


def test_integrator_canonical_solve(oracle):
    """`./run_iLQR integrator`: stops at iteration index 14 (15 iterations), cost 356.1685."""
    from ilqr_amd import BatchILQR
    B, T = 2, 99
    g = BatchILQR("integrator", B, T, DT, goal=[1, .5, 0, 0])
    x0 = np.tile([-1.0, 0, 0, -.2], (B, 1))
    u0 = np.zeros((B, T, 2))
    c0 = g.init_traj(x0, u0)
    assert np.allclose(c0, 494.1509440000001, rtol=1e-13)
    g.generate_trajectory()
    st, it, al = g.status()
    cost = g.cost()
    assert np.allclose(cost, 356.168506469842, rtol=1e-9)
    # The reference ends at iteration index 14 = 15 iterations; WHY it ends there is decided by rounding
    # noise (test_oracle_anchors: "cost change < tolFun" in the reference binary, "lambda > lambdaMax" in the oracle,
    # same iteration, same trajectory).  The device must end in that iteration too, by one of the two.
    assert np.all(it == 15) and np.all(np.isin(st, (2, 3))), (it, st)
    # ... and every one of its 15 iterations is the oracle's iteration from the same state (device-driven walk)
    om = oracle.Model("integrator", goal=[1, .5, 0, 0], u_lim=0.5)
    r = walk_iterations(oracle, om, g, x0, u0, DT, 15, drive="gpu")
    assert r["checked"] == 15 * B and r["ties_backward"] == 0 and r["conditioned"] == 0, r


def test_small_scale_multi_iteration(oracle):
    """x0 scale 0.01, 5 iterations: stable regime of SURVEY.md 0.3 -> free-running end-to-end 1e-6 parity
    for every trajectory the per-iteration walk (tests/parity.py) finds no tie on."""
    from ilqr_amd import BatchILQR
    B, T = 32, 499
    om = oracle.Model("acrobot")
    x0 = acrobot_x0(B, scale=0.01)
    u0 = np.zeros((B, T, 1))
    g = BatchILQR("acrobot", B, T, DT)
    tied, r, r_g = walk_both(oracle, om, g, x0, u0, DT, 5)
    g.init_traj(x0, u0)
    g.iterate(5)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=5)
    cost = g.cost()
    ok = assert_free_run(cost, ro["cost"], tied)
    k, K = g.gains()
    xs, us = g.trajectory()
    assert gains_knot_err(k[ok], K[ok], ro["k"][ok], ro["K"][ok], us[ok]).max() < 1e-4  # five iterations of amplification
    st, it, al = g.status()
    assert np.array_equal(it[ok], ro["iters"][ok])


def test_warm_start(oracle):
    """generate_trajectory(x0) (ilqr_core.cpp:65-76): re-roll stored controls with stored gains."""
    from ilqr_amd import BatchILQR
    B, T = 8, 120
    g = BatchILQR("integrator", B, T, DT, goal=[1, .5, 0, 0])
    x0 = integrator_x0(B)
    g.generate_trajectory(x0, np.zeros((B, T, 2)))
    c1 = g.cost()
    xs1, us1 = g.trajectory()
    k1, K1 = g.gains()
    x0b = x0 + 0.01
    g.generate_trajectory(x0b)
    c2 = g.cost()
    assert np.all(np.isfinite(c2)) and np.all(c2 <= 1.5 * c1 + 1.0)
    # the warm-start rollout itself: u = us + K (x - xs)
    om = oracle.Model("integrator", goal=[1, .5, 0, 0])
    xs_w, us_w, c_w = oracle.batch_rollout(om, x0b, us1, DT, xs_nom=xs1, K=K1)
    assert np.all(c2 <= c_w * (1 + 1e-9))


@pytest.mark.parametrize("model", ["integrator", "acrobot", "lq"])
def test_warm_start_rollout_equals_the_oracles(oracle, model):
    """The warm start's own rollout (ilqr_core.cpp:65-76: forward_pass(x0, us) with the stored gains, u = us[t] + K[t] (x - xs[t]))
    against oracle.batch_rollout(xs_nom, K), knot for knot: a handle with max_iter = 0 stops right behind it."""
    from ilqr_amd import BatchILQR
    from tests.test_gpu_lq_end_to_end import dense_mats
    B, T = 24, 90
    rng = np.random.default_rng(77)
    if model == "integrator":
        kw, om, x0, m = dict(goal=[1, .5, 0, 0]), oracle.Model("integrator", goal=[1, .5, 0, 0]), integrator_x0(B), 2
    elif model == "acrobot":
        kw, om, x0, m = dict(u_min=-1.5, u_max=1.5), oracle.Model("acrobot", u_lim=1.5), acrobot_x0(B, scale=0.3, seed=4), 1
    else:
        mats = dense_mats(6, 3)
        kw, om, x0, m = dict(lq=mats, u_min=-0.4, u_max=0.4), oracle.Model("lq", lq=mats, u_lim=0.4), rng.uniform(-1, 1, (B, 6)), 3
    g = BatchILQR(model, B, T, DT, **kw)
    g.init_traj(x0, np.zeros((B, T, m)))
    g.iterate(4)  # a nominal trajectory with gains around it
    xs1, us1 = g.trajectory()
    k1, K1 = g.gains()
    c1 = g.cost()
    g.close()
    g0 = BatchILQR(model, B, T, DT, params=dict(max_iter=0), **kw)
    g0.set_trajectory(x0=x0, xs=xs1, us=us1, cost=c1)
    g0.set_gains(k=k1, K=K1)
    x0b = x0 + 0.02 * rng.standard_normal(x0.shape)
    g0.generate_trajectory(x0b)
    xs_w, us_w = g0.trajectory()
    xs_o, us_o, c_o = oracle.batch_rollout(om, x0b, us1, DT, xs_nom=xs1, K=K1)
    assert np.array_equal(xs_w[:, 0], x0b)
    assert relerr(xs_w, xs_o) < TOL and relerr(us_w, us_o) < TOL
    assert np.max(np.abs(g0.cost() - c_o) / np.abs(c_o)) < TOL
    assert np.array_equal(g0.status()[1], np.zeros(B, dtype=np.int32))  # no iteration ran
    g0.close()


def test_full_size_properties(oracle):
    """BASELINE.json configs[2] (acrobot T=499, B=4096, clamps active): size-independent properties of the whole batch,
    and 64 of its trajectories walked against the oracle iteration by iteration (tests/parity.py: Sampled)."""
    from ilqr_amd import BatchILQR
    B, T, lim = 4096, 499, 1.5
    g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim)
    from ilqr_amd import capi
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_hex"  # the route bench.py's headline takes
    x0 = acrobot_x0(B)
    # 25 iterations: the regime bench.py times (its iterations 6-25), every one of them against the oracle
    r = sampled_walk(oracle, oracle.Model("acrobot", u_lim=lim), g, x0, np.zeros((B, T, 1)), DT, 25)
    print("configs[2] sampled walk:", publish("configs[2] acrobot T=499 B=4096 +-1.5 fp64 (k_solve_hex, one tile per CU)", r, B=B, T=T, u_lim=lim, n_sample=len(r["sel"])))
    assert_walk(r, 25, min_plain=0.70)  # (recorded: 0.77)
    x0[1] = x0[0]
    x0[B - 1] = x0[0]  # duplicates across tiles / waves
    c0 = g.init_traj(x0, np.zeros((B, T, 1)))
    g.iterate(3)
    cost = g.cost()
    st, it, al = g.status()
    xs, us = g.trajectory()
    k, K = g.gains()
    assert np.all(np.isfinite(cost)) and np.all(cost <= c0 * (1 + 1e-12))  # monotone (line search)
    assert cost[0] == cost[1] == cost[B - 1]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(K[0], K[B - 1])
    assert np.array_equal(xs[:, 0], x0)
    dV = g.dV()
    assert (dV[:, 0] <= 0).mean() > 0.9  # k'Qu < 0 unless Quu went indefinite (reference warns, :207)
    accepted = al >= 0
    assert accepted.mean() > 0.5
    # rows of K are zeroed for clamped controls: with m = 1 that is the whole gain at that step
    assert (np.abs(K).reshape(B, T, -1).max(axis=2) == 0).mean() > 0.05


def test_config1_full_size(oracle):
    """BASELINE.json configs[1] at its own size: acrobot T=499, B=1024 random x0, limits +-5, fp64 -- 64 trajectories of
    the free-running batch walked against the oracle for three iterations, plus the batch-wide properties."""
    from ilqr_amd import BatchILQR
    B, T, lim = 1024, 499, 5.0
    g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim)
    x0 = acrobot_x0(B)
    u0 = np.zeros((B, T, 1))
    r = sampled_walk(oracle, oracle.Model("acrobot", u_lim=lim), g, x0, u0, DT, 25)
    print("configs[1] sampled walk:", publish("configs[1] acrobot T=499 B=1024 +-5 fp64", r, B=B, T=T, u_lim=lim, n_sample=len(r["sel"])))
    assert_walk(r, 25, min_plain=0.70)  # (recorded: 0.78)
    c0 = g.init_traj(x0, u0)
    g.iterate(3)
    cost = g.cost()
    assert np.all(np.isfinite(cost)) and np.all(cost <= c0 * (1 + 1e-12))
    assert (g.status()[2] >= 0).mean() > 0.5


def test_saturated_batch_wide_route_against_the_oracle(oracle):
    """The route a saturating batch takes (B = 16384: one 64-trajectory wide tile per CU, thread-per-trajectory backward
    chain, kernels_wide.hpp) at the headline horizon and limits: 64 trajectories of the free-running batch walked against the
    oracle for two iterations, and the first 4096 trajectories of the big batch equal, bit for bit, a 4096-trajectory batch
    of the same problems (which takes the one-tile-per-CU route)."""
    from ilqr_amd import BatchILQR, capi
    B, T, lim = 16384, 499, 1.5
    g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_wide"
    x0 = acrobot_x0(B)
    u0 = np.zeros((B, T, 1))
    r = sampled_walk(oracle, oracle.Model("acrobot", u_lim=lim), g, x0, u0, DT, 2)
    assert_walk(r, 2, max_on_records=0.65)  # (two iterations: the first, where 60 % need the stage-by-stage argument, is half of the walk)
    cost_big = g.cost()  # (two free-running iterations, left by the walk)
    st_big, it_big, al_big = g.status()
    g.close()
    g = BatchILQR("acrobot", 4096, T, DT, u_min=-lim, u_max=lim)
    g.init_traj(x0[:4096], u0[:4096])
    g.iterate(1)
    g.iterate(1)
    assert np.array_equal(g.cost(), cost_big[:4096]) and np.array_equal(g.status()[2], al_big[:4096])
    g.close()


def test_saturated_batch_wide_route_25_iterations(oracle):
    """The wide route over the iterations the saturated bench line times: 25 iterations of B = 16384, 64 trajectories
    against the oracle in every one; the bins go to the tracked statistics file."""
    from ilqr_amd import BatchILQR, capi
    B, T, lim = 16384, 499, 1.5
    g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_wide"
    x0 = acrobot_x0(B)
    r = sampled_walk(oracle, oracle.Model("acrobot", u_lim=lim), g, x0, np.zeros((B, T, 1)), DT, 25)
    print("wide route sampled walk:", publish("saturated acrobot T=499 B=16384 +-1.5 fp64 (k_solve_wide)", r, B=B, T=T, u_lim=lim, n_sample=len(r["sel"])))
    assert_walk(r, 25, min_plain=0.66)  # (recorded: 0.74)
    g.close()


@pytest.mark.parametrize("dtype,lim", [("f64", 1.5), ("f32", 5.0)])
def test_bench_saturated_batch_two_wide_tiles_per_cu_10_iterations(oracle, dtype, lim):
    """The configurations the bench's `saturated` line (fp64, +-1.5) and its configs[3]-on-one-GPU line (fp32, +-5) time: B = 32768,
    TWO 64-trajectory wide tiles per CU (k_solve_wide<.., 2>; the test above runs one per CU).  64 trajectories of the batch against
    the oracle (the float twin for fp32) in each of 10 iterations; the bins go to the tracked statistics file."""
    from ilqr_amd import BatchILQR, capi
    B, T = 32768, 499
    g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, dtype=dtype)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_wide"
    x0 = acrobot_x0(B)
    if dtype == "f32":
        x0 = x0.astype(np.float32).astype(np.float64)
    NIT = 10
    r = sampled_walk(oracle, oracle.Model("acrobot", u_lim=lim), g, x0, np.zeros((B, T, 1)), DT, NIT, precision=dtype)
    print("B=32768 %s sampled walk:" % dtype, publish("bench saturated acrobot T=499 B=32768 +-%g %s (k_solve_wide, two tiles per CU)" % (lim, "fp64" if dtype == "f64" else "fp32"),
                                                    r, B=B, T=T, u_lim=lim, n_sample=len(r["sel"]), precision=dtype))
    # (10 iterations: the first, where 60 % of the fp64 trajectories need the stage-by-stage argument, weighs 2.5 x what it does in a 25-iteration
    #  walk.  Recorded: fp64 0.67 plain / 0.31 on the device's records)
    assert_walk(r, NIT, min_plain_it0=0.95 if dtype == "f64" else 0.8, min_plain=0.58 if dtype == "f64" else 0.75, max_on_records=0.42,
                max_amplified=None if dtype == "f64" else r["checked"] // 8)  # (recorded amplified share: fp64 0.003, fp32 0.083)
    assert r["unresolved"] == 0, r["unresolved"]
    g.close()
