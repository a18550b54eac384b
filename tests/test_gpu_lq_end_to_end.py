"""The synthetic LQ model (BASELINE.json configs[4]: x+ = x + (Ax + Bu) dt, quadratic costs,
n = 32, m = 16 at full size) end to end on the device: its device twin runs the rollout, the
finite-difference sweep, the generic backward pass and the 11-alpha line search (generic.hpp,
backward_wave.hpp), each stage compared with the oracle on the same inputs through the C ABI.
Sizes are the ones the oracle finishes in seconds (the FD Hessian of a 32-dimensional quadratic is
6 Mflop per knot); full dimensions n = 32 / m = 16 are covered at a small batch."""
import numpy as np
import pytest

from tests.util import TOL, mat, relerr, relerr_abs
from tests.parity import assert_walk, check_backward, publish, sampled_walk, walk_iterations

pytestmark = pytest.mark.gpu
DT = 0.02


def lq_mats(n, m, seed=7):
    # SURVEY.md 8(d) cfg 5: A = -I + 0.1 N(0,1)/sqrt(n), B = N(0,1)/sqrt(n), Q = I, R = 0.1 I
    rng = np.random.default_rng(seed)
    A = -np.eye(n) + 0.1 * rng.normal(size=(n, n)) / np.sqrt(n)
    Bm = rng.normal(size=(n, m)) / np.sqrt(n)
    return A, Bm, np.eye(n), 0.1 * np.eye(m), np.eye(n)


def dense_mats(n, m, seed=11):
    """Non-diagonal symmetric weights: exercises every entry of the cost Hessians."""
    rng = np.random.default_rng(seed)
    A = -np.eye(n) + 0.3 * rng.normal(size=(n, n)) / np.sqrt(n)
    Bm = rng.normal(size=(n, m)) / np.sqrt(n)

    def spd(k, s):
        W = rng.normal(size=(k, k)) / np.sqrt(k)
        return s * (np.eye(k) + 0.5 * (W + W.T) * 0.5)
    return A, Bm, spd(n, 1.0), spd(m, 0.2), spd(n, 3.0)


def make(oracle, mats, B, T, lim=1.0, flags=0):
    from ilqr_amd import BatchILQR
    n, m = mats[1].shape
    om = oracle.Model("lq", lq=mats, u_lim=lim)
    g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats, flags=flags)
    assert (g.nx, g.nu) == (n, m)
    return om, g


@pytest.mark.parametrize("n,m,B,T,dense", [(32, 16, 6, 12, False), (32, 16, 4, 8, True), (6, 3, 40, 25, True), (5, 2, 70, 9, True),
                                            (1, 1, 3, 1, True), (2, 16, 5, 2, True), (31, 1, 2, 3, True)])
def test_lq_stages_match_oracle(oracle, n, m, B, T, dense):
    mats = dense_mats(n, m) if dense else lq_mats(n, m)
    om, g = make(oracle, mats, B, T)
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.3
    # init_traj: open-loop rollout
    c0 = g.init_traj(x0, u0)
    xs_o, us_o, c_o = oracle.batch_rollout(om, x0, u0, DT)
    xs, us = g.trajectory()
    assert relerr(xs, xs_o) < 1e-12 and np.array_equal(us, us_o)
    assert np.max(np.abs(c0 - c_o) / np.abs(c_o)) < 1e-12
    # finite-difference sweep on the oracle's trajectory (teacher forcing)
    g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=c_o)
    g.compute_derivatives()
    d = g.derivatives()
    do = oracle.batch_derivatives(om, xs_o, us_o, DT)
    for name in ("fx", "fu", "cx", "cu"):
        ref = do[name] if name in ("cx", "cu") else mat(do[name])
        assert relerr(d[name], ref) < TOL, name
    # second differences of a quadratic: exact value + rounding noise ~ 1e-16 f / eps^2
    for name in ("cxx", "cuu", "cxu"):
        ref = mat(do[name])
        assert relerr_abs(d[name][:, :T], ref[:, :T], 1e-2) < TOL, name
    assert relerr_abs(d["cxx"][:, T], mat(do["cxx"])[:, T], 1e-2) < TOL
    assert np.all(d["fx"][:, T] == 0) and np.all(d["fu"][:, T] == 0) and np.all(d["cu"][:, T] == 0)
    # backward pass from the oracle's derivatives
    g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
    k_prev = rng.normal(size=(B, T, m)) * 0.1
    g.set_gains(k=k_prev, K=np.zeros((B, T, m, n)))
    g.set_lambda(1.0, 1.0)
    div = g.backward_pass()
    ro = oracle.batch_backward(om, us_o, do, k_prev=k_prev, lam=1.0)
    k, K = g.gains()
    Ko = mat(ro["K"])
    check_backward(oracle, om, us_o, do, k_prev, 1.0, k, K, g.dV(), div, ro, max_ties=max(1, B // 8), max_over10=max(1, B // 50))  # gains: per knot
    # the 11 closed-loop rollouts of the line search, with the oracle's gains
    g.set_gains(k=ro["k"], K=Ko)
    costs = g.rollout_candidates()
    from oracle.oracle import ALPHAS
    for a in (0, 4, 10):
        xa, ua, ca = oracle_closed_loop(oracle, om, x0, xs_o, us_o, ro["k"], Ko, ALPHAS[a])
        fin = np.isfinite(ca)
        assert np.max(np.abs(costs[fin, a] - ca[fin]) / np.abs(ca[fin])) < 1e-9, a
        # the matrix-core search keeps its eleven rollouts whole: ilqr_get_candidate hands them out (knot for knot the oracle's)
        xc, uc = g.candidate(a)
        assert relerr(xc[fin], xa[fin]) < 1e-9 and relerr_abs(uc[fin], ua[fin], 1e-3) < 1e-9, a
    g.close()


@pytest.mark.parametrize("n,m,B,T", [(32, 16, 5, 9), (32, 16, 3, 17), (6, 3, 30, 20), (5, 2, 40, 8), (1, 1, 3, 2), (2, 16, 5, 3), (31, 1, 2, 9), (17, 9, 4, 16)])
def test_lq_incremental_finite_differences_equal_the_dense_sweep(n, m, B, T):
    """k_derivatives_lq evaluates every perturbed point of the LQ model by what moved (Q p = Q x + delta_i Q[:, i] + delta_j Q[:, j]: the same
    function value at the same point from 2 n multiply-adds instead of n^2); ILQR_ROUTE_LQ_DENSE_FD keeps the dense matrix-core evaluation of
    every point (k_derivatives_g).  The two sweeps' records: first differences to 1e-9, second differences -- rounding of f amplified by
    1 / 4 eps^2 on both sides -- to the tolerance either has against the oracle; horizons that are not a multiple of the knots per wavefront."""
    from ilqr_amd import BatchILQR, capi
    mats = dense_mats(n, m, seed=23)
    mats = (mats[0], mats[1], mats[2] + 0.1 * np.triu(np.ones((n, n)), 1), mats[3], mats[4])  # a NON-symmetric Q: x'Qx uses rows and columns
    rng = np.random.default_rng(31)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.4
    recs = []
    for route, kernel in ((0, b"k_derivatives_lq"), (capi.ROUTE_LQ_DENSE_FD, b"k_derivatives_g")):
        g = BatchILQR("lq", B, T, DT, u_min=-1.0, u_max=1.0, lq=mats, route=route)
        assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("derivatives")) == kernel
        g.init_traj(x0, u0)
        g.compute_derivatives()
        recs.append(g.derivatives())
        g.close()
    a, b = recs
    for name in ("fx", "fu", "cx", "cu"):
        assert np.abs(a[name] - b[name]).max() <= 1e-9 * max(1.0, np.abs(b[name]).max()), name
    for name in ("cxx", "cuu", "cxu"):
        assert relerr_abs(a[name], b[name], 1e-2) < TOL, name
    for name in a:  # knot T is the generic sweep's on both routes
        assert np.array_equal(a[name][:, T], b[name][:, T]), name


def oracle_closed_loop(oracle, om, x0, xs_nom, us_nom, k, K, alpha):
    """forward_pass(x0, us + alpha k) with feedback K around xs_nom (ilqr_core.cpp:188-190, 305-337)."""
    return oracle.batch_rollout(om, x0, us_nom + alpha * k, DT, xs_nom=xs_nom, K=K)


@pytest.mark.parametrize("n,m,B,T", [(6, 3, 24, 30), (32, 16, 3, 10)])
def test_lq_iterations_match_oracle(oracle, n, m, B, T):
    """A few whole iterations (derivatives -> backward -> search -> accept -> commit), fixed work
    on both sides.  The model is linear, so per-trajectory agreement holds across iterations."""
    from ilqr_amd import capi
    mats = dense_mats(n, m)
    om, g = make(oracle, mats, B, T, flags=capi.FLAG_FIXED_WORK)
    rng = np.random.default_rng(9)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = np.zeros((B, T, m))
    iters = 3
    r = walk_iterations(oracle, om, g, x0, u0, DT, iters, fixed_work=True)  # every deviation a proven tie
    n_ties = len(r["tied"])
    assert r["checked"] == B * iters and n_ties <= max(1, B // 8), r
    g.init_traj(x0, u0)
    g.iterate(iters)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=iters, fixed_work=True)
    cost = g.cost()
    xs, us = g.trajectory()
    rel = np.abs(cost - ro["cost"]) / np.abs(ro["cost"])
    assert (rel >= TOL).sum() <= n_ties and rel.max() < 1e-3, (rel, r)
    ok = rel < TOL
    assert relerr(xs[ok], ro["xs"][ok]) < 1e-5 and relerr_abs(us[ok], ro["us"][ok], 1e-3) < 1e-5
    st, it, al = g.status()
    assert np.all(it == iters)
    g.close()


def test_lq_full_solve_and_warm_start(oracle):
    n, m, B, T = 6, 3, 16, 30
    mats = dense_mats(n, m)
    om, g = make(oracle, mats, B, T)
    rng = np.random.default_rng(2)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = np.zeros((B, T, m))
    c0 = g.init_traj(x0, u0)
    g.generate_trajectory()
    st, it, al = g.status()
    assert g.count_running() == 0 and np.all(st > 0)
    ro = oracle.batch_solve(om, x0, u0, DT)
    cost = g.cost()
    assert np.all(cost <= c0 * (1 + 1e-12))
    rel = np.abs(cost - ro["cost"]) / np.abs(ro["cost"])
    assert (rel < 1e-6).mean() >= 0.75 and rel.max() < 1e-2, rel  # absolute stopping tests: ties stop an iteration apart
    # warm start from perturbed initial states keeps working (ilqr_core.cpp:65-76)
    import ctypes as C
    x1 = np.ascontiguousarray(x0 * 1.01)
    assert g.lib.ilqr_warm_start(g.h, x1.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert g.count_running() == 0 and np.all(np.isfinite(g.cost()))
    g.close()


@pytest.mark.parametrize("n,m,B,T", [(32, 16, 9, 40), (6, 3, 33, 25), (31, 1, 2, 3), (2, 16, 5, 2), (1, 1, 3, 1)])
def test_lq_matrix_core_rollout_equals_thread_per_rollout(n, m, B, T, monkeypatch):
    """k_rollout_lq (one wavefront per trajectory, every product a v_mfma_f64_16x16x4_f64 chain over the
    11 candidate columns) against the generic thread-per-rollout kernel k_rollout_g
    (ILQR_ROUTE_LQ_THREAD_ROLLOUT): the chains run in the order of the scalar sums, so init rollout,
    the 11 search costs, the committed trajectory and everything downstream are bit-identical."""
    from ilqr_amd import BatchILQR, capi
    mats = dense_mats(n, m)
    rng = np.random.default_rng(8)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.3
    out = []
    # default: the matrix-core search keeps its eleven rollouts and the commit is a copy (k_commit_lq); ILQR_ROUTE_LQ_RECOMMIT:
    # no candidate buffers, the accepted rollout is run a second time; ILQR_ROUTE_LQ_THREAD_ROLLOUT: the generic kernel
    for route, kernel in ((0, b"k_rollout_lq"), (capi.ROUTE_LQ_RECOMMIT, b"k_rollout_lq"), (capi.ROUTE_LQ_THREAD_ROLLOUT, b"k_rollout_g")):
        g = BatchILQR("lq", B, T, DT, u_min=-0.4, u_max=0.4, lq=mats, flags=capi.FLAG_ANALYTIC_DERIVATIVES, route=route)
        name = g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("rollout"))
        assert name == kernel
        c0 = g.init_traj(x0, u0)
        g.iterate(3)
        xs, us = g.trajectory()
        k, K = g.gains()
        st, it, al = g.status()
        mid = dict(xs=xs, us=us, cost=g.cost())
        g.generate_trajectory()   # to the end of the solve: per-trajectory exits (commit_idx = -1 for the ones that have left)
        xe, ue = g.trajectory()
        out.append(dict(c0=c0, k=k, K=K, al=al, it=it, end_xs=xe, end_us=ue, end_cost=g.cost(), end_status=g.status()[0], **mid))
        g.close()
    for o in out[1:]:
        for key in out[0]:
            assert np.array_equal(out[0][key], o[key], equal_nan=True), key
    assert np.all(out[0]["cost"] < out[0]["c0"])


@pytest.mark.parametrize("lim", [1.0, 0.3])
def test_lq_full_size_properties(oracle, lim):
    """BASELINE.json configs[4] at full size (n = 32, m = 16, T = 200, B = 8192; limits +-1 as SURVEY.md 8d
    cfg 5 has them, and +-0.3 where far more box-QPs end clamped): size-independent properties of two
    finite-difference iterations."""
    from ilqr_amd import BatchILQR
    n, m, B, T = 32, 16, 8192, 200
    mats = lq_mats(n, m)
    rng = np.random.default_rng(12)
    x0 = rng.uniform(-1, 1, (B, n))
    x0[1] = x0[0]
    x0[B - 1] = x0[0]  # duplicates in other wavefronts / CUs
    g = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats)
    # 24 trajectories of the full batch walked against the oracle, iteration by iteration (tests/parity.py: Sampled)
    NIT = 10
    r = sampled_walk(oracle, oracle.Model("lq", lq=mats, u_lim=lim), g, x0, np.zeros((B, T, m)), DT, NIT, n_sample=24)
    print("configs[4] lim", lim, "sampled walk:", publish("configs[4] LQ n=32 m=16 T=200 B=8192 +-%g fp64 (finite differences)" % lim, r, B=B, T=T, u_lim=lim, n_sample=len(r["sel"])))
    assert_walk(r, NIT, min_plain=0.93, max_on_records=0.05)  # (recorded: 0.97 - 1.00 plain, nothing on the device's records)
    c0 = g.init_traj(x0, np.zeros((B, T, m)))
    g.iterate(2)
    cost = g.cost()
    st, it, al = g.status()
    xs, us = g.trajectory()
    k, K = g.gains()
    g.close()
    assert np.all(np.isfinite(cost)) and np.all(cost <= c0 * (1 + 1e-12))  # monotone (line search)
    assert np.all(cost < c0) and np.mean(cost / c0) < 0.95 and (al >= 0).mean() > 0.99  # an LQ problem: every step helps
    assert cost[0] == cost[1] == cost[B - 1]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(K[0], K[B - 1]) and np.array_equal(us[0], us[B - 1])
    assert np.array_equal(xs[:, 0], x0)
    # (no bound on |us|: the reference's forward pass adds K (x - xs) without clamping, ilqr_core.cpp:316-323)
    assert (np.abs(us) <= lim * (1 + 1e-9)).mean() > 0.9
    # rows of K are zero where the control sits on a limit (boxqp.cpp: only free rows get a gain)
    clamped_rows = (np.abs(K).max(axis=-1) == 0)
    assert (0.001 if lim < 1 else 0.0) < clamped_rows.mean() < 0.9
    print("lim", lim, "clamped gain rows", clamped_rows.mean())
