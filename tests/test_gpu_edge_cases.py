"""Edge cases through the C ABI: ragged batch sizes (B not a multiple of the 16-trajectory tile or
the 64-lane wavefront, B = 1), horizons around the candidate-chunk size (T = 1, 7, 8, 9, 16, 17:
the candidate checkpoints sit at every 8th knot), both backward kernels agreeing, and two handles
driven from two host threads at once."""
import numpy as np
import pytest

from tests.parity import assert_free_run, gains_knot_err, walk_both
from tests.util import TOL, acrobot_x0, integrator_x0, mat, relerr

pytestmark = pytest.mark.gpu
DT = 0.02


@pytest.mark.parametrize("T", [1, 2, 7, 8, 9, 16, 17, 33])
@pytest.mark.parametrize("name,B", [("acrobot", 19), ("integrator", 5)])
def test_horizons_around_chunk_size(oracle, name, B, T):
    from ilqr_amd import ALPHAS, BatchILQR
    if name == "acrobot":
        om = oracle.Model("acrobot", u_lim=1.5)
        g = BatchILQR("acrobot", B, T, DT, u_min=-1.5, u_max=1.5)
        x0 = acrobot_x0(B, seed=T)
    else:
        om = oracle.Model("integrator", goal=[1, .5, 0, 0])
        g = BatchILQR("integrator", B, T, DT, goal=[1, .5, 0, 0])
        x0 = integrator_x0(B, seed=T)
    u0 = np.random.default_rng(T).normal(size=(B, T, om.nu)) * 0.3
    c0 = g.init_traj(x0, u0)
    xs_o, us_o, c_o = oracle.batch_rollout(om, x0, u0, DT)
    xs, us = g.trajectory()
    assert relerr(xs, xs_o) < TOL and np.allclose(c0, c_o, rtol=TOL)
    # candidates: every alpha's trajectory must come back intact through the chunked layout
    g.compute_derivatives()
    g.backward_step()
    k, K = g.gains()
    cc = g.rollout_candidates()
    for a in (0, 5, 10):
        xa, ua, ca = oracle.batch_rollout(om, x0, us_o + ALPHAS[a] * k, DT, xs_nom=xs_o, K=K)
        xg, ug = g.candidate(a)
        assert relerr(xg, xa) < TOL and np.abs(ug - ua).max() <= TOL * (1.0 + np.abs(ua).max()), (a, T)
        assert np.allclose(cc[:, a], ca, rtol=TOL)
    # full iterations (exercise accept + fused commit + flush) stay in step with the oracle
    g.init_traj(x0, u0)
    g.iterate(2)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=2)
    cost2 = g.cost()
    xs2, us2 = g.trajectory()
    st, it, al = g.status()
    # every step checked against the oracle; a trajectory that left the oracle's path did so at a proven tie
    tied, r_o, r_g = walk_both(oracle, om, g, x0, u0, DT, 2)
    ok = assert_free_run(cost2, ro["cost"], tied, (name, T))
    assert relerr(xs2[ok], ro["xs"][ok]) < 1e-5
    assert np.array_equal(it[ok], ro["iters"][ok])


@pytest.mark.parametrize("B", [1, 15, 16, 17, 63, 64, 65, 130])
def test_ragged_batches(oracle, B):
    from ilqr_amd import BatchILQR
    T = 40
    om = oracle.Model("acrobot", u_lim=1.5)
    x0 = acrobot_x0(max(B, 2), seed=B)[:B] * 0.3
    u0 = np.zeros((B, T, 1))
    g = BatchILQR("acrobot", B, T, DT, u_min=-1.5, u_max=1.5)
    g.init_traj(x0, u0)
    g.iterate(3)
    ro = oracle.batch_solve(om, x0, u0, DT, max_iters=3)
    cost3 = g.cost()
    k, K = g.gains()
    xs, us = g.trajectory()
    tied, r_o, r_g = walk_both(oracle, om, g, x0, u0, DT, 3)
    ok = assert_free_run(cost3, ro["cost"], tied, B)
    if ok.any():  # gains of the third backward pass, per knot (two iterations of amplification behind them)
        assert gains_knot_err(k[ok], K[ok], ro["k"][ok], ro["K"][ok], us[ok]).max() < 1e-4


def test_backward_kernel_variants_agree(oracle):
    """quad-per-trajectory (default) vs thread-per-trajectory backward kernels on the same state."""
    from ilqr_amd import BatchILQR, capi
    B, T = 70, 80
    x0 = acrobot_x0(B)
    res = []
    for flags in (0, capi.FLAG_BACKWARD_THREAD_PER_TRAJ):
        g = BatchILQR("acrobot", B, T, DT, u_min=-1.5, u_max=1.5, flags=flags)
        g.init_traj(x0, np.zeros((B, T, 1)))
        g.iterate(2)
        g.compute_derivatives()
        g.backward_step()
        res.append((g.gains(), g.dV(), g.lambdas()[0], g.gnorm()))
    (k0, K0), dV0, l0, gn0 = res[0]
    (k1, K1), dV1, l1, gn1 = res[1]
    same = np.isclose(l0, l1)
    assert same.mean() > 0.9
    assert relerr(k0[same] + 1, k1[same] + 1) < 1e-6 and relerr(K0[same] + 1, K1[same] + 1) < 1e-5
    assert np.allclose(gn0[same], gn1[same], rtol=1e-9)


def test_handles_are_independent_across_host_threads():
    """The reference is not re-entrant (file-static lambda/dlambda, fixed output file);
    here every handle owns its state and its stream: two solves driven from two host threads at the
    same time give exactly what they give one after the other."""
    import threading
    from ilqr_amd import BatchILQR
    jobs = [("acrobot", 48, 120, dict(u_min=-1.5, u_max=1.5), acrobot_x0(48, scale=0.3, seed=1)),
            ("integrator", 33, 99, dict(goal=[1.0, 0.5, 0.0, 0.0]), integrator_x0(33))]

    def run(job, out, i):
        name, B, T, kw, x0 = job
        g = BatchILQR(name, B, T, 0.02, **kw)
        g.init_traj(x0, np.zeros((B, T, g.nu)))
        for _ in range(6):
            g.iterate(2)
        xs, us = g.trajectory()
        out[i] = (g.cost(), xs, us, g.status())
        g.close()

    seq = [None, None]
    for i, job in enumerate(jobs):
        run(job, seq, i)
    par = [None, None]
    th = [threading.Thread(target=run, args=(job, par, i)) for i, job in enumerate(jobs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for a, b in zip(seq, par):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        for u, v in zip(a[3], b[3]):
            assert np.array_equal(u, v)


def test_trajectory_getter_fills_caller_arrays():
    from ilqr_amd import BatchILQR
    B, T = 7, 13
    g = BatchILQR("acrobot", B, T, DT)
    g.init_traj(acrobot_x0(B, seed=2), np.full((B, T, 1), 0.2))
    xs, us = g.trajectory()
    oxs, ous = np.full((B, T + 1, 4), np.nan), np.full((B, T, 1), np.nan)
    rx, ru = g.trajectory(out=(oxs, ous))
    assert rx is oxs and ru is ous and np.array_equal(oxs, xs) and np.array_equal(ous, us)
    with pytest.raises(ValueError):
        g.trajectory(out=(np.zeros((B, T, 4)), ous))
    with pytest.raises(ValueError):
        g.trajectory(out=(oxs.astype(np.float32), ous))
    g.close()
