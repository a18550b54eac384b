"""ILQR_MODEL_USER: a device twin compiled into a build of the library from a header that is NOT part of the library
(examples/user_model_acrobot.hpp, -DILQR_USER_MODEL_HEADER; the reference's Model is an open plugin interface,
include/model.h:6-21).  With the shipped acrobot's parameters the user model must leave, kernel for kernel, the same bits
as the built-in acrobot; with other parameters it must agree with the same model evaluated through host virtuals
(ILQR_MODEL_HOST route of the C++ facade's recipe: rollouts + finite differences by the caller) to 1e-6."""
import numpy as np
import pytest

from tests.util import TOL, acrobot_x0

pytestmark = pytest.mark.gpu
DT = 0.02


@pytest.fixture(scope="module")
def user_lib():
    from ilqr_amd import _build
    return _build.build_user(_build.USER_EXAMPLE_HEADER, _build.USER_EXAMPLE_LIB)


def _state(g):
    xs, us = g.trajectory()
    k, K = g.gains()
    st, it, al = g.status()
    return dict(xs=xs, us=us, k=k, K=K, cost=g.cost(), st=st, it=it, al=al, lam=g.lambdas()[0], gnorm=g.gnorm(), dV=g.dV(), **g.derivatives())


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("flags", ["persistent", "staged", "unfused"])
def test_user_copy_of_the_acrobot_equals_the_builtin_bit_for_bit(user_lib, dtype, flags):
    from ilqr_amd import BatchILQR, capi
    fl = {"persistent": 0, "staged": capi.FLAG_STAGED, "unfused": capi.FLAG_UNFUSED}[flags]
    B, T, lim = 83, 120, 1.5
    x0 = acrobot_x0(B, scale=0.5, seed=31)
    u0 = np.zeros((B, T, 1))
    out = []
    for kw in (dict(model="acrobot"), dict(model="user", lib=user_lib, nx=4, nu=1, user_params=[3.1415, 0, 0, 0, 20, 20])):
        g = BatchILQR(B=B, T=T, dt=DT, u_min=-lim, u_max=lim, dtype=dtype, flags=fl, **kw)
        c0 = g.init_traj(x0, u0)
        g.iterate(5)
        s = _state(g)
        s["c0"] = c0
        g.generate_trajectory()  # ... and to termination
        s["final"] = g.cost()
        s["final_it"] = g.status()[1]
        out.append(s)
        g.close()
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n], equal_nan=True), n


def test_user_model_with_its_own_parameters_matches_the_oracle(user_lib, oracle):
    """Other parameters than the built-in's (goal, terminal weights): one iteration against the oracle's acrobot with the
    same goal is not available (the oracle's acrobot has fixed weights), so the check is structural: the rollout cost is
    the model's own final_cost (computed here in numpy from the returned states), and the finite-difference records of
    the terminal knot equal its analytic Hessian 2 Ks^2 / 2 Kd^2."""
    from ilqr_amd import BatchILQR
    B, T = 20, 40
    goal, Ks, Kd = np.array([1.0, -0.5, 0.2, 0.0]), 7.0, 3.0
    x0 = acrobot_x0(B, scale=0.3, seed=2)
    u0 = np.random.default_rng(0).normal(size=(B, T, 1)) * 0.2
    g = BatchILQR("user", B, T, DT, u_min=-2.0, u_max=2.0, lib=user_lib, nx=4, nu=1, user_params=list(goal) + [Ks, Kd])
    c0 = g.init_traj(x0, u0)
    xs, us = g.trajectory()
    d = goal[None, :] - xs[:, T]
    fin = Ks * Ks * (d[:, 0] ** 2 + d[:, 1] ** 2) + Kd * Kd * (d[:, 2] ** 2 + d[:, 3] ** 2)
    run = (0.1 * 0.1 * us[:, :, 0] ** 2).sum(axis=1)
    assert np.allclose(c0, fin + run, rtol=1e-12)
    g.compute_derivatives()
    rec = g.derivatives()
    H = np.diag([2 * Ks * Ks, 2 * Ks * Ks, 2 * Kd * Kd, 2 * Kd * Kd])
    assert np.allclose(rec["cxx"][:, T], H[None], rtol=1e-5, atol=1e-5 * 2 * Ks * Ks)
    assert np.allclose(rec["cx"][:, T], -2 * np.array([Ks * Ks, Ks * Ks, Kd * Kd, Kd * Kd]) * d, rtol=1e-6, atol=1e-8)
    g.iterate(3)
    assert np.all(g.cost() <= c0 * (1 + 1e-12)) and (g.status()[2] >= 0).mean() > 0.5
    g.close()


def test_builds_say_what_they_carry(user_lib):
    from ilqr_amd import BatchILQR, capi
    assert capi.load().ilqr_has_user_model() == 0 and capi.load(path=user_lib).ilqr_has_user_model() == 1
    with pytest.raises(capi.ILQRError, match="not available in this build"):
        BatchILQR("user", 4, 5, DT, u_min=-1.0, u_max=1.0, nx=4, nu=1)  # the stock library has no user model
    with pytest.raises(capi.ILQRError, match="analytic_record"):
        BatchILQR("user", 4, 5, DT, u_min=-1.0, u_max=1.0, nx=4, nu=1, lib=user_lib, flags=capi.FLAG_ANALYTIC_DERIVATIVES)


@pytest.fixture(scope="module")
def user6_lib():
    from ilqr_amd import _build
    return _build.build_user(_build.USER_EXAMPLE6_HEADER, _build.USER_EXAMPLE6_LIB)


ROUTES6 = ["thread", "wave"]  # a small twin's default (tiled thread kernels) and ILQR_ROUTE_WAVE_PER_TRAJECTORY (the generic kernels)


def _route6(route):
    from ilqr_amd import capi
    return 0 if route == "thread" else capi.ROUTE_WAVE_PER_TRAJECTORY


def _kernels6(g, route):
    from ilqr_amd import capi
    name = lambda st: g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index(st))
    if route == "thread":
        assert (name("derivatives"), name("backward"), name("rollout")) == (b"k_derivatives", b"k_backward_t", b"k_rollout")
    else:
        assert (name("derivatives"), name("rollout")) == (b"k_derivatives_g", b"k_rollout_g")


@pytest.mark.parametrize("route", ROUTES6)
def test_user_model_with_dimensions_of_its_own(user6_lib, oracle, route):
    """A user twin that is not an nx = 4 shape (examples/user_model_linear6.hpp: n = 6, m = 2, written as plain loops over
    its own matrices).  Small as it is (even nx <= 8, nu <= 4) it runs by default in the tiled THREAD kernels -- a thread per knot,
    per trajectory, per rollout, the 6 x 6 algebra in registers --, and under ILQR_ROUTE_WAVE_PER_TRAJECTORY in the generic kernels
    every larger twin gets: thread-per-rollout forward passes, wavefront-per-knot finite differences
    through the model's own dynamics() / cost() / final_cost(), the matrix-core backward pass.  Every iteration of a
    free-running solve against the oracle's LQ model with the same matrices (tests/parity.py: device-driven walk, 1e-6 per
    knot or a proven tie), and against the SHIPPED LQ twin at n = 6, m = 2 (whose sums run in the matrix cores' order:
    agreement to the finite differences' rounding, not to the bit)."""
    from ilqr_amd import BatchILQR, capi
    from tests.parity import walk_iterations
    from tests.test_gpu_lq_end_to_end import dense_mats
    n, m, B, T, lim = 6, 2, 37, 60, 0.4
    mats = dense_mats(n, m, seed=5)
    params = np.concatenate([np.ascontiguousarray(a).ravel() for a in mats])
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.1
    g = BatchILQR("user", B, T, DT, u_min=-lim, u_max=lim, lib=user6_lib, nx=n, nu=m, user_params=params, route=_route6(route))
    _kernels6(g, route)
    om = oracle.Model("lq", lq=mats, u_lim=lim)
    r = walk_iterations(oracle, om, g, x0, u0, DT, 6, drive="gpu")
    print("user n=6 m=2 walk:", {kk: v for kk, v in r.items() if kk not in ("per_iter", "tied")})
    assert r["checked"] >= 3 * B and len(r["tied"]) <= B // 8, r
    # the same problem through the shipped LQ twin
    c_user, (k_user, K_user) = None, (None, None)
    g.init_traj(x0, u0)
    g.iterate(3)
    c_user, (k_user, K_user) = g.cost(), g.gains()
    g2 = BatchILQR("lq", B, T, DT, u_min=-lim, u_max=lim, lq=mats)
    g2.init_traj(x0, u0)
    g2.iterate(3)
    ok = np.abs(c_user - g2.cost()) <= 1e-6 * np.abs(c_user)
    assert ok.mean() > 0.9, ok.mean()  # (a clamp tie on one side moves a trajectory's cost: counted, not hidden)
    k2, K2 = g2.gains()
    assert np.abs(k_user[ok] - k2[ok]).max() <= 1e-6 * max(1.0, np.abs(k2).max())
    # getters, stage calls and the full solve work on this route like on any other
    g.generate_trajectory()
    assert g.count_running() == 0 and np.all(np.isfinite(g.cost())) and np.all(g.cost() <= c_user * (1 + 1e-9))
    if route == "wave":
        with pytest.raises(capi.ILQRError, match="fp64"):
            BatchILQR("user", 4, 5, DT, u_min=-1.0, u_max=1.0, nx=n, nu=m, lib=user6_lib, user_params=params, dtype="f32", route=_route6(route))
    else:  # the tiled kernels have an fp32 mode: float storage and rollouts, the double chain, finite differences in the double twin
        g32 = BatchILQR("user", B, T, DT, u_min=-lim, u_max=lim, lib=user6_lib, nx=n, nu=m, user_params=params, dtype="f32")
        g32.init_traj(x0, u0)
        g32.iterate(3)
        rel = np.abs(g32.cost() - c_user) / np.abs(c_user)
        assert np.median(rel) < 1e-4 and (rel < 1e-2).mean() > 0.9, rel
        g32.close()
    g.close()
    g2.close()


@pytest.mark.parametrize("route", ROUTES6)
def test_generic_user_model_whose_cost_reads_its_own_limits(user6_lib, oracle, route):
    """A Model subclass may read its public u_min / u_max members in cost() (include/model.h:17).  The linear6 example's optional
    penalty wb sum_j (u_j / u_max_j)^2 does: the generic kernels' copy of the model must carry the limits (the rollout cost is
    checked against numpy), and the solve must be the oracle's LQ solve with R + 2 wb diag(1 / u_max^2)."""
    from ilqr_amd import BatchILQR
    from tests.parity import walk_iterations
    from tests.test_gpu_lq_end_to_end import dense_mats
    n, m, B, T, wb = 6, 2, 21, 40, 0.35
    umax = np.array([0.4, 0.7])
    A, Bm, Q, R, Qf = dense_mats(n, m, seed=9)
    params = np.concatenate([np.ascontiguousarray(a).ravel() for a in (A, Bm, Q, R, Qf)] + [[wb]])
    rng = np.random.default_rng(4)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.1
    g = BatchILQR("user", B, T, DT, u_min=-umax, u_max=umax, lib=user6_lib, nx=n, nu=m, user_params=params, route=_route6(route))
    _kernels6(g, route)
    c0 = g.init_traj(x0, u0)
    xs, us = g.trajectory()
    run = 0.5 * (np.einsum("bti,ij,btj->b", xs[:, :T], Q, xs[:, :T]) + np.einsum("bti,ij,btj->b", us, R, us)) + wb * ((us / umax) ** 2).sum(axis=(1, 2))
    fin = 0.5 * np.einsum("bi,ij,bj->b", xs[:, T], Qf, xs[:, T])
    assert np.allclose(c0, run + fin, rtol=1e-12)
    om = oracle.Model("lq", lq=(A, Bm, Q, R + 2 * wb * np.diag(1 / umax ** 2), Qf), u_lim=1.0)
    om.set_limits(-umax, umax)
    r = walk_iterations(oracle, om, g, x0, u0, DT, 4, drive="gpu")
    assert r["checked"] >= 2 * B and len(r["tied"]) <= B // 8, r
    g.close()


@pytest.mark.parametrize("route", ROUTES6)
def test_generic_user_model_analytic_record(user6_lib, route):
    """A user twin's own analytic_record (examples/user_model_linear6.hpp) on the generic kernels (ILQR_FLAG_ANALYTIC_DERIVATIVES): its
    records against the finite-difference sweep's of the same model (second differences of a quadratic: exact value + rounding noise
    ~ 1e-16 f / eps^2), and the solve they drive against the finite-difference solve."""
    from ilqr_amd import BatchILQR, capi
    from tests.test_gpu_lq_end_to_end import dense_mats
    n, m, B, T, wb = 6, 2, 19, 30, 0.2
    umax = np.array([0.5, 0.8])
    mats = dense_mats(n, m, seed=13)
    params = np.concatenate([np.ascontiguousarray(a).ravel() for a in mats] + [[wb]])
    rng = np.random.default_rng(6)
    x0 = rng.uniform(-1, 1, (B, n))
    u0 = rng.normal(size=(B, T, m)) * 0.2
    out = {}
    for name, fl in (("fd", 0), ("exact", capi.FLAG_ANALYTIC_DERIVATIVES)):
        g = BatchILQR("user", B, T, DT, u_min=-umax, u_max=umax, lib=user6_lib, nx=n, nu=m, user_params=params, flags=fl, route=_route6(route))
        _kernels6(g, route)
        g.init_traj(x0, u0)
        g.compute_derivatives()
        d = g.derivatives()
        g.iterate(3)
        k, K = g.gains()
        out[name] = dict(d=d, cost=g.cost(), k=k, K=K)
        g.close()
    fd, ex = out["fd"]["d"], out["exact"]["d"]
    for key in ("fx", "fu", "cx", "cu"):
        assert np.abs(fd[key] - ex[key]).max() <= 1e-8 * max(1.0, np.abs(ex[key]).max()), key
    for key in ("cxx", "cuu"):
        assert np.abs(fd[key] - ex[key]).max() <= 1e-6 * max(1.0, np.abs(ex[key]).max()), key
    assert np.abs(fd["cxu"][:, :T] - ex["cxu"][:, :T]).max() <= 1e-6
    assert np.all(ex["fx"][:, T] == 0) and np.all(ex["cu"][:, T] == 0)
    rel = np.abs(out["fd"]["cost"] - out["exact"]["cost"]) / np.abs(out["exact"]["cost"])
    assert (rel < 1e-6).mean() > 0.9 and rel.max() < 1e-2, rel  # (a clamp tie on one side moves a trajectory: counted, not hidden)
