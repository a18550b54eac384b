"""ilqr_iterate runs whole iterations of a tile in one persistent kernel (k_solve_hex, k_solve_tile, k_solve_wide: the fused sweep + backward
pass, producer wavefronts + the quad backward wavefront of a tile sharing a CU and an LDS ring, then the rollouts
and the accept logic, again and again); ILQR_FLAG_STAGED launches the same two phases as kernels of their own per
iteration (k_sweep_backward, k_rollout); the stage calls, and ILQR_FLAG_UNFUSED, run sweep and backward pass as
two kernels with the records in HBM.  All routes execute the same device functions on the same data, so
everything they leave behind must be bit-identical."""
import numpy as np
import pytest

from tests.util import acrobot_x0, integrator_x0

pytestmark = pytest.mark.gpu
DT = 0.02


def _state(g):
    xs, us = g.trajectory()
    k, K = g.gains()
    d = g.derivatives()
    st, it, al = g.status()
    lam, dlam = g.lambdas()
    return dict(xs=xs, us=us, k=k, K=K, cost=g.cost(), st=st, it=it, al=al, lam=lam, dlam=dlam,
                gnorm=g.gnorm(), **{"d_" + n: a for n, a in d.items()})


def _same(a, b):
    for n in a:
        assert np.array_equal(a[n], b[n], equal_nan=True), n


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("B", [5, 48, 200])  # partial tile, whole tiles, many tiles
def test_acrobot_fused_equals_unfused(B, fixed):
    from ilqr_amd import BatchILQR, capi
    T = 120
    x0 = acrobot_x0(B, scale=0.3, seed=5)
    u0 = np.zeros((B, T, 1))
    base = capi.FLAG_FIXED_WORK if fixed else 0
    out = []
    for fl in (0, capi.FLAG_STAGED, capi.FLAG_UNFUSED):
        g = BatchILQR("acrobot", B, T, DT, u_min=-1.5, u_max=1.5, flags=base | fl)
        g.init_traj(x0, u0)
        g.iterate(6)
        out.append(_state(g))
        g.close()
    _same(out[0], out[1])
    _same(out[0], out[2])


def test_integrator_fused_equals_unfused_through_termination():
    """Normal mode until every trajectory has left its loop: covers wavefronts whose trajectories
    are all finished (the consumer returns at once and must still release its producers) and the
    lambda-retry passes that re-read records produced earlier in the same launch."""
    from ilqr_amd import BatchILQR, capi
    B, T = 40, 99
    x0 = integrator_x0(B)
    u0 = np.zeros((B, T, 2))
    out = []
    for fl in (0, capi.FLAG_STAGED, capi.FLAG_UNFUSED):
        g = BatchILQR("integrator", B, T, DT, goal=[1.0, 0.5, 0.0, 0.0], flags=fl)
        g.generate_trajectory(x0, u0)
        assert g.count_running() == 0
        out.append(_state(g))
        g.close()
    _same(out[0], out[1])
    _same(out[0], out[2])


def test_stage_kernel_name_reports_the_fused_kernel():
    from ilqr_amd import BatchILQR, capi
    g = BatchILQR("acrobot", 16, 10, DT)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("backward")) == b"k_sweep_backward"
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_hex"  # m = 1: four matrix-core chains per tile
    g.close()
    g = BatchILQR("integrator", 16, 10, DT)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_tile"  # m = 2: the quad chain
    g.close()
    g = BatchILQR("acrobot", 16, 10, DT, route=capi.ROUTE_QUAD_CHAIN)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_tile"
    g.close()
    g = BatchILQR("acrobot", 16, 10, DT, flags=capi.FLAG_STAGED)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b""
    g.close()
    g = BatchILQR("acrobot", 16, 10, DT, flags=capi.FLAG_UNFUSED)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("backward")) == b"k_backward_q"
    g.close()


@pytest.mark.parametrize("T", [1, 2, 3, 5, 11, 12, 13, 25])
def test_short_horizons_fused_equals_unfused(T):
    """Horizons around the producers' round size (12 knots), the ring size (24) and the candidate
    checkpoint spacing (8): partial rounds, a single knot, B = 1."""
    from ilqr_amd import BatchILQR, capi
    for B in (1, 19):
        x0 = acrobot_x0(B, scale=0.2, seed=T)
        u0 = np.full((B, T, 1), 0.3)
        out = []
        for fl in (0, capi.FLAG_STAGED, capi.FLAG_UNFUSED):
            g = BatchILQR("acrobot", B, T, DT, u_min=-1.0, u_max=1.0, flags=fl)
            g.init_traj(x0, u0)
            g.iterate(4)
            out.append(_state(g))
            g.close()
        _same(out[0], out[1])
        _same(out[0], out[2])
        assert np.all(np.isfinite(out[0]["cost"]))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_two_tiles_per_cu_variants_equal_unfused(name, dtype, monkeypatch):
    """The two-tiles-per-CU instantiations -- k_solve_tile<.., 2> (persistent: <= 256 registers, 72 KB ring, one register
    set of records, roles assigned by SIMD; what ilqr_iterate picks for every B > 16 x #CU) and, with ILQR_FLAG_STAGED,
    k_sweep_backward<1 producer, 60 KB> -- forced here on a small batch, normal mode until every trajectory has left its
    loop (lambda retries make the producers sweep again), against the two-kernel route: bit-identical."""
    from ilqr_amd import BatchILQR, capi
    B, T = 37, 61
    if name == "acrobot":
        x0, nu, kw = acrobot_x0(B, scale=0.3, seed=9), 1, dict(u_min=-1.5, u_max=1.5, params=dict(max_iter=12))
    else:
        x0, nu, kw = integrator_x0(B), 2, dict(goal=[1.0, 0.5, 0.0, 0.0])
    u0 = np.zeros((B, T, nu))
    out = []
    for fl, route in ((0, capi.ROUTE_TWO_TILES_PER_CU), (capi.FLAG_STAGED, capi.ROUTE_TWO_TILES_PER_CU), (capi.FLAG_UNFUSED, 0)):
        g = BatchILQR(name, B, T, DT, flags=fl, dtype=dtype, route=route, **kw)
        g.generate_trajectory(x0, u0)
        out.append(_state(g))
        g.close()
    _same(out[0], out[1])
    _same(out[0], out[2])


@pytest.mark.parametrize("occ", ["1", "2"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("B,T", [(37, 61), (64, 5), (130, 13), (300, 120), (5, 1)])
def test_wide_tiles_equal_unfused(B, T, dtype, occ, monkeypatch):
    """k_solve_wide (kernels_wide.hpp: 64-trajectory tiles, thread-per-trajectory backward chain, what ilqr_iterate picks
    once every CU gets a wide tile) forced on small batches -- ragged wide tiles, partial producer rounds, horizons shorter
    than the ring -- in normal mode until every trajectory has left its loop (lambda retries, abandoned passes), against
    the two-kernel route: every array and scalar bit-identical.  occ: one wide tile per CU (8 wavefronts, 148 KB ring) or two
    (4 wavefronts each, 74 KB ring = three slots in fp64, roles by SIMD)."""
    from ilqr_amd import BatchILQR, capi
    wide = capi.ROUTE_WIDE_TILES | (capi.ROUTE_WIDE_ONE_PER_CU if occ == "1" else capi.ROUTE_WIDE_TWO_PER_CU)
    x0 = acrobot_x0(B, scale=0.3, seed=B + T)
    u0 = np.zeros((B, T, 1))
    kw = dict(u_min=-1.5, u_max=1.5, params=dict(max_iter=14), dtype=dtype)
    sv = capi.STAGE_NAMES.index("solve")
    out = []
    for fl, env in ((0, wide), (capi.FLAG_UNFUSED, 0)):
        g = BatchILQR("acrobot", B, T, DT, flags=fl, route=env, **kw)
        if env:
            assert g.lib.ilqr_stage_kernel_name(g.h, sv) == b"k_solve_wide"
        g.init_traj(x0, u0)
        g.iterate(3)
        s3 = _state(g)
        g.generate_trajectory()
        out.append(dict(_state(g), **{"i3_" + n: a for n, a in s3.items()}))
        g.close()
    _same(out[0], out[1])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("B,T", [(64, 40), (130, 99), (37, 3), (300, 100)])
def test_wide_tiles_two_controls_equal_unfused(B, T, dtype):
    """k_solve_wide2 (kernels_wide2.hpp: the 64-trajectory tile with the thread-per-trajectory chain for m = 2, the 2 x 2 box-QP
    per thread, four wavefronts per tile) forced on small double-integrator batches -- ragged tiles, horizons shorter than the ring
    -- in fixed-iteration and in normal mode until every trajectory has left its loop, against the two-kernel route (the quad
    kernel): every array and scalar bit-identical."""
    from ilqr_amd import BatchILQR, capi
    x0 = integrator_x0(B)
    u0 = np.zeros((B, T, 2))
    kw = dict(u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0], params=dict(max_iter=14), dtype=dtype)
    sv = capi.STAGE_NAMES.index("solve")
    out = []
    for fl, route in ((0, capi.ROUTE_WIDE_TILES), (capi.FLAG_UNFUSED, 0)):
        g = BatchILQR("integrator", B, T, DT, flags=fl, route=route, **kw)
        if route:
            assert g.lib.ilqr_stage_kernel_name(g.h, sv) == b"k_solve_wide2"
        g.init_traj(x0, u0)
        g.iterate(3)
        s3 = _state(g)
        g.generate_trajectory()
        out.append(dict(_state(g), **{"i3_" + n: a for n, a in s3.items()}))
        g.close()
    _same(out[0], out[1])


def test_route_selection_by_batch_size(monkeypatch):
    """Route selection by batch size (ilqr_desc.assume_cus scales the thresholds down to test sizes): up to one tile per CU
    the persistent kernel with a CU per tile, up to two per CU the persistent kernel with two tiles per CU, beyond that -- at ANY
    batch size -- persistent wide tiles (the records never reach HBM); with ILQR_FLAG_STAGED the per-stage kernels, and two kernels beyond two tiles per CU.
    Every route leaves the same bits."""
    from ilqr_amd import BatchILQR, capi
    cus = 6
    T = 20
    bw, sv = capi.STAGE_NAMES.index("backward"), capi.STAGE_NAMES.index("solve")
    for B in (16 * cus + 16, 32 * cus, 48 * cus + 3, 80 * cus + 3):  # one tile more than one per CU; two; three per CU; five per CU and a ragged last tile
        x0 = acrobot_x0(B, scale=0.3, seed=4)
        u0 = np.zeros((B, T, 1))
        out = []
        for fl in (0, capi.FLAG_STAGED, capi.FLAG_UNFUSED):
            g = BatchILQR("acrobot", B, T, DT, u_min=-1.5, u_max=1.5, flags=fl, assume_cus=cus)
            name = lambda st: g.lib.ilqr_stage_kernel_name(g.h, st)
            if fl == 0:  # up to two tiles per CU the 16-trajectory kernel, beyond that wide (64-trajectory) tiles
                assert name(sv) == (b"k_solve_tile<2>" if B <= 32 * cus else b"k_solve_wide")
            elif fl == capi.FLAG_STAGED:
                assert name(sv) == b"" and name(bw) == (b"k_sweep_backward" if B <= 32 * cus else b"k_backward_q")
            else:
                assert name(bw) == b"k_backward_q"
            g.init_traj(x0, u0)
            g.iterate(3)
            out.append(_state(g))
            g.close()
        _same(out[0], out[1])
        _same(out[0], out[2])
    g = BatchILQR("acrobot", 64, 4, DT, assume_cus=cus)  # four tiles on "six CUs"
    assert g.lib.ilqr_stage_kernel_name(g.h, sv) == b"k_solve_hex"
    g.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("B,T,lim", [(64, 60, 1.5), (37, 123, 0.4), (200, 499, 1.5), (5, 1, 1.5), (16, 7, 5.0)])
def test_matrix_core_chains_equal_the_quad_chain(B, T, lim, dtype):
    """One tile per CU, m = 1: k_solve_hex (four chains per tile on the matrix cores -- v_mfma_f64_4x4x4_4b_f64 --, 16 step sizes
    per Armijo pass, the commit right after the line search) against k_solve_tile<1> (one 4-lane DPP chain, the quad search, the
    commit on the producers' way) and the two-kernel route: every array and scalar bit-identical, through late iterations (lambda
    retries, slow QP exits), partial tiles and sub-tiles, horizons shorter than a producer round."""
    from ilqr_amd import BatchILQR, capi
    x0 = acrobot_x0(B, scale=0.6, seed=11)
    if dtype == "f32":
        x0 = x0.astype(np.float32).astype(np.float64)
    u0 = np.zeros((B, T, 1))
    out = []
    for fl, route, name in ((0, 0, b"k_solve_hex"), (0, capi.ROUTE_QUAD_CHAIN, b"k_solve_tile"), (capi.FLAG_UNFUSED, 0, None)):
        g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, flags=fl, dtype=dtype, route=route, params=dict(max_iter=40))
        if name:
            assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == name
        g.init_traj(x0, u0)
        g.iterate(3)
        s = _state(g)
        g.iterate(1)
        g.generate_trajectory()
        s.update({"end_" + n: a for n, a in _state(g).items()})
        out.append(s)
        g.close()
    _same(out[0], out[1])
    _same(out[0], out[2])


@pytest.mark.parametrize("name", ["acrobot", "integrator"])
def test_committed_trajectory_is_the_rollout_that_was_scored(name):
    """Candidate states are not stored: the commit (k_commit, the sweep's fused commit) and the getter
    (k_unpack_cand) re-integrate them from checkpoints with their own inlined copies of
    integrate_dynamics.  The committed trajectory must be, bit for bit, the rollout whose cost was
    accepted -- whatever FMA-contraction choices the compiler made in each kernel: (i) commit == getter,
    (ii) an open-loop replay of the committed controls by yet another instantiation of the rollout kernel
    (init_traj) reproduces the committed states and the accepted cost exactly."""
    from ilqr_amd import BatchILQR
    B, T = 53, 77
    if name == "acrobot":
        x0, nu, kw = acrobot_x0(B, scale=0.5, seed=21), 1, dict(u_min=-1.5, u_max=1.5)
    else:
        x0, nu, kw = integrator_x0(B), 2, dict(goal=[1.0, 0.5, 0.0, 0.0])
    g = BatchILQR(name, B, T, DT, **kw)
    g.init_traj(x0, np.zeros((B, T, nu)))
    for n_it in (1, 2):  # 2: the first accept is committed by the second iteration's sweep
        g.init_traj(x0, np.zeros((B, T, nu)))
        g.iterate(n_it)
        xs, us = g.trajectory()
        cost = g.cost()
        st, it, al = g.status()
        assert (al >= 0).sum() > B // 2
        for a in np.unique(al[al >= 0]):
            xa, ua = g.candidate(int(a))
            sel = al == a
            assert np.array_equal(xa[sel], xs[sel]) and np.array_equal(ua[sel], us[sel]), (n_it, a)
        g2 = BatchILQR(name, B, T, DT, **kw)
        c2 = g2.init_traj(x0, us)
        xs2, us2 = g2.trajectory()
        g2.close()
        assert np.array_equal(xs2, xs), n_it
        assert np.array_equal(c2[al >= 0], cost[al >= 0]), n_it
    g.close()
