"""The double-integrator (n = 4, m = 2: the reference's default example, include/double_integrator.h:16-48, src/run_ilqr.cpp:22-36)
lines of bench.py --extra-configs walked against the ORACLE at their own sizes, in the fixed-work mode the bench times:
  * T = 100, B = 4096  -- the quad chain with box_qp2 (k_solve_tile), fp64 and fp32
  * T = 100, B = 32768 -- k_solve_wide2 (64-trajectory tiles, thread-per-trajectory chain with the 2 x 2 box-QP), fp64 and fp32
and k_solve_wide2 forced on small ragged batches against the oracle (round 5 compared it with the unfused route only:
tests/test_gpu_fused_sweep.py::test_wide_tiles_two_controls_equal_unfused -- a self-comparison)."""
import numpy as np
import pytest

from tests.parity import assert_walk, publish, sampled_walk, walk_iterations
from tests.util import integrator_x0

pytestmark = pytest.mark.gpu
DT = 0.02
GOAL = [1.0, 0.5, 0.0, 0.0]


def bench_x0(B):
    """bench.py's integrator batch (rng 4321: the same draw as tests.util.integrator_x0)."""
    x0 = integrator_x0(B)
    rd = np.random.default_rng(4321)
    assert np.array_equal(x0, rd.uniform(-1, 1, size=(B, 4)) * np.array([1.5, 1.5, 0.5, 0.5]))
    return x0


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("B,kernel", [(4096, b"k_solve_tile"), (32768, b"k_solve_wide2")])
def test_integrator_bench_lines_walked_against_the_oracle(oracle, B, kernel, dtype):
    from ilqr_amd import BatchILQR, capi
    T, lim, NIT = 100, 0.5, 10
    g = BatchILQR("integrator", B, T, DT, u_min=-lim, u_max=lim, goal=GOAL, dtype=dtype, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=NIT + 2))
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == kernel
    x0 = bench_x0(B)
    if dtype == "f32":
        x0 = x0.astype(np.float32).astype(np.float64)
    om = oracle.Model("integrator", goal=GOAL, u_lim=lim)
    r = sampled_walk(oracle, om, g, x0, np.zeros((B, T, 2)), DT, NIT, precision=dtype, fixed_work=True)
    doc = publish("bench integrator T=100 B=%d +-0.5 %s (%s)" % (B, "fp64" if dtype == "f64" else "fp32", kernel.decode()), r,
                  B=B, T=T, u_lim=lim, n_sample=len(r["sel"]), precision=dtype)
    print("integrator B=%d %s sampled walk:" % (B, dtype), {k: doc[k] for k in ("checked", "plain_fraction", "plain_or_plain_on_device_records_fraction", "totals")})
    # linear dynamics, quadratic cost, T = 100: nothing here amplifies rounding -- the plain end-to-end share is held high (fp64
    # recorded: 0.99).  fp32 in FIXED-WORK mode: the integrator has converged after four iterations and keeps iterating; the cost
    # change of every candidate of a converged float trajectory is float rounding, so the sign of z (ilqr_core.cpp:199-206) is
    # noise on both sides -- PROVEN line-search ties (tests/parity.py), 20 % of the checked trajectory-iterations from iteration 4
    # on (recorded: plain 0.80, ties 0.20, nothing else); the bound on ties is widened for that dtype, not the proof.
    if dtype == "f64":
        assert_walk(r, NIT, min_plain_it0=0.95, min_plain=0.90, max_on_records=0.10)
    else:
        assert_walk(r, NIT, min_plain_it0=0.95, min_plain=0.70, max_on_records=0.10, tied_div=3, max_amplified=r["checked"] // 8)
    assert r["unresolved"] == 0, r["unresolved"]
    assert g.count_running() == B
    g.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("B,T,iters", [(130, 99, 16), (37, 3, 4), (300, 100, 8)])
def test_wide_tiles_two_controls_against_the_oracle(oracle, B, T, iters, dtype):
    """ILQR_ROUTE_WIDE_TILES forces k_solve_wide2 on small, ragged double-integrator batches (horizons shorter than the ring too):
    every iteration of both drives of walk_iterations against the oracle, normal mode -- lambda growth, the cost-change exit, no-step
    iterations."""
    from ilqr_amd import BatchILQR, capi
    lim = 0.5
    g = BatchILQR("integrator", B, T, DT, u_min=-lim, u_max=lim, goal=GOAL, dtype=dtype, route=capi.ROUTE_WIDE_TILES)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_wide2"
    om = oracle.Model("integrator", goal=GOAL, u_lim=lim)
    x0 = integrator_x0(B)
    if dtype == "f32":
        x0 = x0.astype(np.float32).astype(np.float64)
    u0 = np.zeros((B, T, 2))
    for drive in ("oracle", "gpu"):
        r = walk_iterations(oracle, om, g, x0, u0, DT, iters, drive=drive, precision=dtype)
        assert r["checked"] >= B * min(iters, 2), r["checked"]
        ties = r["ties_backward"] + r["ties_search"] + r["ties_stop"]
        # (fp32: a cost change of rounding size is 1e-7 of the cost, not 1e-16 -- proven ties are no rare events near the optimum;
        #  the same bound as tests/test_gpu_fp32.py::test_iterations_teacher_forced)
        assert ties + r["conditioned_branch"] <= (max(2, r["checked"] // 20) if dtype == "f64" else max(4, r["checked"] // 3)), r
        assert r["cond_over10"] <= max(1, r["checked"] // 50), r
        assert r["unresolved"] == 0, r
    g.close()
