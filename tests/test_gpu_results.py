"""ABI 5: results without one host round trip per array -- ilqr_get_results_async (pinned and pageable destinations),
ilqr_copy_trajectory_to_device / ilqr_copy_gains_to_device (canonical layouts in caller-owned device memory), and what ABI 5 retired."""
import ctypes as C

import numpy as np
import pytest

from tests.util import acrobot_x0, integrator_x0

pytestmark = pytest.mark.gpu
DT = 0.02


def handles():
    from ilqr_amd import BatchILQR
    from bench import lq_mats
    yield "acrobot f64", BatchILQR("acrobot", 70, 57, DT, u_min=-1.5, u_max=1.5), acrobot_x0(70), 1
    yield "acrobot f32", BatchILQR("acrobot", 33, 40, DT, u_min=-5.0, u_max=5.0, dtype="f32"), acrobot_x0(33).astype(np.float32).astype(np.float64), 1
    yield "integrator", BatchILQR("integrator", 19, 30, DT, u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0]), integrator_x0(19), 2
    yield "lq (generic layout)", BatchILQR("lq", 5, 12, DT, u_min=-1.0, u_max=1.0, lq=lq_mats(8, 3)), np.random.default_rng(0).uniform(-1, 1, (5, 8)), 3


@pytest.mark.parametrize("pinned", [True, False])
def test_results_async_equals_the_getters(pinned):
    for name, g, x0, m in handles():
        g.init_traj(x0, np.zeros((g.B, g.T, m)))
        g.iterate(3)
        xs, us = g.trajectory()
        k, K = g.gains()
        c = g.cost()
        bufs = g.result_buffers(pinned=pinned)
        for a in bufs.values():
            a.fill(np.nan)
        g.results_async(bufs)
        g.synchronize()
        assert np.array_equal(bufs["xs"], xs) and np.array_equal(bufs["us"], us), name
        assert np.array_equal(bufs["k"], k) and np.array_equal(np.swapaxes(bufs["K"], -1, -2), K), name
        assert np.array_equal(bufs["cost"], c), name
        # a subset: NULL pointers are skipped
        part = g.result_buffers(pinned=pinned, K=False)
        part["xs"].fill(np.nan)
        g.results_async(part)
        g.synchronize()
        assert np.array_equal(part["xs"], xs) and np.array_equal(part["cost"], c), name
        # the next upload reuses the staging buffer behind the copies: stream order keeps both intact
        g.results_async(bufs)
        g.init_traj(x0, np.zeros((g.B, g.T, m)))
        g.synchronize()
        assert np.array_equal(bufs["xs"], xs), name
        g.close()


def test_results_into_device_memory():
    import torch
    for name, g, x0, m in handles():
        g.init_traj(x0, np.zeros((g.B, g.T, m)))
        g.iterate(2)
        xs, us = g.trajectory()
        k, K = g.gains()
        dev = torch.device("cuda", 0)
        txs = torch.full((g.B, g.T + 1, g.nx), float("nan"), dtype=torch.float64, device=dev)
        tus = torch.full((g.B, g.T, g.nu), float("nan"), dtype=torch.float64, device=dev)
        tk = torch.full((g.B, g.T, g.nu), float("nan"), dtype=torch.float64, device=dev)
        tK = torch.full((g.B, g.T, g.nx, g.nu), float("nan"), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        g.copy_trajectory_to_device(txs.data_ptr(), tus.data_ptr())
        g.copy_gains_to_device(tk.data_ptr(), tK.data_ptr())
        g.synchronize()
        assert np.array_equal(txs.cpu().numpy(), xs) and np.array_equal(tus.cpu().numpy(), us), name
        assert np.array_equal(tk.cpu().numpy(), k) and np.array_equal(np.swapaxes(tK.cpu().numpy(), -1, -2), K), name
        g.copy_trajectory_to_device(None, tus.data_ptr())  # NULL = skip
        g.synchronize()
        g.close()


def test_retired_and_rejected_route_bits():
    from ilqr_amd import BatchILQR, capi
    with pytest.raises(capi.ILQRError, match="retired in ABI 5"):
        BatchILQR("host", 2, 5, DT, nx=6, nu=2, u_min=[-1, -1], u_max=[1, 1], route=128)
    with pytest.raises(capi.ILQRError, match="one tile per CU"):
        BatchILQR("integrator", 64, 10, DT, u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0], route=capi.ROUTE_WIDE_TILES | capi.ROUTE_WIDE_TWO_PER_CU)
    g = BatchILQR("integrator", 64, 10, DT, u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0], route=capi.ROUTE_WIDE_TILES | capi.ROUTE_WIDE_ONE_PER_CU)
    assert g.lib.ilqr_stage_kernel_name(g.h, capi.STAGE_NAMES.index("solve")) == b"k_solve_wide2"
    g.close()
