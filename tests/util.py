"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

TOL = 1e-6  # north_star: K/k gains and trajectory cost within 1e-6 relative, fp64


def relerr(a, b):
    """Worst per-trajectory norm-wise relative error: max_b  max|a_b - b_b| / max|b_b|."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    B = a.shape[0]
    da = np.abs(a - b).reshape(B, -1).max(axis=1)
    sb = np.abs(b).reshape(B, -1).max(axis=1)
    return float(np.max(da / np.maximum(sb, 1e-300)))


def relerr_abs(a, b, floor):
    """Like relerr but with an absolute floor on the scale (for arrays that are pure FD noise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    B = a.shape[0]
    da = np.abs(a - b).reshape(B, -1).max(axis=1)
    sb = np.abs(b).reshape(B, -1).max(axis=1)
    return float(np.max(da / np.maximum(sb, floor)))


def acrobot_x0(B, scale=1.0, seed=1234):
    """Synthetic acrobot initial conditions: (pi a, pi b, c, d) * scale, U(-1,1)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, size=(B, 4)) * np.array([np.pi, np.pi, 1.0, 1.0]) * scale


def integrator_x0(B, seed=4321):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, size=(B, 4)) * np.array([1.5, 1.5, 0.5, 0.5])


def mat(mem):
    """memory layout [..., col, row] -> matrix view [..., row, col]."""
    return np.swapaxes(mem, -1, -2)
