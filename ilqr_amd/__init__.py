"""ilqr_amd -- MI355X-native batched iLQR hot path (HIP kernels behind a C ABI).

The compute lives in ilqr_amd/lib/libilqr_amd.so (sources in ilqr_amd/csrc, ABI in
include/ilqr_amd.h).  This package is the thin Python host side: a ctypes binding (capi) and
a numpy-facing BatchILQR that mirrors the reference's iLQR interface for B trajectories.
There is no CPU fallback; everything here fails loudly if the library or a GPU is missing.
"""
from .batch import BatchILQR, ALPHAS, STATUS_NAMES  # noqa: F401
from . import capi  # noqa: F401
