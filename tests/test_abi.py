"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and
exports every symbol include/ilqr_amd.h declares; without a GPU it fails loudly (no fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ilqr_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ilqr_[a-z_A-Z0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from ilqr_amd import capi
    lib = capi.load()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libilqr_amd.so does not export %s" % n
    # the ctypes table covers exactly the header
    assert sorted(capi.SYMBOLS) == names


def test_abi_version_and_defaults():
    from ilqr_amd import capi
    lib = capi.load()
    assert lib.ilqr_abi_version() == capi.ABI_VERSION
    p = capi.Params()
    lib.ilqr_default_params(C.byref(p))
    # include/ilqr.h:14-24
    assert (p.max_iter, p.tol_fun, p.tol_grad, p.lambda_init, p.dlambda_init) == (100, 1e-6, 1e-6, 1.0, 1.0)
    assert (p.lambda_factor, p.lambda_max, p.lambda_min, p.z_min) == (1.6, 1e11, 1e-8, 0.0)


def test_route_and_flag_constants_match_the_header():
    """ilqr_amd/capi.py mirrors enum ilqr_route / ilqr_flags by value: a constant that drifts selects another kernel silently."""
    from ilqr_amd import capi
    src = open(os.path.join(ROOT, "include", "ilqr_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    enums = {k: int(v, 0) for k, v in re.findall(r"\b(ILQR_(?:ROUTE|FLAG)_[A-Z0-9_]+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", src)}
    routes = {k: v for k, v in enums.items() if k.startswith("ILQR_ROUTE_")}
    assert len(routes) >= 12
    for k, v in routes.items():
        if k == "ILQR_ROUTE_AUTO":
            continue
        assert getattr(capi, k[len("ILQR_"):]) == v, k
    for k, v in enums.items():
        if k.startswith("ILQR_FLAG_") and hasattr(capi, k[len("ILQR_"):]):
            assert getattr(capi, k[len("ILQR_"):]) == v, k


def test_bad_arguments_are_rejected():
    from ilqr_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    d = capi.Desc()
    d.abi_version = 999
    assert lib.ilqr_create(C.byref(d), C.byref(h)) == -1
    assert b"ABI version" in lib.ilqr_last_error()
    d.abi_version = capi.ABI_VERSION
    d.model, d.nx, d.nu, d.T, d.B, d.dt = 0, 4, 1, 0, 4, 0.02
    assert lib.ilqr_create(C.byref(d), C.byref(h)) == -1  # T must be positive
    assert lib.ilqr_iterate(None, 1) == -1


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ilqr_amd import BatchILQR, capi
    with pytest.raises(capi.ILQRError, match="no HIP device"):
        BatchILQR("acrobot", 4, 10, 0.02)


def test_product_never_touches_the_oracle():
    """The product package must not import/link/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "ilqr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower().replace("no cpu", ""), os.path.join(dirpath, f)


def test_user_model_build_exports_the_same_abi():
    """A build with a user device model (-DILQR_USER_MODEL_HEADER, examples/user_model_acrobot.hpp) is the same library:
    every symbol of the header, and ilqr_has_user_model() says which build is loaded (no device needed for that)."""
    from ilqr_amd import _build, capi
    if not os.path.exists(_build.HIPCC):
        pytest.skip("hipcc not available")
    path = _build.build_user(_build.USER_EXAMPLE_HEADER, _build.USER_EXAMPLE_LIB)
    user = capi.load(path=path)
    for n in _declared_symbols():
        assert hasattr(user, n), n
    assert user.ilqr_has_user_model() == 1 and capi.load().ilqr_has_user_model() == 0
    assert user.ilqr_abi_version() == capi.ABI_VERSION
