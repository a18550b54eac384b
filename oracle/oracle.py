"""ctypes binding of the TEST-ONLY CPU oracle (oracle/liboracle_ilqr.so) and, when it has been
built in the container, of the real-reference shim (oracle/_ref/libref_ilqr.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (ilqr_amd/) must never do so.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle_ilqr.so")
REF_SO = os.path.join(HERE, "_ref", "libref_ilqr.so")

MAXN = 32
MAXM = 32
NALPHA = 11
ALPHAS = np.array([1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316, 0.0158, 0.0079, 0.0040,
                   0.0020, 0.0010])

MODEL_ACROBOT, MODEL_DOUBLE_INTEGRATOR, MODEL_LQ = 0, 1, 2
STATUS = {0: "running", 1: "converged_grad", 2: "converged_cost", 3: "lambda_max", 4: "max_iter"}

_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)

# The oracle is ONE C source compiled in three arithmetic flavours (ilqr_oracle.h, oracle/Makefile):
#   f64: the reference's arithmetic;  f32: twin of the product's fp32 mode;  f80: x87 extended precision,
#   the yardstick of the per-knot parity metric.  `real` = per-knot data, `acc` = per-trajectory scalars.
FLAVOURS = {
    "f64": dict(so="liboracle_ilqr.so", real=(np.float64, C.c_double), acc=(np.float64, C.c_double)),
    "f32": dict(so="liboracle_ilqr_f32.so", real=(np.float32, C.c_float), acc=(np.float64, C.c_double)),
    "f80": dict(so="liboracle_ilqr_f80.so", real=(np.longdouble, C.c_longdouble), acc=(np.longdouble, C.c_longdouble)),
}
_cur = "f64"  # flavour the module-level functions work in (see `flavour()`)


class flavour:
    """`with oracle.flavour("f80"): ...` -- every call inside runs the extended-precision build and takes /
    returns numpy arrays of that flavour's types (np.longdouble)."""

    def __init__(self, name):
        assert name in FLAVOURS, name
        self.name = name

    def __enter__(self):
        global _cur
        self.prev, _cur = _cur, self.name
        return self

    def __exit__(self, *a):
        global _cur
        _cur = self.prev


def _rt():
    return FLAVOURS[_cur]["real"][0]


def _at():
    return FLAVOURS[_cur]["acc"][0]


def _rc():
    return FLAVOURS[_cur]["real"][1]


def _ac():
    return FLAVOURS[_cur]["acc"][1]


_struct_cache = {}


def _structs():
    """ctypes mirrors of orc_model / orc_traj for the current flavour."""
    if _cur in _struct_cache:
        return _struct_cache[_cur]
    rc, ac = _rc(), _ac()
    rp = C.POINTER(rc)

    class _Model(C.Structure):
        _fields_ = [
            ("id", C.c_int), ("nx", C.c_int), ("nu", C.c_int),
            ("u_min", rc * MAXM), ("u_max", rc * MAXM),
            ("dynamics", C.c_void_p), ("cost", C.c_void_p), ("final_cost", C.c_void_p),
            ("dynamics_fd", C.c_void_p), ("cost_fd", C.c_void_p), ("final_cost_fd", C.c_void_p),
            ("goal", rc * MAXN),
            ("A", rp), ("Bm", rp), ("Q", rp), ("R", rp), ("Qf", rp),
        ]

    class _Traj(C.Structure):
        _fields_ = [
            ("nx", C.c_int), ("nu", C.c_int), ("T", C.c_int), ("dt", rc),
            ("x0", rp), ("xs", rp), ("us", rp), ("fx", rp), ("fu", rp), ("cx", rp),
            ("cu", rp), ("cxx", rp), ("cxu", rp), ("cuu", rp), ("Vx", rp), ("Vxx", rp),
            ("k", rp), ("K", rp), ("dV", ac * 2), ("cost_s", ac),
            ("lambda_", ac), ("dlambda", ac), ("has_gains", C.c_int),
            ("iters", C.c_int), ("status", C.c_int), ("gnorm", ac),
            ("last_alpha_idx", C.c_int), ("n_backward", C.c_int), ("n_rollouts", C.c_int),
            ("owned", C.c_void_p),
        ]
    _struct_cache[_cur] = (_Model, _Traj)
    return _struct_cache[_cur]


def build(force=False):
    """(Re)build the oracle libraries (and _ref when /root/reference is mounted)."""
    srcs = [os.path.join(HERE, f) for f in ("ilqr_oracle.c", "ilqr_oracle.h", "orc_models.inc", "Makefile")]
    newest = max(os.path.getmtime(f) for f in srcs)
    libs = [os.path.join(HERE, f["so"]) for f in FLAVOURS.values()]
    if force or any(not os.path.exists(l) or os.path.getmtime(l) < newest for l in libs):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_libs = {}
_fixes = 0


def lib():
    if _cur not in _libs:
        build()
        L = C.CDLL(os.path.join(HERE, FLAVOURS[_cur]["so"]))
        _Model, _Traj = _structs()
        L.orc_traj_alloc.restype = C.POINTER(_Traj)
        L.orc_traj_alloc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_traj_free.argtypes = [C.POINTER(_Traj)]
        for name in ("orc_forward_pass", "orc_init_traj", "orc_gradient_norm"):
            getattr(L, name).restype = _ac()
        L.orc_quad_cost.restype = _rc()
        L.orc_set_fixes(int(_fixes))  # (a flavour loaded after set_fixes -- the fp80 yardstick, lazily -- carries the same switch)
        _libs[_cur] = L
    return _libs[_cur]


def _c(a):
    return np.ascontiguousarray(a, dtype=_rt())


def _ca(a):
    return np.ascontiguousarray(a, dtype=_at())


def _zr(shape):
    return np.zeros(shape, dtype=_rt())


def _za(shape):
    return np.zeros(shape, dtype=_at())


def _p(a):
    if a is None:
        return None
    ct = _rc() if a.dtype == _rt() else _ac()
    assert a.dtype in (_rt(), _at()), a.dtype
    return a.ctypes.data_as(C.POINTER(ct))


def _pa(a):
    """pointer to an array of per-trajectory scalars (orc_acc)"""
    if a is None:
        return None
    assert a.dtype == _at(), a.dtype
    return a.ctypes.data_as(C.POINTER(_ac()))


def set_fixes(bits=0):
    """Opt-in fixes (ILQR_FLAG_REFERENCE_FIXES on the product side): 1 = clamped rollout, 2 = Cholesky failure ends the box-QP, 3 = both."""
    global _fixes
    _fixes = int(bits)
    for name in list(_libs) or ["f64"]:
        with flavour(name):
            lib().orc_set_fixes(int(bits))


def set_params(tol_fun=1e-6, tol_grad=1e-6, lambda_factor=1.6, lambda_max=1e11, lambda_min=1e-8, z_min=0.0):
    """Solver tunables of include/ilqr.h:14-24 (process-wide in the oracle); no arguments = the reference's."""
    lib().orc_set_params(*[C.c_double(v) for v in (tol_fun, tol_grad, lambda_factor, lambda_max, lambda_min, z_min)])


def _pi(a):
    return a.ctypes.data_as(_ip) if a is not None else None


class Model:
    """A Model plugin instance (include/model.h) on the oracle side, in the flavour current at construction."""

    def __init__(self, kind, goal=None, lq=None, u_lim=None, chain=None):
        self.flavour = _cur
        _Model, _ = _structs()
        self.m = _Model()
        self._keep = []
        L = lib()
        if kind in ("acrobot", MODEL_ACROBOT):
            L.orc_model_init_acrobot(C.byref(self.m))
        elif kind in ("double_integrator", "integrator", MODEL_DOUBLE_INTEGRATOR):
            g = _c(goal if goal is not None else [1.0, 0.5, 0.0, 0.0])
            L.orc_model_init_double_integrator(C.byref(self.m), _p(g))
        elif kind in ("lq", MODEL_LQ):
            A, Bm, Q, R, Qf = [_c(a) for a in lq]
            self._keep = [A, Bm, Q, R, Qf]
            nx, nu = A.shape[0], Bm.shape[1]
            lim = 1.0 if u_lim is None else float(u_lim)
            L.orc_model_init_lq(C.byref(self.m), nx, nu, _p(A), _p(Bm), _p(Q), _p(R), _p(Qf),
                                C.c_double(-lim), C.c_double(lim))
            u_lim = None
        elif kind == "chain":  # chain = (N, params[8]): the pendulum chain of orc_models.inc / examples/user_model_pendulum_chain.hpp
            N, prm = int(chain[0]), np.ascontiguousarray(chain[1], dtype=np.float64)
            assert prm.shape == (8,) and 2 <= N <= 16 and N % 2 == 0
            lim = 1.0 if u_lim is None else float(u_lim)
            L.orc_model_init_chain(C.byref(self.m), N, prm.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(-lim), C.c_double(lim))
            u_lim = None
        else:
            raise ValueError(kind)
        self.chain_arg = chain
        self.kind, self.goal_arg, self.lq_arg, self.lim_arg = kind, goal, lq, u_lim
        if u_lim is not None:
            self.set_limits(-abs(u_lim), abs(u_lim))

    def twin(self, name):
        """The same model in another arithmetic flavour (constructed there from the same arguments)."""
        with flavour(name):
            m = Model(self.kind, goal=self.goal_arg, lq=[np.asarray(a, dtype=np.float64) for a in self.lq_arg] if self.lq_arg is not None else None,
                      u_lim=self.lim_arg, chain=self.chain_arg)
            if self.kind in ("lq", MODEL_LQ, "chain"):
                m.set_limits(np.asarray(self.u_min, dtype=np.float64), np.asarray(self.u_max, dtype=np.float64))
        return m

    def set_limits(self, lo, hi):
        for i in range(self.nu):
            self.m.u_min[i] = float(np.broadcast_to(lo, (self.nu,))[i])
            self.m.u_max[i] = float(np.broadcast_to(hi, (self.nu,))[i])

    @property
    def nx(self):
        return self.m.nx

    @property
    def nu(self):
        return self.m.nu

    @property
    def u_min(self):
        return np.array([float(v) for v in self.m.u_min[: self.nu]])

    @property
    def u_max(self):
        return np.array([float(v) for v in self.m.u_max[: self.nu]])

    @property
    def ref(self):
        assert self.flavour == _cur, "model built for flavour %s used in %s" % (self.flavour, _cur)
        return C.byref(self.m)

    def dynamics(self, x, u):
        _Model, _ = _structs()
        rp = C.POINTER(_rc())
        fn = C.CFUNCTYPE(None, C.POINTER(_Model), rp, rp, rp)(self.m.dynamics)
        x, u = _c(x), _c(u)
        dx = _zr(self.nx)
        fn(C.byref(self.m), _p(x), _p(u), _p(dx))
        return dx

    def cost(self, x, u):
        _Model, _ = _structs()
        rp = C.POINTER(_rc())
        fn = C.CFUNCTYPE(_rc(), C.POINTER(_Model), rp, rp)(self.m.cost)
        x, u = _c(x), _c(u)
        return fn(C.byref(self.m), _p(x), _p(u))

    def final_cost(self, x):
        _Model, _ = _structs()
        rp = C.POINTER(_rc())
        fn = C.CFUNCTYPE(_rc(), C.POINTER(_Model), rp)(self.m.final_cost)
        x = _c(x)
        return fn(C.byref(self.m), _p(x))

    def integrate(self, x, u, dt):
        x, u = _c(x), _c(u)
        x1 = _zr(self.nx)
        lib().orc_integrate_dynamics(self.ref, _p(x), _p(u), C.c_double(dt), _p(x1))
        return x1


# ---------------------------------------------------------------------------------------------
# box-QP family
# ---------------------------------------------------------------------------------------------
def _colmajor(Q):
    return np.ascontiguousarray(np.asarray(Q, dtype=_rt()).T).ravel()


def clamp_to_limits(x, lo, hi):
    x, lo, hi = _c(x), _c(lo), _c(hi)
    out = np.zeros_like(x)
    lib().orc_clamp_to_limits(len(x), _p(x), _p(lo), _p(hi), _p(out))
    return out


def quad_cost(Q, c, x):
    q, c, x = _colmajor(Q), _c(c), _c(x)
    return lib().orc_quad_cost(len(x), _p(q), _p(c), _p(x))


def line_search(x0, d, Q, c, lo, hi):
    x0, d, q, c, lo, hi = _c(x0), _c(d), _colmajor(Q), _c(c), _c(lo), _c(hi)
    xo = np.full(len(x0), np.nan, dtype=_rt())
    v = _rc()(np.nan)
    ns = C.c_int(0)
    failed = lib().orc_quadclamp_line_search(len(x0), _p(x0), _p(d), _p(q), _p(c), _p(lo), _p(hi),
                                             _p(xo), C.byref(v), C.byref(ns))
    return dict(failed=bool(failed), x_opt=xo, v_opt=v.value, n_steps=ns.value)


def boxqp(Q, c, x0, lo, hi):
    n = len(x0)
    q, c, x0, lo, hi = _colmajor(Q), _c(c), _c(x0), _c(lo), _c(hi)
    xo = _zr(n)
    vf = np.zeros(n, dtype=np.int32)
    R = _zr(n * n)
    nf = C.c_int(0)
    it = C.c_int(0)
    res = lib().orc_boxqp(n, _p(q), _p(c), _p(x0), _p(lo), _p(hi), _p(xo), _pi(vf), _p(R),
                          C.byref(nf), C.byref(it))
    k = nf.value
    Rm = R[: k * k].reshape(k, k).T.copy()  # column-major with ld = nfree
    return dict(result=res, x_opt=xo, v_free=vf, R_free=Rm, iters=it.value)


# ---------------------------------------------------------------------------------------------
# single-trajectory solver object (mirrors class iLQR, include/ilqr.h)
# ---------------------------------------------------------------------------------------------
class Solver:
    def __init__(self, model, T, dt):
        self.model = model
        self.T, self.dt = T, dt
        self.flavour = _cur
        self._lib = lib()
        self.s = self._lib.orc_traj_alloc(model.nx, model.nu, T, C.c_double(dt))

    def __del__(self):
        try:
            self._lib.orc_traj_free(self.s)
        except Exception:
            pass

    def _arr(self, name, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(getattr(self.s.contents, name), shape=(n,)).reshape(shape)

    # views into the C state ([T+1][n] etc.; matrices col-major flattened)
    @property
    def xs(self):
        return self._arr("xs", (self.T + 1, self.model.nx))

    @property
    def us(self):
        return self._arr("us", (self.T, self.model.nu))

    @property
    def k(self):
        return self._arr("k", (self.T, self.model.nu))

    @property
    def K(self):  # [T][nx][nu] memory == col-major (nu x nx); return as [T][nu][nx]
        n, m = self.model.nx, self.model.nu
        return self._arr("K", (self.T, n, m)).transpose(0, 2, 1)

    def mat(self, name):
        n, m, T = self.model.nx, self.model.nu, self.T
        dims = dict(fx=(n, n), fu=(n, m), cxx=(n, n), cxu=(n, m), cuu=(m, m), Vxx=(n, n))[name]
        r, c = dims
        return self._arr(name, (T + 1, c, r)).transpose(0, 2, 1)

    def vecs(self, name):
        n, m, T = self.model.nx, self.model.nu, self.T
        d = dict(cx=n, cu=m, Vx=n)[name]
        return self._arr(name, (T + 1, d))

    @property
    def dV(self):
        return np.array(self.s.contents.dV[:])

    @property
    def cost(self):
        return self.s.contents.cost_s

    @property
    def lam(self):
        return self.s.contents.lambda_

    @lam.setter
    def lam(self, v):
        self.s.contents.lambda_ = v

    @property
    def dlam(self):
        return self.s.contents.dlambda

    @property
    def iters(self):
        return self.s.contents.iters

    @property
    def status(self):
        return self.s.contents.status

    @property
    def gnorm(self):
        return self.s.contents.gnorm

    def init_traj(self, x0, u0):
        x0, u0 = _c(x0), _c(u0)
        return self._lib.orc_init_traj(self.model.ref, self.s, _p(x0), _p(u0))

    def compute_derivatives(self):
        self._lib.orc_compute_derivatives(self.model.ref, self.s)

    def backward_pass(self):
        return self._lib.orc_backward_pass(self.model.ref, self.s)

    def gradient_norm(self):
        return self._lib.orc_gradient_norm(self.s)

    def line_search(self):
        nc, dc, ex = _ac()(0), _ac()(0), _ac()(0)
        a = self._lib.orc_line_search(self.model.ref, self.s, C.byref(nc), C.byref(dc), C.byref(ex))
        return a, nc.value, dc.value, ex.value

    def generate_trajectory(self, x0=None, u0=None, max_iters=0, fixed_work=False, log=False):
        if x0 is not None:
            self.init_traj(x0, u0)
        cl = _za(100) if log else None
        st = self._lib.orc_generate_trajectory(self.model.ref, self.s, max_iters, int(fixed_work), _pa(cl))
        return (st, cl[: self.iters]) if log else st


# ---------------------------------------------------------------------------------------------
# batched stages in the canonical layouts of include/ilqr_amd.h
# ---------------------------------------------------------------------------------------------
def batch_solve(model, x0, u0, dt, max_iters=0, fixed_work=False, nthreads=0):
    x0, u0 = _c(x0), _c(u0)
    B, T = u0.shape[0], u0.shape[1]
    n, m = model.nx, model.nu
    out = dict(xs=_zr((B, T + 1, n)), us=_zr((B, T, m)), k=_zr((B, T, m)),
               K=_zr((B, T, n, m)), cost=_za(B), iters=np.zeros(B, dtype=np.int32),
               status=np.zeros(B, dtype=np.int32), lam=_za(B))
    lib().orc_batch_solve(model.ref, B, T, C.c_double(dt), _p(x0), _p(u0), max_iters,
                          int(fixed_work), nthreads, _p(out["xs"]), _p(out["us"]), _p(out["k"]),
                          _p(out["K"]), _pa(out["cost"]), _pi(out["iters"]), _pi(out["status"]),
                          _pa(out["lam"]))
    out["K"] = out["K"].transpose(0, 1, 3, 2)  # -> [B][T][nu][nx] view
    return out


def batch_iterate_from(model, x0, xs, us, k, K, cost, lam, dlam, dt, n_iters=1, fixed_work=False, nthreads=0):
    """n_iters outer iterations from the given state (K as [B][T][nu][nx]); returns the state after."""
    x0, xs, us, k, cost = _c(x0), _c(xs), _c(us), _c(k), _ca(cost)
    B, T = us.shape[0], us.shape[1]
    n, m = model.nx, model.nu
    Kc = _c(np.asarray(K).transpose(0, 1, 3, 2))
    lam = _ca(np.broadcast_to(lam, (B,)))
    dlam = _ca(np.broadcast_to(dlam, (B,)))
    out = dict(xs=_zr((B, T + 1, n)), us=_zr((B, T, m)), k=_zr((B, T, m)),
               K=_zr((B, T, n, m)), cost=_za(B), iters=np.zeros(B, dtype=np.int32),
               status=np.zeros(B, dtype=np.int32), lam=_za(B), dlam=_za(B),
               alpha=np.zeros(B, dtype=np.int32), gnorm=_za(B), dV=_za((B, 2)))
    lib().orc_batch_iterate_from(model.ref, B, T, C.c_double(dt), _p(x0), _p(xs), _p(us), _p(k), _p(Kc),
                                 _pa(cost), _pa(lam), _pa(dlam), int(n_iters), int(fixed_work), nthreads,
                                 _p(out["xs"]), _p(out["us"]), _p(out["k"]), _p(out["K"]), _pa(out["cost"]),
                                 _pi(out["iters"]), _pi(out["status"]), _pa(out["lam"]), _pa(out["dlam"]),
                                 _pi(out["alpha"]), _pa(out["gnorm"]), _pa(out["dV"]))
    out["K"] = out["K"].transpose(0, 1, 3, 2)
    return out


def batch_rollout(model, x0, u, dt, xs_nom=None, K=None, nthreads=0):
    """K given as [B][T][nu][nx]."""
    x0, u = _c(x0), _c(u)
    B, T = u.shape[0], u.shape[1]
    n, m = model.nx, model.nu
    xs = _zr((B, T + 1, n))
    us = _zr((B, T, m))
    cost = _za(B)
    Kc = _c(np.asarray(K).transpose(0, 1, 3, 2)) if K is not None else None
    xn = _c(xs_nom) if xs_nom is not None else None
    lib().orc_batch_rollout(model.ref, B, T, C.c_double(dt), _p(x0), _p(u), _p(xn), _p(Kc),
                            nthreads, _p(xs), _p(us), _pa(cost))
    return xs, us, cost


DERIV_NAMES = ("fx", "fu", "cx", "cu", "cxx", "cxu", "cuu")


def deriv_shapes(n, m):
    """Canonical (memory) trailing shapes; matrices are column-major, i.e. stored [col][row]."""
    return dict(fx=(n, n), fu=(m, n), cx=(n,), cu=(m,), cxx=(n, n), cxu=(m, n), cuu=(m, m))


def batch_derivatives(model, xs, us, dt, nthreads=0):
    """Returns dict of arrays in MEMORY layout [B][T+1][col][row] (column-major matrices)."""
    xs, us = _c(xs), _c(us)
    B, T = us.shape[0], us.shape[1]
    n, m = model.nx, model.nu
    sh = deriv_shapes(n, m)
    out = {k: _zr((B, T + 1) + sh[k]) for k in DERIV_NAMES}
    lib().orc_batch_derivatives(model.ref, B, T, C.c_double(dt), _p(xs), _p(us), nthreads,
                                *[_p(out[k]) for k in DERIV_NAMES])
    return out


def batch_backward(model, us, derivs, k_prev=None, lam=None, nthreads=0):
    """derivs in memory layout (see batch_derivatives). Returns k [B][T][m], K memory layout
    [B][T][nx][nu] (col-major nu x nx), dV [B][2], diverge [B], Vx0, Vxx0."""
    us = _c(us)
    B, T = us.shape[0], us.shape[1]
    n, m = model.nx, model.nu
    d = {k: _c(derivs[k]) for k in DERIV_NAMES}
    kp = _c(k_prev) if k_prev is not None else None
    lm = _ca(np.broadcast_to(1.0 if lam is None else lam, (B,)))
    k = _zr((B, T, m))
    K = _zr((B, T, n, m))
    dV = _za((B, 2))
    div = np.zeros(B, dtype=np.int32)
    Vx0 = _zr((B, n))
    Vxx0 = _zr((B, n, n))
    lib().orc_batch_backward(model.ref, B, T, _p(us), *[_p(d[kk]) for kk in DERIV_NAMES], _p(kp),
                             _pa(lm), nthreads, _p(k), _p(K), _pa(dV), _pi(div), _p(Vx0), _p(Vxx0))
    return dict(k=k, K=K, dV=dV, diverge=div, Vx0=Vx0, Vxx0=Vxx0)


# ---------------------------------------------------------------------------------------------
# the real reference (container only)
# ---------------------------------------------------------------------------------------------
_ref = None


def ref_available():
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/src"):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.ref_quad_cost.restype = C.c_double
        _ref.ref_fd_scalar_negquad.restype = C.c_double
        _ref.ref_fd_scalar_negquad.argtypes = [C.c_double]
    return _ref


def ref_boxqp(Q, c, x0, lo, hi):
    n = len(x0)
    q, c, x0, lo, hi = _colmajor(Q), _c(c), _c(x0), _c(lo), _c(hi)
    xo = np.zeros(n)
    vf = np.zeros(n, dtype=np.int32)
    R = np.zeros(n * n)
    nf = C.c_int(0)
    res = ref().ref_boxqp(n, _p(q), _p(c), _p(x0), _p(lo), _p(hi), _p(xo), _pi(vf), _p(R), C.byref(nf))
    k = nf.value
    return dict(result=res, x_opt=xo, v_free=vf, R_free=R[: k * k].reshape(k, k).T.copy())


def ref_line_search(x0, d, Q, c, lo, hi):
    x0, d, q, c, lo, hi = _c(x0), _c(d), _colmajor(Q), _c(c), _c(lo), _c(hi)
    xo = np.full(len(x0), np.nan)
    v = C.c_double(np.nan)
    ns = C.c_int(0)
    failed = ref().ref_line_search(len(x0), _p(x0), _p(d), _p(q), _p(c), _p(lo), _p(hi), _p(xo),
                                   C.byref(v), C.byref(ns))
    return dict(failed=bool(failed), x_opt=xo, v_opt=v.value, n_steps=ns.value)


def ref_quad_cost(Q, c, x):
    q, c, x = _colmajor(Q), _c(c), _c(x)
    return ref().ref_quad_cost(len(x), _p(q), _p(c), _p(x))


def ref_clamp(x, lo, hi):
    x, lo, hi = _c(x), _c(lo), _c(hi)
    out = np.zeros_like(x)
    ref().ref_clamp(len(x), _p(x), _p(lo), _p(hi), _p(out))
    return out


def ref_model_eval(mid, goal, x, u, dt):
    nx, nu = 4, (1 if mid == 0 else 2)
    g = _c(goal if goal is not None else np.zeros(4))
    x, u = _c(x), _c(u)
    dx, x1 = np.zeros(nx), np.zeros(nx)
    c, f = C.c_double(0), C.c_double(0)
    ref().ref_model_eval(mid, _p(g), _p(x), _p(u), C.c_double(dt), _p(dx), _p(x1), C.byref(c), C.byref(f))
    return dx, x1, c.value, f.value


def ref_fd_knot(mid, goal, x, u, dt, is_final):
    n, m = 4, (1 if mid == 0 else 2)
    g = _c(goal if goal is not None else np.zeros(4))
    x, u = _c(x), _c(u)
    o = dict(fx=np.zeros((n, n)), fu=np.zeros((m, n)), cx=np.zeros(n), cu=np.zeros(m),
             cxx=np.zeros((n, n)), cuu=np.zeros((m, m)))
    ref().ref_fd_knot(mid, _p(g), _p(x), _p(u), C.c_double(dt), int(is_final), _p(o["fx"]),
                      _p(o["fu"]), _p(o["cx"]), _p(o["cu"]), _p(o["cxx"]), _p(o["cuu"]))
    return o  # memory layout [col][row]
