"""BatchILQR -- numpy-facing host mirror of the reference's iLQR class for B trajectories.

Method names follow include/ilqr.h:28-107 (init_traj, generate_trajectory, its warm-start
overload) plus the stage calls and accessors the reference keeps private (tests reach them
through FRIEND_TEST there).  All arrays use the canonical layouts of include/ilqr_amd.h;
matrices are returned as [..., rows, cols] numpy arrays (transposed views of the column-major
memory), so K[b, t] is the nu x nx gain matrix and fx[b, t] the nx x nx Jacobian.
"""
import ctypes as C

import numpy as np

from . import capi

ALPHAS = np.array([1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316, 0.0158, 0.0079, 0.0040,
                   0.0020, 0.0010])  # include/ilqr.h:24
STATUS_NAMES = {0: "running", 1: "converged_grad", 2: "converged_cost", 3: "lambda_max", 4: "max_iter"}
_MODELS = {"acrobot": (capi.MODEL_ACROBOT, 4, 1), "double_integrator": (capi.MODEL_DOUBLE_INTEGRATOR, 4, 2),
           "integrator": (capi.MODEL_DOUBLE_INTEGRATOR, 4, 2),
           # host-evaluated model: only the backward pass runs on the device (nx, nu given by the caller)
           "host": (capi.MODEL_HOST, None, None),
           # synthetic LQ model (BASELINE.json configs[4]): lq=(A, B, Q, R, Qf), row-major; nx, nu from their shapes
           "lq": (capi.MODEL_LQ, None, None),
           # the caller's own device twin, compiled into the library given by lib= (ilqr_amd._build.build_user); nx, nu by the caller
           "user": (capi.MODEL_USER, None, None)}
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class BatchILQR:
    def __init__(self, model, B, T, dt, u_min=None, u_max=None, goal=None, device=0, flags=0,
                 stream=None, params=None, nx=None, nu=None, lq=None, dtype="f64", lib=None, user_params=None, route=0, assume_cus=0):
        self.lib = capi.load(path=lib)
        self._check = lambda rc: capi.check(rc, self.lib)
        self._ctor = dict(model=model, B=B, T=T, dt=dt, u_min=u_min, u_max=u_max, goal=goal, device=device, flags=flags,
                          stream=stream, params=params, nx=nx, nu=nu, lq=lq, dtype=dtype, lib=lib, user_params=user_params,
                          route=route, assume_cus=assume_cus)
        mid, mnx, mnu = _MODELS[model]
        if lq is not None:
            lq = [_c(a) for a in lq]
            nx, nu = lq[1].shape
        nx, nu = (mnx or nx), (mnu or nu)
        self.model, self.nx, self.nu, self.B, self.T, self.dt = model, nx, nu, int(B), int(T), float(dt)
        self._keep = []
        d = capi.Desc()
        d.abi_version = capi.ABI_VERSION
        d.model, d.nx, d.nu, d.T, d.B, d.dt = mid, nx, nu, self.T, self.B, self.dt
        d.device, d.flags = device, flags
        d.route, d.assume_cus = int(route), int(assume_cus)  # enum ilqr_route: equivalent kernels (A/B runs, bit-identity tests)
        d.dtype = {"f64": capi.DTYPE_F64, "f32": capi.DTYPE_F32}[dtype]
        self.dtype = dtype
        for name, val, n in (("u_min", u_min, nu), ("u_max", u_max, nu), ("goal", goal, nx)):
            if val is not None:
                arr = _c(np.broadcast_to(np.asarray(val, dtype=np.float64), (n,)))
                self._keep.append(arr)
                setattr(d, name, _p(arr))
        if lq is not None:
            assert lq[0].shape == (nx, nx) and lq[2].shape == (nx, nx) and lq[3].shape == (nu, nu) and lq[4].shape == (nx, nx)
            self._keep.extend(lq)
            d.lq_A, d.lq_B, d.lq_Q, d.lq_R, d.lq_Qf = (_p(a) for a in lq)
        d.stream = stream
        if user_params is not None:
            up = _c(user_params).ravel()
            self._keep.append(up)
            d.user_params, d.n_user_params = _p(up), len(up)
        if params is not None:
            p = capi.Params()
            self.lib.ilqr_default_params(C.byref(p))
            for k, v in params.items():
                setattr(p, k, v)
            self._keep.append(p)
            d.params = C.pointer(p)
        self.h = C.c_void_p()
        self._check(self.lib.ilqr_create(C.byref(d), C.byref(self.h)))

    def clone(self):
        """A second, independent handle for the same problem description (own device memory and state)."""
        return BatchILQR(**self._ctor)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ilqr_synchronize(self.h)
            for a in getattr(self, "_pinned", []):
                self.lib.ilqr_host_unregister(a.ctypes.data)
            self._pinned = []
            self.lib.ilqr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- solves (include/ilqr.h:49-54) ----
    def init_traj(self, x0, u0):
        x0, u0 = _c(x0), _c(u0)
        assert x0.shape == (self.B, self.nx) and u0.shape == (self.B, self.T, self.nu)
        cost = np.zeros(self.B)
        self._check(self.lib.ilqr_init_traj(self.h, _p(x0), _p(u0), _p(cost)))
        return cost

    def generate_trajectory(self, x0=None, u0=None):
        if x0 is not None and u0 is not None:
            x0, u0 = _c(x0), _c(u0)
            self._check(self.lib.ilqr_solve(self.h, _p(x0), _p(u0)))
        elif x0 is not None:
            x0 = _c(x0)
            self._check(self.lib.ilqr_warm_start(self.h, _p(x0)))
        else:
            self._check(self.lib.ilqr_generate_trajectory(self.h))

    solve = generate_trajectory  # BASELINE.json's "iLQR::solve()"

    def iterate(self, n=1):
        self._check(self.lib.ilqr_iterate(self.h, int(n)))

    def synchronize(self):
        self._check(self.lib.ilqr_synchronize(self.h))

    # ---- stages ----
    def compute_derivatives(self):
        self._check(self.lib.ilqr_compute_derivatives(self.h))

    def backward_pass(self):
        div = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.ilqr_backward_pass(self.h, div.ctypes.data_as(_ip)))
        return div

    def backward_step(self):
        self._check(self.lib.ilqr_backward_step(self.h))

    def rollout_candidates(self):
        cost = np.zeros((self.B, len(ALPHAS)))
        self._check(self.lib.ilqr_rollout_candidates(self.h, _p(cost)))
        return cost

    def line_search(self):
        self._check(self.lib.ilqr_line_search(self.h))

    def reset_state(self, warm=False):
        """warm=False: the non-rollout part of init_traj; warm=True: a new outer loop on the stored
        solution (status / iteration count / flgChange restart, lambda and gains persist)."""
        self._check(self.lib.ilqr_reset_state(self.h, int(bool(warm))))

    # ---- setters (canonical layouts; matrices given as [..., rows, cols]) ----
    def set_trajectory(self, x0=None, xs=None, us=None, cost=None):
        a = [None if v is None else _c(v) for v in (x0, xs, us, cost)]
        self._check(self.lib.ilqr_set_trajectory(self.h, *[_p(v) for v in a]))

    def set_gains(self, k=None, K=None):
        k = None if k is None else _c(k)
        K = None if K is None else _c(np.swapaxes(np.asarray(K), -1, -2))
        self._check(self.lib.ilqr_set_gains(self.h, _p(k), _p(K)))

    def set_derivatives(self, **d):
        """fx, fu, cx, cu, cxx, cxu, cuu as [B][T+1][rows][cols] (vectors [B][T+1][n])."""
        order = ("fx", "fu", "cx", "cu", "cxx", "cxu", "cuu")
        arrs = []
        for name in order:
            v = d.get(name)
            if v is None:
                arrs.append(None)
            elif name in ("cx", "cu"):
                arrs.append(_c(v))
            else:
                arrs.append(_c(np.swapaxes(np.asarray(v), -1, -2)))
        self._check(self.lib.ilqr_set_derivatives(self.h, *[_p(v) for v in arrs]))

    def set_lambda(self, lam=None, dlam=None):
        lam = None if lam is None else _c(np.broadcast_to(lam, (self.B,)))
        dlam = None if dlam is None else _c(np.broadcast_to(dlam, (self.B,)))
        self._check(self.lib.ilqr_set_lambda(self.h, _p(lam), _p(dlam)))

    # ---- getters ----
    def trajectory(self, out=None):
        """xs [B][T+1][nx], us [B][T][nu].  `out=(xs, us)`: C-contiguous float64 arrays to fill -- a
        caller that keeps its result arrays across solves avoids the first-touch page faults of fresh
        ones, which cost several times the PCIe transfer itself (PCIe-inclusive)."""
        if out is None:
            xs = np.zeros((self.B, self.T + 1, self.nx))
            us = np.zeros((self.B, self.T, self.nu))
        else:
            xs, us = out
            for a, shape in ((xs, (self.B, self.T + 1, self.nx)), (us, (self.B, self.T, self.nu))):
                if a.shape != shape or a.dtype != np.float64 or not a.flags.c_contiguous:
                    raise ValueError("out arrays must be C-contiguous float64 of shape %s" % (shape,))
        self._check(self.lib.ilqr_get_trajectory(self.h, _p(xs), _p(us)))
        return xs, us

    def gains(self):
        k = np.zeros((self.B, self.T, self.nu))
        K = np.zeros((self.B, self.T, self.nx, self.nu))  # memory: column-major nu x nx
        self._check(self.lib.ilqr_get_gains(self.h, _p(k), _p(K)))
        return k, np.swapaxes(K, -1, -2)

    def result_buffers(self, pinned=True, K=True):
        """Result arrays a caller keeps across solves, in the ABI's memory layouts: dict(xs, us, k, K, cost) (K [B][T][nx][nu] in memory =
        column-major nu x nx per knot; K=False: without the gains).  pinned: page-locked with ilqr_host_register, so that results_async's
        copies are DMA transfers that the call does not wait for."""
        bufs = dict(xs=np.zeros((self.B, self.T + 1, self.nx)), us=np.zeros((self.B, self.T, self.nu)), cost=np.zeros(self.B))
        if K:
            bufs.update(k=np.zeros((self.B, self.T, self.nu)), K=np.zeros((self.B, self.T, self.nx, self.nu)))
        if pinned:
            for a in bufs.values():
                self._check(self.lib.ilqr_host_register(a.ctypes.data, a.nbytes))
            self._pinned = getattr(self, "_pinned", []) + [a for a in bufs.values()]
        return bufs

    def results_async(self, bufs):
        """ilqr_get_results_async into `bufs` (result_buffers()): every array in one call, nothing waited for -- valid after synchronize()."""
        self._check(self.lib.ilqr_get_results_async(self.h, _p(bufs.get("xs")), _p(bufs.get("us")), _p(bufs.get("k")), _p(bufs.get("K")), _p(bufs.get("cost"))))

    def copy_trajectory_to_device(self, xs_ptr=None, us_ptr=None):
        """canonical xs [B][T+1][nx] / us [B][T][nu] (double) into caller-owned device memory (raw pointers, e.g. torch.Tensor.data_ptr());
        enqueued on the handle's stream."""
        self._check(self.lib.ilqr_copy_trajectory_to_device(self.h, xs_ptr, us_ptr))

    def copy_gains_to_device(self, k_ptr=None, K_ptr=None):
        self._check(self.lib.ilqr_copy_gains_to_device(self.h, k_ptr, K_ptr))

    def derivatives(self):
        n, m, B, T1 = self.nx, self.nu, self.B, self.T + 1
        mem = dict(fx=np.zeros((B, T1, n, n)), fu=np.zeros((B, T1, m, n)), cx=np.zeros((B, T1, n)),
                   cu=np.zeros((B, T1, m)), cxx=np.zeros((B, T1, n, n)), cxu=np.zeros((B, T1, m, n)),
                   cuu=np.zeros((B, T1, m, m)))
        order = ("fx", "fu", "cx", "cu", "cxx", "cxu", "cuu")
        self._check(self.lib.ilqr_get_derivatives(self.h, *[_p(mem[k]) for k in order]))
        return {k: (v if k in ("cx", "cu") else np.swapaxes(v, -1, -2)) for k, v in mem.items()}

    def cost(self):
        c = np.zeros(self.B)
        self._check(self.lib.ilqr_get_cost(self.h, _p(c)))
        return c

    def lambdas(self):
        lam, dlam = np.zeros(self.B), np.zeros(self.B)
        self._check(self.lib.ilqr_get_lambda(self.h, _p(lam), _p(dlam)))
        return lam, dlam

    def dV(self):
        d = np.zeros((self.B, 2))
        self._check(self.lib.ilqr_get_dV(self.h, _p(d)))
        return d

    def gnorm(self):
        g = np.zeros(self.B)
        self._check(self.lib.ilqr_get_gnorm(self.h, _p(g)))
        return g

    def status(self):
        st, it, al = (np.zeros(self.B, dtype=np.int32) for _ in range(3))
        self._check(self.lib.ilqr_get_status(self.h, st.ctypes.data_as(_ip), it.ctypes.data_as(_ip),
                                            al.ctypes.data_as(_ip)))
        return st, it, al

    def candidate(self, a):
        xs = np.zeros((self.B, self.T + 1, self.nx))
        us = np.zeros((self.B, self.T, self.nu))
        self._check(self.lib.ilqr_get_candidate(self.h, int(a), _p(xs), _p(us)))
        return xs, us

    def count_running(self):
        n = C.c_int(0)
        self._check(self.lib.ilqr_count_running(self.h, C.byref(n)))
        return n.value

    # ---- measurement ----
    def profile(self, enable=True):
        self._check(self.lib.ilqr_profile_enable(self.h, int(enable)))

    def profile_reset(self):
        self._check(self.lib.ilqr_profile_reset(self.h))

    def profile_read(self):
        ms = (C.c_double * capi.NUM_STAGES)()
        n = (C.c_int * capi.NUM_STAGES)()
        self._check(self.lib.ilqr_profile_read(self.h, ms, n))
        return {capi.STAGE_NAMES[i]: (ms[i], n[i]) for i in range(capi.NUM_STAGES)}

    def shader_clock_mhz(self):
        """clock of the CUs during the persistent kernel's launches since the last profile_reset (ilqr_profile_shader_clock)"""
        mhz = C.c_double(0)
        self._check(self.lib.ilqr_profile_shader_clock(self.h, C.byref(mhz)))
        return mhz.value
