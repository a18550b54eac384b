"""Timing of the generic wave-per-trajectory backward kernel at n=32, m=16, T=200 (config 5 shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR
from oracle import oracle as O
from tests.test_gpu_generic_backward import lq_model
from tests.util import mat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
T, DT = 200, 0.02
om = lq_model(O, n, m, lim=0.2)
rng = np.random.default_rng(0)
nb = 4
x0 = rng.uniform(-1, 1, (nb, n)); u0 = rng.normal(size=(nb, T, m)) * 0.3
t0 = time.time(); xs, us, cost = O.batch_rollout(om, x0, u0, DT); dv = O.batch_derivatives(om, xs, us, DT); print("oracle derivs for %d trajectories: %.1f s" % (nb, time.time() - t0))
t0 = time.time(); ro = O.batch_backward(om, us, dv, lam=1.0, nthreads=1); tcpu = (time.time() - t0) / nb; print("oracle backward: %.3f s per trajectory (1 core) -> %.3e timesteps/s/core" % (tcpu, T / tcpu))
rep = B // nb
tile = lambda a: np.tile(a, (rep,) + (1,) * (a.ndim - 1))
g = BatchILQR("host", B, T, DT, nx=n, nu=m, u_min=om.u_min, u_max=om.u_max)
g.set_trajectory(x0=tile(x0), xs=tile(xs), us=tile(us), cost=tile(cost))
g.set_derivatives(**{k: tile(dv[k] if k in ("cx", "cu") else mat(dv[k])) for k in dv})
g.set_lambda(1.0, 1.0)
g.profile(True)
for rep_i in range(2):
    g.profile_reset(); g.backward_step(); p = g.profile_read()["backward"]
    print("B=%d: backward %.2f ms -> %.3e trajectory-timesteps/s; n=%d m=%d" % (B, p[0], B * T / (p[0] * 1e-3), n, m))
k, K = g.gains()
print("max |k - oracle| rel", np.abs(k[:nb] - ro["k"]).max() / np.abs(ro["k"]).max())
