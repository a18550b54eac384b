"""Debug helper (GPU box): locate the first backward-pass mismatch vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util import integrator_x0, acrobot_x0, mat
from tests.test_gpu_parity import make, u_init, DT
np.set_printoptions(precision=17, linewidth=200)
name, B, T, lim, lam = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
om, g, x0 = make(O, name, B, T, lim)
u0 = u_init(B, T, om.nu, scale=1.0)
xs_o, us_o, cost_o = O.batch_rollout(om, x0, u0, DT)
do = O.batch_derivatives(om, xs_o, us_o, DT)
k_prev = u_init(B, T, om.nu, seed=11, scale=0.2)
ro = O.batch_backward(om, us_o, do, k_prev=k_prev, lam=lam)
g.set_trajectory(x0=x0, xs=xs_o, us=us_o, cost=cost_o)
g.set_derivatives(**{k: (do[k] if k in ("cx", "cu") else mat(do[k])) for k in do})
g.set_gains(k=k_prev, K=np.zeros((B, T, om.nu, om.nx)))
g.set_lambda(lam, 1.0)
div = g.backward_pass()
k, K = g.gains()
err = np.abs(k - ro["k"]).reshape(B, T, -1).max(axis=2)
bad = np.argwhere(err > 1e-9 * np.abs(ro["k"]).max())
print("n mismatching (b,t):", len(bad), "trajectories:", sorted(set(bad[:, 0])))
for b in sorted(set(bad[:, 0]))[:3]:
    ts = bad[bad[:, 0] == b][:, 1]
    t = ts.max()
    print("b", b, "first mismatch at t =", t, "of", len(ts))
    print(" gpu k", k[b, t], "oracle k", ro["k"][b, t], "us", us_o[b, t], "lo", om.u_min - us_o[b, t], "hi", om.u_max - us_o[b, t])
    print(" gpu k[t+1]", k[b, min(t + 1, T - 1)], "oracle", ro["k"][b, min(t + 1, T - 1)])
    # rebuild the QP of that step from the oracle's own Vx/Vxx
    s = O.Solver(om, T, DT)
    s.xs[:] = xs_o[b]; s.us[:] = us_o[b]
    for nm in ("fx", "fu", "cxx", "cxu", "cuu"):
        s.mat(nm)[:] = mat(do[nm][b])
    for nm in ("cx", "cu"):
        s.vecs(nm)[:] = do[nm][b]
    s.k[:] = k_prev[b]; s.lam = lam
    s.backward_pass()
    Vx1, Vxx1 = s.vecs("Vx")[t + 1], s.mat("Vxx")[t + 1]
    fx, fu = s.mat("fx")[t], s.mat("fu")[t]
    Qu = s.vecs("cu")[t] + fu.T @ Vx1
    Quu = s.mat("cuu")[t] + fu.T @ Vxx1 @ fu
    QuuF = Quu + lam * np.eye(om.nu)
    kw = ro["k"][b, min(t + 1, T - 1)] if t < T - 1 else k_prev[b, T - 1]
    print(" QuuF", QuuF.tolist(), "Qu", Qu.tolist(), "x0", kw.tolist())
    r = O.boxqp(QuuF, Qu, kw, om.u_min - us_o[b, t], om.u_max - us_o[b, t])
    print(" oracle boxqp:", r)
    Kerr = np.abs(K[b] - mat(ro["K"])[b]).reshape(T, -1).max(axis=1)
    print(" K mismatch steps:", np.flatnonzero(Kerr > 1e-9 * np.abs(ro["K"][b]).max()))
    for tt in (t + 1, t):
        if tt < T:
            print(" t", tt, "gpu K", K[b, tt].tolist(), "\n      oracle K", mat(ro["K"])[b, tt].tolist(), "\n      k", k[b, tt], ro["k"][b, tt], "lo/hi", om.u_min - us_o[b, tt], om.u_max - us_o[b, tt])
