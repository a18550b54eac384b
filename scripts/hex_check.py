"""k_backward_h (16 lanes per trajectory) against k_backward_q (4 lanes) on the headline workload: agreement and time.
    python scripts/hex_check.py [B] [iters_before] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
T, DT = 499, 0.02
rng = np.random.default_rng(0)
x0 = rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]) * 0.5
u0 = np.zeros((B, T, 1))
outs = {}
for label, fl in (("quad", capi.FLAG_UNFUSED), ("hex", capi.FLAG_UNFUSED | capi.FLAG_BACKWARD_LANE_GROUP)):
    g = BatchILQR("acrobot", B, T, DT, flags=fl | capi.FLAG_FIXED_WORK, u_min=-1.5, u_max=1.5, dtype=dtype, params=dict(max_iter=1000))
    g.init_traj(x0, u0)
    if label == "quad":
        g.iterate(warm)
        state = (g.trajectory(), g.gains(), g.lambdas())
    else:  # same state as the quad handle reached
        (xs, us), (k, K), (lam, dlam) = state
        g.set_trajectory(x0=x0, xs=xs, us=us, cost=outs["quad"]["cost"])
        g.set_gains(k=k, K=K)
        g.set_lambda(lam, dlam)
    if label == "quad":
        outs[label] = dict(cost=g.cost())
    g.compute_derivatives()
    g.profile(True)
    g.profile_reset()
    for _ in range(5):
        div = g.backward_pass()
    p = g.profile_read()
    k, K = g.gains()
    outs[label] = dict(outs.get(label, {}), k=k, K=K, dV=g.dV(), div=np.asarray(div), gnorm=g.gnorm(), ms=p["backward"][0] / p["backward"][1])
    print("%-5s backward %.4f ms per launch" % (label, outs[label]["ms"]))
    g.close()
q, h = outs["quad"], outs["hex"]
print("diverge equal:", np.array_equal(q["div"], h["div"]))
for key in ("k", "K", "dV", "gnorm"):
    a, b_ = q[key], h[key]
    den = np.maximum(np.abs(a).reshape(B, -1).max(axis=1), 1e-300)
    err = np.abs(a - b_).reshape(B, -1).max(axis=1) / den
    print("%-6s per-trajectory max relative difference: median %.2e  95%% %.2e  max %.2e" % (key, np.median(err), np.quantile(err, 0.95), err.max()))
