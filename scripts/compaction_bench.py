"""Full solves (every trajectory leaves its own loop) of big batches, with and without the compaction of running
trajectories between chunks (ilqr_generate_trajectory): wall time and identity of the results."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0, integrator_x0

def run(name, B, T, x0, kw, nu):
    res = {}
    for label, route in (("compaction", 0), ("no compaction", capi.ROUTE_NO_COMPACTION)):
        g = BatchILQR(name, B, T, 0.02, route=route, **kw)
        u0 = np.zeros((B, T, nu))
        g.generate_trajectory(x0, u0)  # warm-up (code load, allocations)
        g.init_traj(x0, u0)
        g.synchronize()
        t0 = time.perf_counter()
        g.generate_trajectory()
        g.synchronize()
        el = time.perf_counter() - t0
        st, it, al = g.status()
        res[label] = (el, g.cost(), st, it)
        print("%s B=%d T=%d %-14s %.1f ms  iterations: mean %.1f, median %d, max %d; trajectory-iterations %.3g" % (name, B, T, label, el * 1e3, it.mean(), np.median(it), it.max(), it.sum()), flush=True)
        g.close()
    a, b = res["compaction"], res["no compaction"]
    print("   identical costs/status/iterations:", np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), " speedup %.2fx" % (b[0] / a[0]))

B = int(os.environ.get("B", 32768))
x0 = integrator_x0(B); x0[::3] *= 0.05
run("integrator", B, 99, x0, dict(goal=[1.0, 0.5, 0.0, 0.0]), 2)
B2 = B // 2
x0 = acrobot_x0(B2, scale=0.3, seed=3); x0[::2] *= 0.02
run("acrobot", B2, 499, x0, dict(u_min=-1.5, u_max=1.5, params=dict(max_iter=60)), 1)
