"""PCIe-inclusive rate: host buffers in (x0, u0), results out, 20 fixed-work iterations of the bench workload (B = 4096).
Four ways of fetching the results: the per-array getters into fresh arrays / into reused arrays (ABI <= 4), and ilqr_get_results_async
(ABI 5: one call, no synchronisation per array) into reused page-locked buffers -- all of xs, us, k, K, cost (164 MB), and xs, us, cost alone
(82 MB: what a caller of the reference's solve() reads back, include/ilqr.h:48-56)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T, iters = 4096, 499, 20
x0 = acrobot_x0(B); u0 = np.zeros((B, T, 1))
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=10**6))
g.init_traj(x0, u0); g.iterate(2); g.synchronize()
from ilqr_amd.batch import _p
# result buffers a caller keeps across solves (touched once: a fresh np.zeros pays ~40 K page faults inside the copy)
oxs = np.zeros((B, T + 1, 4)); ous = np.zeros((B, T, 1)); ok = np.zeros((B, T, 1)); oK = np.zeros((B, T, 4, 1)); oc = np.zeros(B)
for a in (oxs, ous, ok, oK, oc):
    a.fill(1.0)
pinned = g.result_buffers(pinned=True)
pinned_xu = {k: pinned[k] for k in ("xs", "us", "cost")}
modes = ["fresh result arrays, getters", "reused result arrays, getters", "reused result arrays, getters",
         "page-locked buffers, ilqr_get_results_async (xs,us,k,K,cost: %.0f MB)" % (sum(a.nbytes for a in pinned.values()) / 1e6)] * 1
modes += [modes[-1]] * 2 + ["page-locked buffers, ilqr_get_results_async (xs,us,cost: %.0f MB)" % (sum(a.nbytes for a in pinned_xu.values()) / 1e6)] * 3
for rep, mode in enumerate(modes):
    t0 = time.perf_counter(); g.init_traj(x0, u0); t1 = time.perf_counter()
    g.iterate(iters); g.synchronize(); t2 = time.perf_counter()
    if mode.startswith("fresh"):
        xs, us = g.trajectory(); k, K = g.gains(); c = g.cost()
    elif mode.startswith("reused"):
        capi.check(g.lib.ilqr_get_trajectory(g.h, _p(oxs), _p(ous))); capi.check(g.lib.ilqr_get_gains(g.h, _p(ok), _p(oK)))
        capi.check(g.lib.ilqr_get_cost(g.h, _p(oc)))
    else:
        g.results_async(pinned if "k,K" in mode else pinned_xu); g.synchronize()
    t3 = time.perf_counter()
    print("[%s] upload+init %.2f ms | %d iterations %.2f ms | download %.2f ms | resident %.3e ts/s | PCIe-inclusive %.3e ts/s"
          % (mode, (t1 - t0) * 1e3, iters, (t2 - t1) * 1e3, (t3 - t2) * 1e3, B * T * iters / (t2 - t1), B * T * iters / (t3 - t0)))

g.close()
