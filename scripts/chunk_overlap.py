"""Experiment: C independent sub-batches on C HIP streams (latency-bound kernels overlap)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T, steps, warm = 4096, 499, 20, 3
x0 = acrobot_x0(B)
for C in (1, 2, 4, 8, 16):
    per = B // C
    gs = []
    for c in range(C):
        g = BatchILQR("acrobot", per, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK)
        g.init_traj(x0[c * per:(c + 1) * per], np.zeros((per, T, 1)))
        gs.append(g)
    for it in range(warm):
        for g in gs: g.iterate(1)
    for g in gs: g.synchronize()
    t0 = time.perf_counter()
    for g in gs: g.iterate(steps)
    for g in gs: g.synchronize()
    el = time.perf_counter() - t0
    print("chunks %2d: %.3f ms/iteration  %.3e timesteps/s" % (C, el / steps * 1e3, B * T * steps / el))
    for g in gs: g.close()
