"""The pendulum-chain user twin (examples/user_model_pendulum_chain.hpp, n = 16, m = 4) T = 200: stage times per fixed-work iteration.
    python scripts/bench_chain.py [B] [iters] [lib]      (lib: another build of the twin's library, for A/B runs)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, _build, capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = sys.argv[3] if len(sys.argv) > 3 else _build.USER_CHAIN_LIB
NL, T, DT, lim = 8, 200, 0.02, 2.0
prm = np.array([9.81, 0.1, 2.0, 10.0, 1.0, 0.1, 50.0, 0.0])
rng = np.random.default_rng(3)
x0 = np.concatenate([rng.uniform(-1, 1, (B, NL)), rng.uniform(-1, 1, (B, NL)) * 0.5], axis=1)
g = BatchILQR("user", B, T, DT, u_min=-lim, u_max=lim, lib=lib, nx=2 * NL, nu=NL // 2, user_params=prm, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=iters + 3))
c0 = g.init_traj(x0, np.zeros((B, T, NL // 2)))
g.iterate(1)
g.profile(True)
g.profile_reset()
g.synchronize()
t0 = time.perf_counter()
g.iterate(iters)
g.synchronize()
dt = time.perf_counter() - t0
p = g.profile_read()
print("%s: chain n=16 m=4 T=%d B=%d: %.2f ms per iteration -> %.3e trajectory-timesteps/s" % (os.path.basename(lib), T, B, dt / iters * 1e3, B * T * iters / dt),
      {k: round(ms / n, 3) for k, (ms, n) in p.items() if n}, "cost %.6g -> %.6g" % (c0.mean(), g.cost().mean()))
g.close()
