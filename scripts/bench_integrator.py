"""Double integrator (n=4, m=2) T=100 B=4096, limits +-0.5: ms per fixed-work iteration and the kernel's phase clocks."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR, capi
B, T, steps = int(os.environ.get("B", 4096)), 100, 20
for dtype in ("f64", "f32"):
    g = BatchILQR("integrator", B, T, 0.02, u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0], flags=capi.FLAG_FIXED_WORK,
                  params=dict(max_iter=steps + 10), dtype=dtype)
    rd = np.random.default_rng(4321)
    g.init_traj(rd.uniform(-1, 1, size=(B, 4)) * np.array([1.5, 1.5, 0.5, 0.5]), np.zeros((B, T, 2)))
    g.iterate(3)
    g.profile(True); g.profile_reset(); g.synchronize()
    t0 = time.perf_counter(); g.iterate(steps); g.synchronize(); el = time.perf_counter() - t0
    p = g.profile_read()
    print(dtype, "B", B, "ms/iter %.4f" % (el / steps * 1e3), "value %.4g" % (B * T * steps / el), {k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if v[1]})
    g.close()
