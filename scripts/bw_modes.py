"""Why does the stand-alone backward kernel time differently in different processes?  (experiment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0)
    x = torch.empty(4096, dtype=torch.float64, device="cuda")
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
flags = capi.FLAG_FIXED_WORK | (capi.FLAG_UNFUSED if "unfused" in sys.argv else 0)
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=flags)
g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
g.profile(True)
g.iterate(5); g.profile_reset(); g.iterate(10)
print("iterate", {k: round(v[0] / max(v[1], 1), 4) for k, v in g.profile_read().items()})
g.compute_derivatives()
for name, fn in (("backward_pass(mode0)", lambda: g.lib.ilqr_backward_pass(g.h, None)), ("backward_step(mode1)", lambda: g.lib.ilqr_backward_step(g.h))):
    for _ in range(5): fn()
    g.profile_reset()
    for _ in range(10): fn()
    print(name, g.profile_read()["backward"][0] / 10)
