"""The fused sweep + backward pass with four 16-lane backward wavefronts per tile (ILQR_AMD_HEX=1) against the default
(one 4-lane backward wavefront): same solves to rounding, and the time of the phase.
    python scripts/hex_fused_check.py [B] [iters] [flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
flags = int(sys.argv[3]) if len(sys.argv) > 3 else capi.FLAG_STAGED
T, DT = 499, 0.02
rng = np.random.default_rng(0)
x0 = rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]) * 0.5
outs = {}
for label in ("quad", "hex"):
    if label == "hex":
        os.environ["ILQR_AMD_HEX"] = "1"
    try:
        g = BatchILQR("acrobot", B, T, DT, flags=flags | capi.FLAG_FIXED_WORK, u_min=-1.5, u_max=1.5, params=dict(max_iter=1000))
        g.init_traj(x0, np.zeros((B, T, 1)))
        g.iterate(2)
        g.profile(True)
        g.profile_reset()
        g.iterate(iters)
        p = g.profile_read()
        st, it, al = g.status()
        lam, dlam = g.lambdas()
        outs[label] = dict(cost=g.cost(), lam=lam, it=it, prof={k: round(ms / max(n, 1), 4) for k, (ms, n) in p.items() if n})
        print(label, outs[label]["prof"], "mean cost %.6f" % outs[label]["cost"].mean())
        g.close()
    finally:
        os.environ.pop("ILQR_AMD_HEX", None)
q, h = outs["quad"], outs["hex"]
rel = np.abs(q["cost"] - h["cost"]) / np.abs(q["cost"])
print("cost rel diff: median %.2e  90%% %.2e  99%% %.2e  max %.2e; same lambda: %.4f" % (np.median(rel), np.quantile(rel, 0.9), np.quantile(rel, 0.99), rel.max(), (q["lam"] == h["lam"]).mean()))
