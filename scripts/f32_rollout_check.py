"""fp32 closed-loop rollouts: device float rollouts and the oracle's float rollouts of the SAME float gains, each against the
fp64 rollout of those gains (how far is each float realisation from the truth?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR
from oracle import oracle as O
from oracle.oracle import ALPHAS
from tests.util import acrobot_x0
from tests import parity as P
B, T, DT, lim = 64, 499, 0.02, 5.0
O.build()
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
x0 = f32(acrobot_x0(B, seed=5))
u0 = np.zeros((B, T, 1))
g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, dtype="f32")
g.init_traj(x0, u0)
g.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
st = P.gpu_state(g)          # float-valued state: nominal, gains
g.compute_derivatives(); g.backward_step()
k, K = g.gains()
cc = g.rollout_candidates()  # [B][11] device float rollouts of (k, K) from the nominal
om64 = O.Model("acrobot", u_lim=lim)
ed, eo = [], []
for a in range(11):
    with O.flavour("f32"):
        om32 = om64.twin("f32")
        _, _, c32 = O.batch_rollout(om32, x0.astype(np.float32), (st["us"] + ALPHAS[a] * k).astype(np.float32), DT, xs_nom=st["xs"].astype(np.float32), K=K.astype(np.float32))
    _, _, c64 = O.batch_rollout(om64, x0, st["us"] + ALPHAS[a] * k, DT, xs_nom=st["xs"], K=K)
    ed.append(np.abs(cc[:, a] - c64) / np.abs(c64)); eo.append(np.abs(np.asarray(c32, dtype=np.float64) - c64) / np.abs(c64))
ed, eo = np.array(ed), np.array(eo)
fin = np.isfinite(ed) & np.isfinite(eo)
print("device float rollouts vs fp64: median %.2e  p90 %.2e  max %.2e" % (np.median(ed[fin]), np.percentile(ed[fin], 90), ed[fin].max()))
print("oracle float rollouts vs fp64: median %.2e  p90 %.2e  max %.2e" % (np.median(eo[fin]), np.percentile(eo[fin], 90), eo[fin].max()))
print("ratio device/oracle: median %.2f, fraction with device > 10 x oracle: %.3f" % (np.median(ed[fin] / np.maximum(eo[fin], 1e-12)), np.mean(ed[fin] > 10 * np.maximum(eo[fin], 1e-9))))
