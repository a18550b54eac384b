"""experiment builds only (-DILQR_HEX_DEBUG): per-segment shader cycles of the hex chain's step, tile 0 / pair 0"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK | int(os.environ.get("FL", "0")), lib=os.environ.get("LIB"))
g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
g.iterate(4)
out = (C.c_longlong * 16)()
g.lib.ilqr_debug_read(out, 1)
N = 10
g.iterate(N)
g.cost()
g.lib.ilqr_debug_read(out, 1)
seg = np.array(list(out)[:6], dtype=float) / (N * T)
names = ["5->0 loop/top", "0->1 W exch", "1->2 Qxx..", "2->3 prefetch issue", "3->4 box-QP", "4->5 V update+exch"]
for n, s_ in zip(names, seg):
    print("%-22s %8.1f cycles/step" % (n, s_))
print("total %.1f" % seg.sum())
