"""The separable-cost route of k_derivatives_g against the every-point route (experiment build
-DILQR_LQ_SEPARABLE=0): same records, bit for bit when the MFMA route is off (-DILQR_LQ_MFMA_CXX=0);
with it (default) every x'Qx is summed in the matrix cores' order and the cost derivatives differ
by the finite differences' own rounding noise (cxx ~1e-9, first derivatives ~1e-12).  The same script
pins the matrix-core Jacobian sweep: a build with -DILQR_LQ_MFMA_FX=0 (thread-per-point dynamics) gives
identical fx, fu (and everything else).   python scripts/lq_separable_check.py <other.so>"""
import os, subprocess, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 2 and sys.argv[2] == "child":
    from ilqr_amd import BatchILQR
    from tests.test_gpu_lq_end_to_end import dense_mats
    out = {}
    for (n, m, B, T) in ((32, 16, 4, 9), (7, 3, 20, 12)):
        mats = dense_mats(n, m)
        g = BatchILQR("lq", B, T, 0.02, u_min=-1.0, u_max=1.0, lq=mats)
        rng = np.random.default_rng(1)
        g.init_traj(rng.uniform(-1, 1, (B, n)), rng.normal(size=(B, T, m)) * 0.3)
        g.compute_derivatives()
        out[(n, m)] = g.derivatives()
    pickle.dump(out, open(sys.argv[3], "wb"))
else:
    res = []
    for i, lib in enumerate((None, sys.argv[1])):
        env = dict(os.environ)
        if lib:
            env["ILQR_AMD_LIB"] = lib
        f = "/tmp/lqsep%d.pkl" % i
        subprocess.check_call([sys.executable, __file__, "x", "child", f], env=env)
        res.append(pickle.load(open(f, "rb")))
    for key in res[0]:
        for name in res[0][key]:
            a, b = res[0][key][name], res[1][key][name]
            print(key, name, "identical" if np.array_equal(a, b) else "max abs diff %.3e" % np.abs(a - b).max())
