"""Long-horizon, late-iteration soak of the matrix-core chains (k_solve_hex) against the quad chain (ILQR_ROUTE_QUAD_CHAIN):
random batches, horizons up to 499, limits, scales, fp64 and fp32; whole solves (per-trajectory exits, lambda retries, slow
box-QP exits, the k >= 16 search) and fixed-work iterations; every array and scalar must be the same bits.

    python scripts/soak_hex.py [seconds] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi

DT = 0.02


def state(g):
    xs, us = g.trajectory()
    k, K = g.gains()
    st, it, al = g.status()
    lam, dlam = g.lambdas()
    return dict(xs=xs, us=us, k=k, K=K, cost=g.cost(), st=st, it=it, al=al, lam=lam, dlam=dlam)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_traj = n_iter = 0
    sv = capi.STAGE_NAMES.index("solve")
    while time.time() < t_end:
        B = int(rng.choice([1, 5, 16, 37, 64, 130, 256, 700]))
        T = int(rng.choice([3, 17, 60, 200, 350, 499]))
        lim = float(rng.choice([0.3, 1.5, 5.0]))
        dtype = "f32" if rng.random() < 0.3 else "f64"
        scale = float(rng.choice([0.05, 0.3, 1.0]))
        fixed = rng.random() < 0.3
        iters = int(rng.integers(2, 30)) if fixed else int(rng.choice([20, 60, 120]))
        x0 = rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]) * scale
        if dtype == "f32":
            x0 = x0.astype(np.float32).astype(np.float64)
        u0 = rng.normal(size=(B, T, 1)) * float(rng.choice([0.0, 0.2]))
        out = []
        for route, kernel in ((0, b"k_solve_hex"), (capi.ROUTE_QUAD_CHAIN, b"k_solve_tile")):
            g = BatchILQR("acrobot", B, T, DT, u_min=-lim, u_max=lim, dtype=dtype, route=route, assume_cus=4096,
                          flags=capi.FLAG_FIXED_WORK if fixed else 0, params=dict(max_iter=iters))
            assert g.lib.ilqr_stage_kernel_name(g.h, sv) == kernel, g.lib.ilqr_stage_kernel_name(g.h, sv)
            g.init_traj(x0, u0)
            if fixed:
                g.iterate(iters)
            else:
                g.generate_trajectory()
            out.append(state(g))
            g.close()
        for key in out[0]:
            if not np.array_equal(out[0][key], out[1][key], equal_nan=True):
                bad = np.argwhere(~((out[0][key] == out[1][key]) | (np.isnan(out[0][key]) & np.isnan(out[1][key]))))
                print("MISMATCH in %s: B %d T %d lim %g %s scale %g fixed %s iters %d seed %d case %d, first at %s" % (
                    key, B, T, lim, dtype, scale, fixed, iters, seed, n_cases, bad[0] if len(bad) else "?"))
                sys.exit(1)
        n_cases += 1
        n_traj += B
        n_iter += int(np.sum(out[0]["it"])) if not fixed else B * iters
    print("soak_hex ok: %d cases, %d trajectories, %d trajectory-iterations, k_solve_hex == k_solve_tile<1> bit for bit, seed %d" % (n_cases, n_traj, n_iter, seed))


if __name__ == "__main__":
    main()
