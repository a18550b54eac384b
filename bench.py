#!/usr/bin/env python
"""bench.py -- iLQR iterations/sec as trajectory-timesteps/sec on MI355X.

A "step" is one full iLQR iteration of the hot path over one batch of synthetic acrobot
problems: finite-difference derivatives -> backward Riccati pass with box-QP -> 11-alpha
line-search rollouts -> accept/commit, for every trajectory (ILQR_FLAG_FIXED_WORK: no
trajectory leaves its loop, so the work per step is exactly B*T trajectory-timesteps).

Workload (BASELINE.json metric, configs[2]): acrobot n=4 m=1, T=499 transitions (500 knots),
B=4096 per GPU, fp64, control limits +-1.5 (box-QP clamps active), x0 ~ (pi a, pi b, c, d),
a..d ~ U(-1,1), u0 = 0.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see README/DESIGN.md for the field contract).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)


def algorithmic_bytes_per_timestep(n, m):
    """SURVEY.md 8(d) / BASELINE.md 4: fp64 bytes per trajectory-timestep of each kernel."""
    s = 8
    return {
        "backward": (2 * n * n + 2 * n * m + m * m + n + 2 * m + m * n + m) * s,          # acrobot 416
        "derivatives": (n + m + 2 * n * n + 2 * n * m + m * m + n + m) * s,              # acrobot 408
        # per alpha: reads us,k (2m), K (mn), xs (n); writes u_t (m) and one checkpoint state per 8 knots
        "rollout": 11 * (2 * m + m * n + n + m + n / 8.0) * s,                            # acrobot 11*92
        "accept": 2 * (n + m) * s,                                                       # commit copy
        # k_sweep_backward (ilqr_iterate): the sweep's records reach the backward wavefront through LDS,
        # so HBM sees: candidate u + 1/8 checkpoint x read, committed x,u written, the record written
        # (retry passes / getters read it there), K and k written, the nominal u read.      acrobot 460
        "sweep_backward": (m + n / 8.0 + (n + m) + (2 * n * n + 2 * n * m + m * m + n + m) + (m * n + m) + m) * s,
    }


def cpu_baseline(B_total, T, dt, lim, target_wall_s=4.0):
    """The CPU oracle (plain-C restatement of the reference, OpenMP over trajectories) on a
    bounded sample of the same workload, on this box's host cores: the same fixed-work
    iterations over as many of the bench's own trajectories as fit the time budget."""
    from oracle import oracle as O
    from tests.util import acrobot_x0
    cores = os.cpu_count() or 1
    om = O.Model("acrobot", u_lim=lim)
    x0_all = acrobot_x0(B_total)

    def run(nb, iters):
        x0 = x0_all[:nb]
        u0 = np.zeros((nb, T, 1))
        t0 = time.perf_counter()
        O.batch_solve(om, x0, u0, dt, max_iters=iters, fixed_work=True, nthreads=cores)
        return time.perf_counter() - t0

    nb = min(B_total, 4 * cores)
    t = run(nb, 1)                      # calibration
    rate = nb * T / t
    iters = 4
    nb2 = int(min(B_total, max(cores, target_wall_s * rate / (T * iters))))
    nb2 = max(cores, (nb2 // cores) * cores) if nb2 >= cores else nb2
    if nb2 == B_total:                  # the whole batch is still too quick: run more iterations
        iters = int(min(40, max(iters, target_wall_s * rate / (T * nb2))))
    t2 = run(nb2, iters)
    # backward pass alone on fixed derivatives (the north star's backward-only figure), same threads
    nb3 = min(B_total, 8 * cores)
    u3 = np.zeros((nb3, T, 1))
    xs3, us3, _ = O.batch_rollout(om, x0_all[:nb3], u3, dt, nthreads=cores)
    dv3 = O.batch_derivatives(om, xs3, us3, dt, nthreads=cores)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        O.batch_backward(om, us3, dv3, lam=1.0, nthreads=cores)
    t3 = (time.perf_counter() - t0) / reps
    return {"value": nb2 * T * iters / t2, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
            "backward_only_value": nb3 * T / t3,
            "sample": "%d trajectories x %d fixed-work iterations of the same acrobot workload, %.1f s wall "
                      "(%.0f core-seconds), oracle/liboracle_ilqr.so with OpenMP over trajectories on all host "
                      "threads" % (nb2, iters, t2, t2 * cores)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/): the
    bench cannot run rocprofv3 on itself, so it quotes the last collected figures when present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        e = d["kernels"].get(kernel)
        return (e["hbm_bytes_per_launch"], d.get("source")) if e else (None, None)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--T", type=int, default=499)
    ap.add_argument("--limit", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra ilqr_flags (kernel variant selection)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the N-rank job (one process per GPU) instead
        # of quietly measuring one GPU
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, have))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    from ilqr_amd import BatchILQR, capi
    from ilqr_amd import dist as D
    from tests.util import acrobot_x0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: one rank per GPU" % (args.gpus, world))
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: no HIP device %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B, T, dt, lim, n, m = args.batch, args.T, 0.02, args.limit, 4, 1
    # this rank's shard of the global synthetic batch (trajectory b of rank r = global r*B + b)
    lo, hi = D.shard(B * world, rank, world)
    x0 = acrobot_x0(B * world)[lo:hi]
    u0 = np.zeros((B, T, m))
    stream = torch.cuda.current_stream().cuda_stream
    g = BatchILQR("acrobot", B, T, dt, u_min=-lim, u_max=lim, device=local_rank,
                  flags=capi.FLAG_FIXED_WORK | args.flags, stream=stream)
    g.init_traj(x0, u0)
    g.iterate(args.warmup)
    cost_dev = torch.zeros(B, dtype=torch.float64, device="cuda")
    if world > 1:  # warm-up of the one collective of the path: RCCL sets its rings up on the first call
        D.gather_costs(cost_dev)
        torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    g.profile(True)
    g.profile_reset()
    barrier()
    t0 = time.perf_counter()
    g.iterate(args.steps)
    # the one exchange step of the path: gather of per-trajectory costs (RCCL over xGMI)
    capi.check(g.lib.ilqr_copy_cost_to_device(g.h, cost_dev.data_ptr()))
    g.synchronize()  # the copy runs on the handle's stream, the collective on torch's: order them
    gathered = D.gather_costs(cost_dev)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device="cuda")
    prof = g.profile_read()
    g.profile(False)

    # backward-pass-only figure of the north star (outside the timed region): the stand-alone quad
    # kernel, one pass at the current lambda, on the derivative records of the final nominal trajectory
    capi.check(g.lib.ilqr_compute_derivatives(g.h))
    R = 20
    for _ in range(10):  # code load of this kernel + clocks back up after the host-side gather
        capi.check(g.lib.ilqr_backward_pass(g.h, None))
    g.profile(True)
    g.profile_reset()
    for _ in range(R):
        capi.check(g.lib.ilqr_backward_pass(g.h, None))
    bw_ms = g.profile_read()["backward"][0] / R
    g.profile(False)

    if rank == 0:
        costs = gathered.cpu().numpy()
        assert np.all(np.isfinite(costs)), "non-finite cost in the gathered result"
        steps = args.steps
        value = world * B * T * steps / elapsed
        bytes_ts = algorithmic_bytes_per_timestep(n, m)
        bytes_ts_backward_only = bytes_ts["backward"]
        stages = {}
        for name, (ms, launches) in prof.items():
            if launches:
                stages[name] = {"ms_per_launch": ms / launches, "launches": launches,
                                "algorithmic_GBps": bytes_ts[name] * B * T / (ms / launches * 1e-3) / 1e9}
        name_of = {i: g.lib.ilqr_stage_kernel_name(g.h, i).decode() for i in range(capi.NUM_STAGES)}
        if name_of[capi.STAGE_NAMES.index("backward")] == "k_sweep_backward":
            # one kernel does the derivative sweep AND the backward pass of the tile
            bytes_ts["backward"] = bytes_ts["sweep_backward"]
            stages["backward"]["algorithmic_GBps"] = bytes_ts["backward"] * B * T / (stages["backward"]["ms_per_launch"] * 1e-3) / 1e9
            stages["backward"]["includes"] = "derivative sweep (fused kernel)"
        dom = max(stages, key=lambda k: stages[k]["ms_per_launch"])
        dom_kernel = name_of[capi.STAGE_NAMES.index(dom)]
        achieved = stages[dom]["algorithmic_GBps"]
        traffic, traffic_src = pmc_traffic(dom_kernel)
        out = {
            "metric": "iLQR iterations/sec (batch x T timesteps/sec), acrobot T=500 batch=4096",
            "value": value, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "acrobot n=4 m=1 T=499 transitions (500 knots) B=%d per GPU, u in [-%.1f,%.1f] "
                                   "(box-QP clamps active), fp64, full iteration = FD derivatives + backward/box-QP "
                                   "+ 11-alpha rollouts + accept, fixed work" % (B, lim, lim),
                       "batch_per_gpu": B, "T": T, "parallelism": "batch-sharded x%d, no data-path collective; "
                       "one all_gather of per-trajectory costs at the end" % world},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_ts[dom] * B * T,
                         "avg_launch_ms": stages[dom]["ms_per_launch"]},
            "stages": stages,
            # north star "backward-pass throughput": k_backward_q alone on fixed derivative records
            "backward_only": {"kernel": "k_backward_q", "ms_per_launch": bw_ms,
                              "timesteps_per_s": B * T / (bw_ms * 1e-3) * world,
                              "algorithmic_GBps": bytes_ts_backward_only * B * T / (bw_ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": bytes_ts_backward_only * B * T / (bw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "final_cost_mean": float(np.mean(costs)),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, T, dt, lim)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
