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

Rank 0 prints ONE JSON line on stdout: a COMPACT record (< 2 KB: metric, value, config, roofline, cpu_baseline -- what the
driver parses).  Everything else this script can measure (per-stage table, the issue roofline with its sources, the
backward-only figure, and -- with --extra-configs -- the other BASELINE configurations, each with a CPU baseline of its own)
goes to a side file (--extras-out, default gpurun_out/bench_extras.json), never to stdout.
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
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (dense)


def algorithmic_bytes_per_timestep(n, m, s=8):
    """SURVEY.md 8(d) / BASELINE.md 4: bytes per trajectory-timestep of each kernel (s = sizeof scalar)."""
    return {
        "backward": (2 * n * n + 2 * n * m + m * m + n + 2 * m + m * n + m) * s,          # acrobot fp64 416
        "derivatives": (n + m + 2 * n * n + 2 * n * m + m * m + n + m) * s,              # acrobot fp64 408
        # UNIQUE bytes of the 11-alpha search: the nominal us,k (2m), K (mn), xs (n) are read once per tile and
        # shared by the 11 candidates; every candidate writes its u_t (m) and one checkpoint state per 8 knots.
        # (SURVEY's nominal figure charges the reads to every alpha: 11 x 92 B; that is not what reaches HBM.)
        "rollout": ((2 * m + m * n + n) + 11 * (m + n / 8.0)) * s,                        # acrobot fp64 212
        "accept": 2 * (n + m) * s,                                                       # commit copy
        # fused sweep + backward (k_sweep_backward / the first phase of k_solve_tile): the records live in LDS
        # only, so HBM sees: candidate u + 1/8 checkpoint x read, committed x,u written, K and k written, the
        # nominal u read.                                                                    acrobot fp64 100
        "sweep_backward": (m + n / 8.0 + (n + m) + (m * n + m) + m) * s,
    }


def cpu_baseline(B_total, T, dt, lim, target_wall_s=4.0, flavour="f64", backward_only=True, reps=3):
    """The CPU oracle (plain-C restatement of the reference, OpenMP over trajectories) on a
    bounded sample of the same workload, on this box's host cores: the same fixed-work
    iterations over as many of the bench's own trajectories as fit the time budget.
    flavour "f32": the oracle's float twin (what an fp32 handle is compared with).

    Sample length: one 1.6-2 s run moved +-25 % between boxes and runs (rounds 4/5: thread start-up, page first-touch
    and host clocks inside the sample).  The figure is now the MEDIAN of `reps` runs of about `target_wall_s` seconds
    each after an untimed run of the same size (threads and pages warm), and the line carries the spread
    (max - min) / median of those runs -- within +-5 % at 3 x 4 s on the boxes this was tried on."""
    from oracle import oracle as O
    with O.flavour(flavour):
        return _cpu_baseline_acrobot(O, B_total, T, dt, lim, target_wall_s, flavour, backward_only, reps)


def _cpu_baseline_acrobot(O, B_total, T, dt, lim, target_wall_s, flavour, backward_only, reps):
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
    run(nb, 1)                          # threads up, code paged in
    t = run(nb, 1)                      # calibration
    rate = nb * T / t
    iters = 4
    nb2 = int(min(B_total, max(cores, target_wall_s * rate / (T * iters))))
    nb2 = max(cores, (nb2 // cores) * cores) if nb2 >= cores else nb2
    if nb2 == B_total:                  # the whole batch is still too quick: run more iterations
        iters = int(min(40, max(iters, target_wall_s * rate / (T * nb2))))
    t1 = run(nb2, 1)                    # first touch of this sample's pages; also the better calibration (threads and pages warm, this sample)
    iters = int(min(40, max(iters, round(target_wall_s / max(t1, 1e-9)))))
    ts = sorted(run(nb2, iters) for _ in range(max(1, reps)))
    t2 = ts[len(ts) // 2]
    lib_name = {"f64": "liboracle_ilqr.so", "f32": "liboracle_ilqr_f32.so"}.get(flavour, flavour)
    res = {"value": nb2 * T * iters / t2, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
           "spread": (ts[-1] - ts[0]) / t2,
           "sample": "median of %d runs of %d trajectories x %d fixed-work iterations of this workload (%.1f s each), oracle/%s, OpenMP on all host threads"
                     % (len(ts), nb2, iters, t2, lib_name)}
    if not backward_only:
        return res
    # backward pass alone on fixed derivatives (the north star's backward-only figure), same threads
    nb3 = min(B_total, 8 * cores)
    u3 = np.zeros((nb3, T, 1))
    xs3, us3, _ = O.batch_rollout(om, x0_all[:nb3], u3, dt, nthreads=cores)
    dv3 = O.batch_derivatives(om, xs3, us3, dt, nthreads=cores)
    nrep = 3
    t0 = time.perf_counter()
    for _ in range(nrep):
        O.batch_backward(om, us3, dv3, lam=1.0, nthreads=cores)
    t3 = (time.perf_counter() - t0) / nrep
    res["backward_only_value"] = nb3 * T / t3
    return res


def cpu_baseline_other(kind, T, dt, target_wall_s=3.0):
    """The same bounded-sample timing for the other bench workloads (kind = "integrator" | "lq"): fixed-work iterations
    of the oracle with OpenMP over trajectories on this box's host threads."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    if kind == "integrator":
        om = O.Model("integrator", goal=[1.0, 0.5, 0.0, 0.0], u_lim=0.5)
        rd = np.random.default_rng(4321)
        x0_all = rd.uniform(-1, 1, size=(4096, 4)) * np.array([1.5, 1.5, 0.5, 0.5])
        nu, what = 2, "double integrator T=%d +-0.5" % T
    else:
        om = O.Model("lq", lq=lq_mats(32, 16), u_lim=1.0)
        x0_all = np.random.default_rng(0).uniform(-1, 1, (8192, 32))
        nu, what = 16, "LQ n=32 m=16 T=%d +-1, finite differences" % T

    def run(nb, iters):
        t0 = time.perf_counter()
        O.batch_solve(om, x0_all[:nb], np.zeros((nb, T, nu)), dt, max_iters=iters, fixed_work=True, nthreads=cores)
        return time.perf_counter() - t0

    nb, iters = min(len(x0_all), cores), 1
    t = run(nb, iters)  # one trajectory per thread, one iteration: already the whole sample for the LQ model
    while t < target_wall_s / 2 and (nb < len(x0_all) or iters < 32):  # grow the sample until it is worth timing
        if nb < len(x0_all):
            nb, iters = min(len(x0_all), nb * 4), 4
        else:
            iters *= 2
        t = run(nb, iters)
    return {"value": nb * T * iters / t, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
            "sample": "%d trajectories x %d fixed-work iteration(s) of %s, %.1f s wall, oracle/liboracle_ilqr.so with OpenMP over "
                      "trajectories on all host threads" % (nb, iters, what, t)}


def cpu_baseline_lq_exact(T, dt, target_wall_s=3.0):
    """The LQ workload with EXACT derivatives has no counterpart in the reference (src/derivatives.cpp is finite differences only)
    and none in the oracle.  What the device's iteration consists of there -- one backward pass with the box-QP, the eleven
    closed-loop rollouts of the search -- timed with the oracle's own stages on fixed derivative records, OpenMP over trajectories."""
    from oracle import oracle as O
    from tests.util import mat
    cores = os.cpu_count() or 1
    om = O.Model("lq", lq=lq_mats(32, 16), u_lim=1.0)
    nb = cores
    x0 = np.random.default_rng(0).uniform(-1, 1, (nb, 32))
    xs, us, _ = O.batch_rollout(om, x0, np.zeros((nb, T, 16)), dt, nthreads=cores)
    dv = O.batch_derivatives(om, xs, us, dt, nthreads=cores)  # (not timed: the exact records are constant up to cx, cu)
    reps, t_total = 0, 0.0
    while t_total < target_wall_s and reps < 64:
        t0 = time.perf_counter()
        ro = O.batch_backward(om, us, dv, lam=1.0, nthreads=cores)
        K = mat(ro["K"])  # [B][T][nu][nx]
        for a in O.ALPHAS:
            O.batch_rollout(om, x0, us + a * ro["k"], dt, xs_nom=xs, K=K, nthreads=cores)
        t_total += time.perf_counter() - t0
        reps += 1
    return {"value": nb * T * reps / t_total, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
            "sample": "%d trajectories x %d repetitions of (oracle backward pass + the 11 closed-loop rollouts of the search) on fixed records of "
                      "the LQ n=32 m=16 T=%d workload, %.1f s wall, oracle/liboracle_ilqr.so with OpenMP over trajectories on all host threads; the "
                      "reference has no exact-derivative mode, so no sweep is timed" % (nb, reps, T, t_total)}


def counters():
    """profiles/traffic.json: what the rocprofv3 PMC passes of scripts/collect_profiles.sh counted per kernel (HBM bytes,
    VALU instructions).  The bench cannot run rocprofv3 on itself; it may only quote counters that describe THE CODE IT
    IS TIMING: the file carries the content hash of the library sources it was collected on, and a file collected on other
    sources is not used (the fields it would have filled say "stale")."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/traffic.json"
    try:
        from ilqr_amd import _build
        d = json.load(open(path))
        if d.get("source_hash") != _build._source_hash():
            return None, "stale: profiles/traffic.json was collected on other library sources (hash %s..., now %s...)" % (
                str(d.get("source_hash"))[:12], _build._source_hash()[:12])
        return d, d.get("source")
    except Exception as e:  # noqa: BLE001
        return None, "unreadable profiles/traffic.json: %s" % e


def pmc_traffic(kernel, iterations_per_launch=1):
    """HBM bytes per launch of `kernel` from the counter passes (the persistent kernel's are kept per iteration and scaled
    to the launch timed here); None + the reason when there are no counters for this code."""
    d, note = counters()
    if d is None:
        return None, note
    e = d["kernels"].get(kernel)
    if not e:
        return None, "no counters for %s in profiles/traffic.json" % kernel
    if "hbm_bytes_per_iteration" in e:
        return e["hbm_bytes_per_iteration"] * iterations_per_launch, note
    return e["hbm_bytes_per_launch"], note


# VALU issue: one wave64 instruction occupies a SIMD's 16 lanes for 4 cycles, so a SIMD issues at most 0.25
# VALU instructions per cycle (MI355X_MICROARCH.md).  The dominant kernel is one dependent chain per tile and is
# bound by that, not by bytes: the figure quoted is from the committed SQ-counter pass (profiles/), which the
# bench cannot collect on itself.
def issue_roofline(kernel, iteration_ms, sclk_mhz, timesteps, n_simds=1024):
    """VALU issue of `kernel` in THIS run: instructions per trajectory-timestep (a property of the code: SQ_INSTS_VALU of
    the counter pass, valid only for the sources it was collected on) x the timesteps of an iteration, over the SIMD
    cycles the iteration took HERE -- duration from the HIP events of this run, clock from the kernel's own
    s_memtime / wall-clock ratio in this run (the chip lowers its clock with the number of busy SIMDs)."""
    d, note = counters()
    e = d["kernels"].get(kernel) if d else None
    if not e or "valu_insts_per_iteration" not in e or not sclk_mhz:
        return {"bound": "valu_issue", "kernel": kernel, "achieved": None, "peak": 0.25, "unit": "VALU instructions / cycle / SIMD",
                "frac": None, "sclk_mhz_measured": sclk_mhz, "note": note}
    per_ts = e["valu_insts_per_iteration"] / e["timesteps_per_iteration"]
    ipc = per_ts * timesteps / (n_simds * iteration_ms * 1e-3 * sclk_mhz * 1e6)
    return {"bound": "valu_issue", "kernel": kernel, "achieved": ipc, "peak": 0.25, "unit": "VALU instructions / cycle / SIMD",
            "frac": ipc / 0.25, "valu_insts_per_timestep": per_ts, "sclk_mhz_measured": sclk_mhz, "iteration_ms_measured": iteration_ms,
            "wait_fraction_of_wave_cycles": e.get("wait_fraction_of_wave_cycles"),
            "lds_bank_conflict_fraction": e.get("lds_bank_conflict_fraction"), "source": note}


COMPACT_LIMIT = 2048  # bytes: the driver keeps the last 8 KB of stdout; round 1's 2.3 KB line parsed, round 5's 27 KB line did not


def _r(x, sig=6):
    """Round floats to `sig` significant digits (the line is a record, not a dump)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


def compact_record(full):
    """The ONE stdout line: the contract's fields + roofline + cpu_baseline, short strings, numbers to 6 digits.
    `full` is the complete result dict (what goes to the extras file)."""
    roof, cpu = full.get("roofline") or {}, full.get("cpu_baseline")
    rec = {k: _r(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data") if k in full}
    cfg = full.get("config") or {}
    rec["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "global_batch", "T", "u_limit", "parallelism") if k in cfg}
    rec["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                     "algorithmic_bytes_per_launch", "avg_launch_ms", "limiter", "bound_frac")}
    if cpu:
        rec["cpu_baseline"] = {k: _r(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "spread")}
    for k in ("rccl_ranks", "backward_only_timesteps_per_s", "extras"):
        if k in full:
            rec[k] = _r(full[k])
    line = json.dumps(rec, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, "bench.py's stdout record grew to %d bytes (limit %d): move detail to the extras file" % (len(line), COMPACT_LIMIT)
    return line


def lq_mats(n, m, seed=7):
    """SURVEY.md 8(d) cfg 5: A = -I + 0.1 N(0,1)/sqrt(n), B = N(0,1)/sqrt(n), Q = I, R = 0.1 I."""
    rng = np.random.default_rng(seed)
    A = -np.eye(n) + 0.1 * rng.normal(size=(n, n)) / np.sqrt(n)
    Bm = rng.normal(size=(n, m)) / np.sqrt(n)
    return A, Bm, np.eye(n), 0.1 * np.eye(m), np.eye(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU (weak scaling: the default)")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: this many trajectories in total, each of the N ranks its contiguous "
                    "1/N (e.g. 32768: BASELINE configs[3]'s partition at --dtype f32 --limit 5); overrides --batch")
    ap.add_argument("--T", type=int, default=499)
    ap.add_argument("--limit", type=float, default=1.5)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64", help="arithmetic of the HEADLINE run (the other "
                    "one is reported under configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-configs", action="store_true", help="also run the other BASELINE configurations (saturated batch, configs[1], [3], [4], the "
                    "double integrator, a user twin), each with a CPU baseline of its own: minutes; their records go to the extras file")
    ap.add_argument("--no-extra-configs", action="store_true", help="(the default since round 6; accepted for the scripts that pass it)")
    ap.add_argument("--extras-out", default=os.path.join(ROOT, "gpurun_out", "bench_extras.json"),
                    help="where the full record (stages, issue roofline, sources, --extra-configs results) is written; '' = nowhere")
    ap.add_argument("--flags", type=int, default=0, help="extra ilqr_flags (kernel variant selection)")
    ap.add_argument("--test-backend", default="", help="TESTS ONLY (tests/test_bench_record.py): run the N-rank path on a box with fewer GPUs than ranks -- "
                    "this torch.distributed backend (gloo) instead of nccl = RCCL, every rank on the device LOCAL_RANK modulo the visible ones, the one "
                    "collective on host tensors.  A line produced this way says so (collective_backend) and is no measurement")
    ap.add_argument("--route", type=int, default=0, help="enum ilqr_route for the headline handle (A/B runs of equivalent kernels)")
    args = ap.parse_args()
    args.no_extra_configs = not args.extra_configs

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
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("bench.py --global-batch %d is not a multiple of %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
    if local_rank >= torch.cuda.device_count() and not args.test_backend:
        raise SystemExit("rank %d: no HIP device %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    if args.test_backend:  # (several ranks share a device: the N-rank code path on a one-GPU test box)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    coll_device = "cpu" if args.test_backend else "cuda"  # where the tensors of the one collective live (RCCL: on the device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.test_backend:
            dist.init_process_group(args.test_backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    T, dt, n, m = args.T, 0.02, 4, 1
    stream = torch.cuda.current_stream().cuda_stream
    cost_dev = torch.zeros(args.batch, dtype=torch.float64, device="cuda")
    if world > 1:  # warm-up of the one collective of the path: RCCL sets its rings up on the first call
        D.gather_costs(cost_dev.to(coll_device))
        torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def acrobot_run(dtype, B, lim, steps, warmup, flags=0, gather=True):
        """W untimed + K timed fixed-work iterations of this rank's shard; returns (handle, seconds = max over ranks,
        stage profile, gathered costs or None)."""
        lo, hi = D.shard(B * world, rank, world)   # trajectory b of rank r = global r*B + b
        x0 = acrobot_x0(B * world)[lo:hi]
        # max_iter beyond the run: ILQR_FLAG_FIXED_WORK keeps every trajectory running (checked below)
        g = BatchILQR("acrobot", B, T, dt, u_min=-lim, u_max=lim, device=local_rank, dtype=dtype,
                      flags=capi.FLAG_FIXED_WORK | flags, stream=stream, params=dict(max_iter=warmup + steps + 1),
                      route=args.route if (B == args.batch and dtype == args.dtype) else 0)
        g.init_traj(x0, np.zeros((B, T, m)))
        g.iterate(warmup)
        g.profile(True)
        g.profile_reset()
        barrier()
        t0 = time.perf_counter()
        g.iterate(steps)
        gathered = None
        if gather:  # the one exchange step of the path: gather of per-trajectory costs (RCCL over xGMI)
            cd = cost_dev if B == args.batch else torch.zeros(B, dtype=torch.float64, device="cuda")
            capi.check(g.lib.ilqr_copy_cost_to_device(g.h, cd.data_ptr()))
            g.synchronize()  # the copy runs on the handle's stream, the collective on torch's: order them
            gathered = D.gather_costs(cd.to(coll_device))
        barrier()
        elapsed = D.max_over_ranks(time.perf_counter() - t0, device=coll_device)
        prof = g.profile_read()
        g.sclk_mhz = g.shader_clock_mhz() if prof.get("solve", (0, 0))[1] else None
        g.profile(False)
        assert g.count_running() == B, "fixed-work run lost trajectories: the throughput figure would be inflated"
        return g, elapsed, prof, gathered

    def stage_table(g, prof, B, s_bytes, iters_per_call):
        bytes_ts = algorithmic_bytes_per_timestep(n, m, s_bytes)
        name_of = {i: g.lib.ilqr_stage_kernel_name(g.h, i).decode() for i in range(capi.NUM_STAGES)}
        fused = name_of[capi.STAGE_NAMES.index("backward")] == "k_sweep_backward"
        if fused:  # one kernel does the derivative sweep AND the backward pass of the tile
            bytes_ts["backward"] = bytes_ts["sweep_backward"]
        # the persistent kernel: every iteration of the call, both phases (the commit of the last accepts is k_commit)
        bytes_ts["solve"] = (bytes_ts["sweep_backward"] + bytes_ts["rollout"]) * iters_per_call
        stages = {}
        persistent = prof.get("solve", (0, 0))[1] > 0
        for name, (ms, launches) in prof.items():
            if launches:
                stages[name] = {"kernel": name_of[capi.STAGE_NAMES.index(name)], "ms_per_launch": ms / launches, "launches": launches,
                                "algorithmic_bytes_per_timestep": bytes_ts[name],
                                "algorithmic_GBps": bytes_ts[name] * B * T / (ms / launches * 1e-3) / 1e9}
                if persistent and name in ("backward", "rollout"):
                    stages[name]["kernel"] = name_of[capi.STAGE_NAMES.index("solve")]
                    stages[name]["clock"] = "the kernel's own per-phase clock, mean over tiles and per iteration (not a launch)"
        if fused:
            stages["backward"]["includes"] = "derivative sweep (fused)"
        if persistent:
            stages["solve"]["iterations_per_launch"] = iters_per_call
        return stages, bytes_ts

    def roofline_of(stages, bytes_ts, B, dtype="f64"):
        real_launches = {k: v for k, v in stages.items() if "clock" not in v}
        dom = max(real_launches, key=lambda k: stages[k]["ms_per_launch"])
        kern = stages[dom]["kernel"]
        achieved = stages[dom]["algorithmic_GBps"]
        # (the counter passes are of the fp64 workloads: a float instantiation has an entry of its own in traffic.json, or none)
        traffic, traffic_src = pmc_traffic(kern + ("_f32" if dtype == "f32" else ""), stages[dom].get("iterations_per_launch", 1))
        # `bound` is the contract's roofline for this byte-light path (HBM: algorithmic bytes over the launch duration);
        # `limiter` names what the counters say actually bounds the kernel and `bound_frac` is the fraction of THAT limit
        # (filled in from roofline_issue when counters of THIS code exist), so nobody reads 0.14 as "14 % of the limiter".
        return {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "limiter": "valu_issue", "bound_frac": None,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_ts[dom] * B * T, "avg_launch_ms": stages[dom]["ms_per_launch"],
                "limiter_note": "VALU issue + latency of dependent chains (the Riccati chains of a tile -- four on the matrix cores at one tile per CU --, then its "
                                "rollout wavefronts), not bytes: see roofline_issue; HBM is the contract's nominal bound for this byte-light path"}


    # ---------------- headline: BASELINE.json metric, configs[2] ----------------
    B, lim, steps = args.batch, args.limit, args.steps
    s_bytes = 8 if args.dtype == "f64" else 4
    g, elapsed, prof, gathered = acrobot_run(args.dtype, B, lim, steps, args.warmup, args.flags)

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
    stages, bytes_ts = stage_table(g, prof, B, s_bytes, steps)
    headline_sclk = g.sclk_mhz
    g.close()

    # ---------------- the other configurations (each a fixed-work run of its own) ----------------
    extra = {}
    if not args.no_extra_configs and world == 1:
        # the saturated regime: the largest batch of the sweep (two persistent tiles per CU, the dispatcher hands a CU its
        # next tile as one finishes): what one MI355X sustains when the batch is not the limit
        Bs = 32768
        gs, els, profs, _ = acrobot_run(args.dtype, Bs, lim, steps, args.warmup, args.flags, gather=False)
        sts, bts = stage_table(gs, profs, Bs, s_bytes, steps)
        sclk = gs.sclk_mhz
        gs.close()
        roofs = roofline_of(sts, bts, Bs, args.dtype)
        roofs["bound_frac"] = issue_roofline(roofs["kernel"], els / steps * 1e3, sclk, Bs * T)["frac"]
        extra["saturated"] = {
            "workload": "the headline workload at B=%d per GPU (persistent wide tiles of 64 trajectories, two per CU, thread-per-trajectory "
                        "backward chain; records never reach HBM)" % Bs,
            "value": Bs * T * steps / els, "unit": "trajectory-timesteps/s", "ms_per_step": els / steps * 1e3, "batch_per_gpu": Bs,
            "stages": sts, "roofline": roofs, "roofline_issue": issue_roofline(roofs["kernel"], els / steps * 1e3, sclk, Bs * T)}
        # BASELINE configs[3] at its stated size, all 32768 trajectories on ONE GPU (fp32, limits +-5): the batch the 8-GPU partition shards
        other = "f32" if args.dtype == "f64" else "f64"
        g5, el5, prof5, _ = acrobot_run(other, 32768, 5.0, steps, args.warmup, gather=False)
        st5, bt5 = stage_table(g5, prof5, 32768, 4 if other == "f32" else 8, steps)
        sclk5 = g5.sclk_mhz
        g5.close()
        roof5 = roofline_of(st5, bt5, 32768, other)
        issue5 = issue_roofline(roof5["kernel"] + ("_f32" if other == "f32" else ""), el5 / steps * 1e3, sclk5, 32768 * T)
        roof5["bound_frac"] = issue5["frac"]
        extra["acrobot_T500_B32768_lim5_%s_one_gpu" % other] = {
            "workload": "acrobot T=499 B=32768 on one GPU, u in [-5,5], %s (BASELINE configs[3] at its stated size, unsharded); the 8 x 4096 partition of "
                        "the same batch gives the same bits (tests/test_gpu_fp32.py)" % other,
            "dtype": other, "value": 32768 * T * steps / el5, "unit": "trajectory-timesteps/s", "ms_per_step": el5 / steps * 1e3,
            "stages": st5, "roofline": roof5, "roofline_issue": issue5}
    if not args.no_extra_configs:
        # late in a solve: iterations 4..103 of the same workload -- box-QPs leave the fast path
        # once lambda has reached 0, the launch lasts as long as its slowest tile
        g2, el2, prof2, _ = acrobot_run(args.dtype, B, lim, 100, args.warmup, args.flags, gather=False)
        st2, _ = stage_table(g2, prof2, B, s_bytes, 100)
        g2.close()
        extra["late_solve"] = {"workload": "the headline workload, 100 timed iterations (4..103 of the solve)",
                               "late_ms_per_step": el2 / 100 * 1e3, "value": world * B * T * 100 / el2,
                               "stages_ms": {k: v["ms_per_launch"] for k, v in st2.items()}}
        # BASELINE configs[3]: acrobot T=500, fp32, 4096 per GPU (32768 over 8), limits +-5 (SURVEY 8d cfg 4);
        # under --dtype f32 this slot holds the fp64 run of the same shape instead
        other = "f32" if args.dtype == "f64" else "f64"
        so = 4 if other == "f32" else 8
        for label, fl in ((other, 0), (other + "_analytic", capi.FLAG_ANALYTIC_DERIVATIVES)):
            g3, el3, prof3, ga3 = acrobot_run(other, 4096, 5.0, steps, args.warmup, fl)
            st3, bt3 = stage_table(g3, prof3, 4096, so, steps)
            roof3 = roofline_of(st3, bt3, 4096, other)
            # (the counter passes are of the finite-difference run: the exact-derivative line carries the HBM figures only)
            issue3 = issue_roofline(roof3["kernel"] + ("_f32" if other == "f32" else ""), el3 / steps * 1e3, g3.sclk_mhz, 4096 * T) if not fl else None
            if issue3:
                roof3["bound_frac"] = issue3["frac"]
            else:
                roof3["traffic"], roof3["traffic_source"] = None, "counter passes exist for the finite-difference run of this workload only"
            assert ga3 is None or bool(torch.isfinite(ga3).all())
            cpu3 = None
            if world == 1 and not args.no_cpu_baseline and not fl:
                cpu3 = cpu_baseline(4096, T, dt, 5.0, target_wall_s=2.0, flavour=other, backward_only=False)
            extra["acrobot_T500_B4096_lim5_" + label] = {
                "workload": "acrobot T=499 B=4096 per GPU, u in [-5,5], %s%s, fixed-work iterations (BASELINE configs[3] "
                            "per-GPU shard; x %d GPUs)" % (other, ", exact model derivatives instead of finite differences" if fl else
                                                             ("; finite differences taken in double from the float knot" if other == "f32" else ""), world),
                "dtype": other, "value": world * 4096 * T * steps / el3, "unit": "trajectory-timesteps/s", "ms_per_step": el3 / steps * 1e3,
                "n_gpus": world, "stages": st3, "roofline": roof3}
            if issue3:
                extra["acrobot_T500_B4096_lim5_" + label]["roofline_issue"] = issue3
            if cpu3:
                extra["acrobot_T500_B4096_lim5_" + label]["cpu_baseline"] = cpu3
            g3.close()
    if not args.no_extra_configs and world == 1:
        # BASELINE configs[1]: acrobot B=1024, limits +-5
        g4, el4, prof4, _ = acrobot_run("f64", 1024, 5.0, steps, args.warmup, gather=False)
        st4, bt4 = stage_table(g4, prof4, 1024, 8, steps)
        g4.close()
        extra["acrobot_T500_B1024_lim5_f64"] = {"workload": "acrobot T=499 B=1024, u in [-5,5], fp64 (BASELINE configs[1])",
                                                "value": 1024 * T * steps / el4, "unit": "trajectory-timesteps/s",
                                                "ms_per_step": el4 / steps * 1e3, "stages": st4}
        if not args.no_cpu_baseline:
            extra["acrobot_T500_B1024_lim5_f64"]["cpu_baseline"] = cpu_baseline(1024, T, dt, 5.0, target_wall_s=2.0, backward_only=False)
        # the reference's other shipped model, batched (north_star: "acrobot/double-integrator problems"; BASELINE configs[0]
        # is its single-trajectory T=100 solve, a parity case): n=4, m=2 -- the generic m x m box-QP inside the quad kernel
        Td, goal = 100, [1.0, 0.5, 0.0, 0.0]
        for Bd in (4096, 32768):  # one 16-trajectory tile per CU (the quad chain); the saturated regime (k_solve_wide2: 64-trajectory tiles)
            gd = BatchILQR("integrator", Bd, Td, dt, u_min=-0.5, u_max=0.5, goal=goal, device=local_rank, stream=stream,
                           flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=args.warmup + 2 * steps + 1))
            rd = np.random.default_rng(4321)
            gd.init_traj(rd.uniform(-1, 1, size=(Bd, 4)) * np.array([1.5, 1.5, 0.5, 0.5]), np.zeros((Bd, Td, 2)))
            gd.iterate(args.warmup)
            gd.profile(True)
            eld = None
            for _ in range(2):  # (a few ms right after the host-side CPU baseline: the second of two runs is the one with the clocks up)
                gd.profile_reset()
                barrier()
                t0 = time.perf_counter()
                gd.iterate(steps)
                barrier()
                eld = time.perf_counter() - t0
            pd_ = gd.profile_read()
            assert gd.count_running() == Bd
            named = {i: gd.lib.ilqr_stage_kernel_name(gd.h, i).decode() for i in range(capi.NUM_STAGES)}
            solve_kernel = named[capi.STAGE_NAMES.index("solve")]
            key = "integrator_T100_B%d_lim0.5_f64" % Bd
            extra[key] = {
                "workload": "double integrator (n=4, m=2) T=100 B=%d, u in [-0.5,0.5]^2, goal (1, 0.5, 0, 0), fp64, fixed-work iterations" % Bd,
                "value": Bd * Td * steps / eld, "unit": "trajectory-timesteps/s", "ms_per_step": eld / steps * 1e3,
                "stages": {k: {"kernel": solve_kernel if k in ("backward", "rollout", "solve") else named[capi.STAGE_NAMES.index(k)],
                               "ms_per_launch": ms / ln, "launches": ln} for k, (ms, ln) in pd_.items() if ln}}
            # the contract's HBM figure for this line: Rec<4,2> sweep + backward + 11-alpha rollouts, algorithmic bytes per trajectory-timestep
            # (records stay in LDS: xs, us read by the producers 6 doubles; gains written 10; rollouts read 16 rows and write 11 x (2 controls + 4/8 states))
            bts = 8 * (6 + 10 + 16 + 11 * 2.5)
            extra[key]["roofline"] = {"bound": "hbm", "limiter": "valu_issue", "kernel": solve_kernel, "achieved": bts * Bd * Td / (eld / steps) / 1e9,
                                      "peak": 8000.0, "unit": "GB/s", "frac": bts * Bd * Td / (eld / steps) / 1e9 / 8000.0,
                                      "algorithmic_bytes_per_timestep": bts, "traffic": None,
                                      "limiter_note": "one dependent chain per tile (instruction issue and latency), as the acrobot lines"}
            if solve_kernel == "k_solve_wide2" and Bd == 32768:  # the counter passes of this kernel were taken at this batch (collect_profiles.sh: int_*)
                tr, src = pmc_traffic(solve_kernel, steps)
                extra[key]["roofline"]["traffic"] = tr
                extra[key]["roofline"]["traffic_source"] = src
                extra[key]["roofline"]["algorithmic_bytes_per_launch"] = bts * Bd * Td * steps
                iss = issue_roofline(solve_kernel, eld / steps * 1e3, gd.shader_clock_mhz() if pd_.get("solve", (0, 0))[1] else None, Bd * Td)
                extra[key]["roofline"]["bound_frac"] = iss.get("frac")
                extra[key]["roofline_issue"] = iss
            if not args.no_cpu_baseline and Bd == 4096:
                extra[key]["cpu_baseline"] = cpu_baseline_other("integrator", Td, dt)
            gd.close()
        # BASELINE configs[4]: synthetic LQ n=32 m=16 T=200 B=8192, limits +-1: the generic wave-per-trajectory path
        nq, mq, Tq, Bq = 32, 16, 200, 8192
        flop_ts = 4 * nq ** 3 + 10 * nq * nq * mq + 6 * nq * mq * mq + mq ** 3   # backward flops / timestep
        rq = np.random.default_rng(0)
        x0q = rq.uniform(-1, 1, (Bq, nq))
        for label, fl, itq in (("fd", 0, 3), ("analytic", capi.FLAG_ANALYTIC_DERIVATIVES, 6)):
            gq = BatchILQR("lq", Bq, Tq, dt, u_min=-1.0, u_max=1.0, lq=lq_mats(nq, mq), device=local_rank, stream=stream,
                           flags=capi.FLAG_FIXED_WORK | fl, params=dict(max_iter=itq + 2))
            gq.init_traj(x0q, np.zeros((Bq, Tq, mq)))
            gq.iterate(1)
            gq.profile(True)
            gq.profile_reset()
            barrier()
            t0 = time.perf_counter()
            gq.iterate(itq)
            barrier()
            elq = time.perf_counter() - t0
            pq = gq.profile_read()
            assert gq.count_running() == Bq
            names = {i: gq.lib.ilqr_stage_kernel_name(gq.h, i).decode() for i in range(capi.NUM_STAGES)}
            stq = {k: {"kernel": names[capi.STAGE_NAMES.index(k)], "ms_per_launch": ms / ln, "launches": ln} for k, (ms, ln) in pq.items() if ln}
            bw = stq["backward"]["ms_per_launch"] * 1e-3
            gq.close()
            # the finite-difference sweep of this model: one dense model evaluation per perturbed point as the reference
            # performs it (src/derivatives.cpp) would be 2(n+m) Euler maps + 4800 costs of 2(n^2+m^2) flops; the kernel uses
            # the model's separable cost (cost_x(x) + cost_u(u): the parts of a point that did not move are not re-evaluated),
            # which is what it is priced at: (2n + 2n(n+1)) x'Qx + (2m + 2m(m+1)) u'Ru + 2(n+m) Euler maps per knot
            fd_flops_dense_forms = (2 * nq + 2 * nq * (nq + 1)) * 2 * nq * nq + (2 * mq + 2 * mq * (mq + 1)) * 2 * mq * mq + 2 * (nq + mq) * 2 * nq * (nq + mq)
            # ... and k_derivatives_lq (the default since round 5) evaluates a perturbed point's form from what moved: Q p = Q x + delta_i Q[:,i] + delta_j Q[:,j]
            # (4n flops), x . (Q p) (2n), the two perturbed rows (~12); Q x and R u once per knot; the Jacobian sweep stays dense
            # (A x + B u once per knot as well, the Jacobian sweep's points from it: 4n flops each)
            fd_flops = ((2 * nq + 2 * nq * (nq + 1)) * (6 * nq + 12) + (2 * mq + 2 * mq * (mq + 1)) * (6 * mq + 12) + 2 * (nq * nq + mq * mq)
                        + 2 * nq * (nq + mq) + 2 * (nq + mq) * 4 * nq)
            if stq.get("derivatives", {}).get("kernel") != "k_derivatives_lq":
                fd_flops = fd_flops_dense_forms
            fd_dense = 2 * (nq + mq) * 2 * nq * (nq + mq) + (2 * nq + 2 * mq + 2 * nq * (nq + 1) + 2 * mq * (mq + 1) + 4 * nq * mq) * 2 * (nq * nq + mq * mq)
            extra["lq_n32_m16_T200_B8192_" + label] = {
                "workload": "synthetic LQ n=32 m=16 T=200 B=8192, u in [-1,1], fp64, %s derivatives, fixed-work iterations "
                            "(BASELINE configs[4])" % ("finite-difference" if not fl else "exact"),
                "value": Bq * Tq * itq / elq, "unit": "trajectory-timesteps/s", "ms_per_step": elq / itq * 1e3, "stages": stq,
                "roofline": {"bound": "mfma", "kernel": stq["backward"]["kernel"], "achieved": flop_ts * Bq * Tq / bw / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": flop_ts * Bq * Tq / bw / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "algorithmic_flops_per_timestep": flop_ts, "avg_launch_ms": bw * 1e3}}
            if not fl:
                dv = stq["derivatives"]["ms_per_launch"] * 1e-3
                extra["lq_n32_m16_T200_B8192_" + label]["roofline_derivatives"] = {
                    "bound": "fp64 flops (vector and matrix instructions share one fp64 datapath per SIMD: scripts/ubench/coissue.hip)", "kernel": stq["derivatives"]["kernel"],
                    "achieved": fd_flops * Bq * (Tq + 1) / dv / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": fd_flops * Bq * (Tq + 1) / dv / 1e12 / FP64_MFMA_PEAK_TFLOPS, "executed_flops_per_knot": fd_flops,
                    "reference_dense_flops_per_knot": fd_dense, "avg_launch_ms": dv * 1e3,
                    "dense_quadratic_forms_flops_per_knot": fd_flops_dense_forms,
                    "note": "the same sweep evaluated point by point as the reference does would be %.1f x the flops (every point's forms densely on the "
                            "matrix cores, ILQR_ROUTE_LQ_DENSE_FD: %.1f x); the kernel's time goes into VALU work around few flops: the fraction is not its quality measure, "
                            "its speed-up over the dense sweep is (profiles/)" % (fd_dense / fd_flops, fd_flops_dense_forms / fd_flops)}
                if not args.no_cpu_baseline:
                    extra["lq_n32_m16_T200_B8192_" + label]["cpu_baseline"] = cpu_baseline_other("lq", Tq, dt)
            elif not args.no_cpu_baseline:
                extra["lq_n32_m16_T200_B8192_" + label]["cpu_baseline"] = cpu_baseline_lq_exact(Tq, dt)

    if not args.no_extra_configs and world == 1:
        # a user's own device twin with dimensions of its own (examples/user_model_linear6.hpp: n = 6, m = 2, compiled in from outside the
        # library), on both routes such a small twin has
        from ilqr_amd import _build
        if os.path.exists(_build.USER_EXAMPLE6_LIB):
            nu6, mu6, Tu, Bu, itu = 6, 2, 200, 4096, 5
            ru = np.random.default_rng(11)
            A6 = -np.eye(nu6) + 0.3 * ru.normal(size=(nu6, nu6)) / np.sqrt(nu6)
            B6 = ru.normal(size=(nu6, mu6)) / np.sqrt(nu6)
            mats6 = (A6, B6, np.eye(nu6), 0.1 * np.eye(mu6), np.eye(nu6))
            x06 = ru.uniform(-1, 1, (Bu, nu6))

            def run6(route):
                gu = BatchILQR("user", Bu, Tu, dt, u_min=-0.5, u_max=0.5, lib=_build.USER_EXAMPLE6_LIB, nx=nu6, nu=mu6, device=local_rank, stream=stream, route=route,
                               user_params=np.concatenate([np.ascontiguousarray(a).ravel() for a in mats6]), flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=itu + 2))
                gu.init_traj(x06, np.zeros((Bu, Tu, mu6)))
                gu.iterate(1)
                gu.profile(True)
                gu.profile_reset()
                barrier()
                t0 = time.perf_counter()
                gu.iterate(itu)
                barrier()
                elu = time.perf_counter() - t0
                pu_ = gu.profile_read()
                assert gu.count_running() == Bu
                namu = {i: gu.lib.ilqr_stage_kernel_name(gu.h, i).decode() for i in range(capi.NUM_STAGES)}
                stu = {k: {"kernel": namu[capi.STAGE_NAMES.index(k)], "ms_per_launch": ms / ln, "launches": ln} for k, (ms, ln) in pu_.items() if ln}
                gu.close()
                return elu, stu
            flop6 = 4 * nu6 ** 3 + 10 * nu6 * nu6 * mu6 + 6 * nu6 * mu6 * mu6 + mu6 ** 3
            # default route of a small twin (even nx <= 8, nu <= 4): the tiled thread kernels -- a thread per knot / trajectory / rollout
            elu, stu = run6(0)
            extra["user_linear6_n6_m2_T200_B4096_fd"] = {
                "workload": "a user's device twin (examples/user_model_linear6.hpp, n=6 m=2, ILQR_MODEL_USER) T=200 B=4096, u in [-0.5,0.5], fp64, "
                            "finite differences point by point through the model's own functions, fixed-work iterations; the tiled thread kernels "
                            "(k_derivatives, k_backward_t, k_rollout): the default route of a small twin",
                "value": Bu * Tu * itu / elu, "unit": "trajectory-timesteps/s", "ms_per_step": elu / itu * 1e3, "stages": stu,
                "note": "B = 4096 trajectories are 64 wavefronts of k_backward_t on 1024 SIMDs: the backward pass is a latency-bound chain of T steps; the route scales with B up to ~64 K trajectories"}
            # the contract's roofline for the dominant kernel of this line (k_backward_t: records in, gains out; SURVEY 8d's byte count at n = 6, m = 2)
            bt6 = algorithmic_bytes_per_timestep(nu6, mu6, 8)["backward"]
            bw6 = stu["backward"]["ms_per_launch"] * 1e-3
            extra["user_linear6_n6_m2_T200_B4096_fd"]["roofline"] = {
                "bound": "hbm", "kernel": stu["backward"]["kernel"], "achieved": bt6 * Bu * Tu / bw6 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bt6 * Bu * Tu / bw6 / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_timestep": bt6, "algorithmic_bytes_per_launch": bt6 * Bu * Tu, "avg_launch_ms": bw6 * 1e3,
                "traffic": None, "limiter": "latency", "limiter_note": "64 wavefronts, each a dependent chain of T = 200 steps with the 6 x 6 algebra and the 2 x 2 box-QP in one thread: "
                "6 % of the chip's SIMDs have a wavefront at all; the figure rises with B, not with the kernel"}
            # the same workload on the generic kernels every larger twin runs in (ILQR_ROUTE_WAVE_PER_TRAJECTORY)
            elw, stw = run6(capi.ROUTE_WAVE_PER_TRAJECTORY)
            bwu = stw["backward"]["ms_per_launch"] * 1e-3
            extra["user_linear6_n6_m2_T200_B4096_fd_wave_per_trajectory"] = {
                "workload": "the same twin and workload on the generic kernels (ILQR_ROUTE_WAVE_PER_TRAJECTORY): wavefront-per-knot finite differences, k_backward_w3<1> "
                            "(one 16 x 16 tile per matrix), thread-per-rollout forward passes",
                "value": Bu * Tu * itu / elw, "unit": "trajectory-timesteps/s", "ms_per_step": elw / itu * 1e3, "stages": stw,
                "roofline": {"bound": "fp64 flops", "kernel": stw["backward"]["kernel"], "achieved": flop6 * Bu * Tu / bwu / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": flop6 * Bu * Tu / bwu / 1e12 / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_timestep": flop6,
                             "avg_launch_ms": bwu * 1e3,
                             "note": "one wavefront per trajectory and every matrix padded to a 16 x 16 tile: (6/16)^2 of each matrix instruction is this model's"}}
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                cores = os.cpu_count() or 1
                om6 = O.Model("lq", lq=mats6, u_lim=0.5)
                nb6 = min(Bu, 8 * cores)
                x6 = np.random.default_rng(11).uniform(-1, 1, (nb6, nu6))
                t0 = time.perf_counter()
                O.batch_solve(om6, x6, np.zeros((nb6, Tu, mu6)), dt, max_iters=4, fixed_work=True, nthreads=cores)
                t6 = time.perf_counter() - t0
                extra["user_linear6_n6_m2_T200_B4096_fd"]["cpu_baseline"] = {
                    "value": nb6 * Tu * 4 / t6, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
                    "sample": "%d trajectories x 4 fixed-work iterations of the same model as the oracle's LQ model, %.1f s wall, oracle/liboracle_ilqr.so with OpenMP over "
                              "trajectories on all host threads" % (nb6, t6)}

    if not args.no_extra_configs and world == 1:
        # a user's twin that is neither small nor linear-quadratic (examples/user_model_pendulum_chain.hpp: n = 16, m = 4, trigonometric dynamics,
        # a non-quadratic cost): the generic path measured on a model that can take none of the LQ twin's shortcuts -- finite differences point by
        # point through the model's own functions (40 Euler maps + 880 cost evaluations per knot), k_backward_w3<1>, thread-per-rollout forward passes
        from ilqr_amd import _build
        if os.path.exists(_build.USER_CHAIN_LIB):
            NLc, Tc, Bc, itc, limc = 8, 200, 4096, 5, 2.0
            nc, mc = 2 * NLc, NLc // 2
            prm = np.array([9.81, 0.1, 2.0, 10.0, 1.0, 0.1, 50.0, 0.0])
            rc_ = np.random.default_rng(3)
            x0c = np.concatenate([rc_.uniform(-1, 1, (Bc, NLc)), rc_.uniform(-1, 1, (Bc, NLc)) * 0.5], axis=1)
            gc = BatchILQR("user", Bc, Tc, dt, u_min=-limc, u_max=limc, lib=_build.USER_CHAIN_LIB, nx=nc, nu=mc, device=local_rank, stream=stream,
                           user_params=prm, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=itc + 2))
            gc.init_traj(x0c, np.zeros((Bc, Tc, mc)))
            gc.iterate(1)
            gc.profile(True)
            gc.profile_reset()
            barrier()
            t0 = time.perf_counter()
            gc.iterate(itc)
            barrier()
            elc = time.perf_counter() - t0
            pc_ = gc.profile_read()
            assert gc.count_running() == Bc
            namc = {i: gc.lib.ilqr_stage_kernel_name(gc.h, i).decode() for i in range(capi.NUM_STAGES)}
            stc = {k: {"kernel": namc[capi.STAGE_NAMES.index(k)], "ms_per_launch": ms / ln, "launches": ln} for k, (ms, ln) in pc_.items() if ln}
            gc.close()
            flopc = 4 * nc ** 3 + 10 * nc * nc * mc + 6 * nc * mc * mc + mc ** 3
            bwc = stc["backward"]["ms_per_launch"] * 1e-3
            recc = 2 * nc * nc + 2 * nc * mc + mc * mc + nc + mc   # doubles per knot record
            key = "user_pendulum_chain_n16_m4_T200_B4096_fd"
            extra[key] = {
                "workload": "a user's device twin (examples/user_model_pendulum_chain.hpp: 8 coupled pendulums, n=16 m=4, trigonometric dynamics, non-quadratic cost; "
                            "ILQR_MODEL_USER) T=200 B=4096, u in [-2,2], fp64, finite differences point by point through the model's own functions, fixed-work "
                            "iterations: the generic kernels (k_derivatives_g, k_backward_w3<1>, k_rollout_g)",
                "value": Bc * Tc * itc / elc, "unit": "trajectory-timesteps/s", "ms_per_step": elc / itc * 1e3, "stages": stc,
                "roofline": {"bound": "mfma", "kernel": stc["backward"]["kernel"], "achieved": flopc * Bc * Tc / bwc / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": flopc * Bc * Tc / bwc / 1e12 / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_timestep": flopc,
                             "avg_launch_ms": bwc * 1e3, "traffic": None,
                             "note": "one 16 x 16 tile per matrix at n = 16 (m = 4: a quarter of the control tiles' columns is this model's)"},
                "roofline_records": {"bound": "hbm", "what": "the record array between the sweep and the backward pass (%d doubles per knot, written once and read once per iteration)" % recc,
                                     "bytes_per_iteration": 2 * 8 * recc * Bc * (Tc + 1),
                                     "sweep_GBps": 8 * recc * Bc * (Tc + 1) / (stc["derivatives"]["ms_per_launch"] * 1e-3) / 1e9 if "derivatives" in stc else None,
                                     "backward_GBps": 8 * recc * Bc * (Tc + 1) / bwc / 1e9, "peak": HBM_PEAK_GBS}}
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                cores = os.cpu_count() or 1
                omc = O.Model("chain", chain=(NLc, prm), u_lim=limc)
                nbc = min(Bc, 4 * cores)
                O.batch_solve(omc, x0c[:nbc], np.zeros((nbc, Tc, mc)), dt, max_iters=1, fixed_work=True, nthreads=cores)
                t0 = time.perf_counter()
                O.batch_solve(omc, x0c[:nbc], np.zeros((nbc, Tc, mc)), dt, max_iters=4, fixed_work=True, nthreads=cores)
                tcb = time.perf_counter() - t0
                extra[key]["cpu_baseline"] = {"value": nbc * Tc * 4 / tcb, "unit": "trajectory-timesteps/s", "cores": cores, "kind": "port",
                                              "sample": "%d trajectories x 4 fixed-work iterations of the same model (oracle/orc_models.inc: chain_*), %.1f s wall, "
                                                        "oracle/liboracle_ilqr.so with OpenMP over trajectories on all host threads" % (nbc, tcb)}

    if rank == 0:
        costs = gathered.cpu().numpy()
        assert np.all(np.isfinite(costs)), "non-finite cost in the gathered result"
        value = world * B * T * steps / elapsed
        bytes_bw = algorithmic_bytes_per_timestep(n, m, s_bytes)["backward"]
        roof = roofline_of(stages, bytes_ts, B, args.dtype)
        roof_issue = issue_roofline(roof["kernel"] + ("_f32" if args.dtype == "f32" else ""), elapsed / steps * 1e3, headline_sclk, B * T)
        roof["bound_frac"] = roof_issue["frac"]
        out = {
            "metric": "iLQR iterations/sec (batch x T timesteps/sec), acrobot T=500 batch=4096",
            "value": value, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "acrobot n=4 m=1 T=499 (500 knots), box-QP limits active, FD derivatives + backward/box-QP + 11-alpha search + accept, fixed work",
                       "batch_per_gpu": B, "global_batch": world * B, "T": T, "u_limit": lim,
                       "parallelism": "batch shards x%d, one all_gather of costs" % world},
            "roofline": roof,
            "roofline_issue": roof_issue,
            # how many ranks ran and what carried their one collective (the driver's scaling run reads this)
            "rccl_ranks": world, "collective_backend": (dist.get_backend() if world > 1 else "none (one rank)"),
            "stages": stages,
            # north star "backward-pass throughput": k_backward_q alone on fixed derivative records
            "backward_only": {"kernel": "k_backward_q", "ms_per_launch": bw_ms,
                              "timesteps_per_s": B * T / (bw_ms * 1e-3) * world,
                              "algorithmic_GBps": bytes_bw * B * T / (bw_ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": bytes_bw * B * T / (bw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "backward_only_timesteps_per_s": B * T / (bw_ms * 1e-3) * world,
            "final_cost_mean": float(np.mean(costs)),
            "configs": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, T, dt, lim)
        if args.extras_out:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.extras_out)), exist_ok=True)
                with open(args.extras_out, "w") as f:
                    json.dump(out, f, indent=1)
                out["extras"] = os.path.relpath(args.extras_out, ROOT)
            except OSError as e:  # a read-only checkout: the record on stdout is what counts
                print("bench.py: extras not written (%s)" % e, file=sys.stderr)
        print(compact_record(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
