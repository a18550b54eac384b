"""profiles/traffic.json from the rocprofv3 pass summaries of scripts/collect_profiles.sh (scripts/prof_summary.py text).

    python scripts/make_traffic.py gpurun_out/prof_r02a r02a > profiles/traffic.json

FETCH_SIZE is doubled, WRITE_SIZE is taken as is; both are in KB = 1024 B.  The factors are calibrated on THIS path's access
shapes (scripts/fetch_calibration.sh, profiles/r04_fetch_calibration.txt: 8-byte-per-lane loads of 128-byte tile rows, 8- and
16-byte streams: true bytes / FETCH_SIZE bytes = 2.000; stores 1.000), not only on the guide's 16 B/lane stream.  The persistent kernel k_solve_tile runs a different number of iterations
per launch (warm-up 3, timed 5 in the counter runs), so its figures are normalised PER ITERATION; bench.py scales
them to the launch it times."""
import json
import re
import sys

WARMUP, STEPS = 3, 5   # iterations of the two persistent-kernel launches of the counter runs (collect_profiles.sh)


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(\w+)(<.*?>)?\(.*?\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            name, targs, n, avg = m.group(1), m.group(2) or "", int(m.group(3)), float(m.group(4))
            if name == "k_rollout" and "false, false" in targs:
                name = "k_rollout_init"
            if name == "k_solve_tile" and targs.rstrip().endswith(", 2>"):
                name = "k_solve_tile<2>"
            if name == "k_solve_wide":
                name = "k_solve_wide"
            if "float" in targs:
                name += "_f32"
            out[name] = (avg, n)
    return out


def durations(path):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(\w+)(<.*?>)?\(.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and m.group(1).startswith("k_"):
            name = m.group(1)
            if name == "k_solve_tile" and (m.group(2) or "").rstrip().endswith(", 2>"):
                name = "k_solve_tile<2>"
            name += "_f32" if "float" in (m.group(2) or "") else ""
            if name == "k_rollout" and "false, false" in (m.group(2) or ""):
                name = "k_rollout_init"
            out[name] = dict(calls=int(m.group(3)), total_us=float(m.group(4)), avg_us=float(m.group(5)))
    return out


def solve_entry(d, prefix, key, B, T, f, w, WARMUP=WARMUP, STEPS=STEPS, stats="stats5"):
    """the persistent kernel of one counter-run set (prefix "" = headline batch, "sat_" = saturated batch): per-iteration figures"""
    import os
    valu = read("%s/%spmc_sq1.txt" % (d, prefix), "SQ_INSTS_VALU")
    have_sq3 = os.path.exists("%s/%spmc_sq3.txt" % (d, prefix))
    wavecyc = read("%s/%spmc_sq3.txt" % (d, prefix), "SQ_WAVE_CYCLES") if have_sq3 else {}
    waitany = read("%s/%spmc_sq3.txt" % (d, prefix), "SQ_WAIT_ANY") if have_sq3 else {}
    have_sq2 = os.path.exists("%s/%spmc_sq2.txt" % (d, prefix))
    bank = read("%s/%spmc_sq2.txt" % (d, prefix), "SQ_LDS_BANK_CONFLICT") if have_sq2 else {}
    ldsact = read("%s/%spmc_sq2.txt" % (d, prefix), "SQ_LDS_IDX_ACTIVE") if have_sq2 else {}
    dur = durations("%s/%s%s.txt" % (d, prefix, stats))
    fr, wr = f.get(key, (0.0, 1))[0], w.get(key, (0.0, 1))[0]
    e = {"fetch_size_kb": fr, "write_size_kb": wr, "hbm_read_bytes": 2 * fr * 1024, "hbm_write_bytes": wr * 1024,
         "hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024}
    its = (WARMUP + STEPS) / 2.0  # two launches (WARMUP and STEPS iterations): the averages are per (WARMUP+STEPS)/2 iterations
    e["iterations_per_average_launch"] = its
    e["hbm_bytes_per_iteration"] = e["hbm_bytes_per_launch"] / its
    e["batch"] = B
    if key in valu and key in dur:
        e["valu_insts_per_iteration"] = valu[key][0] / its
        e["avg_iteration_us"] = dur[key]["total_us"] / (WARMUP + STEPS)
        e["timesteps_per_iteration"] = B * T
        e["wait_fraction_of_wave_cycles"] = waitany[key][0] / wavecyc[key][0] if key in waitany and key in wavecyc else None
        e["lds_bank_conflict_fraction"] = bank[key][0] / ldsact[key][0] if key in bank and key in ldsact and ldsact[key][0] else None
        if os.path.exists("%s/%spmc_sq4.txt" % (d, prefix)):  # the matrix unit (k_solve_hex: nine v_mfma_f64_4x4x4 per chain step)
            mf = read("%s/%spmc_sq4.txt" % (d, prefix), "SQ_INSTS_MFMA")
            if key in mf:
                e["mfma_insts_per_iteration"] = mf[key][0] / its
    return e


def main(d, tag):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ilqr_amd import _build
    f, w = read("%s/pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE"), read("%s/pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE")
    f.update({k: v for k, v in read("%s/staged_pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE").items() if k not in f})
    w.update({k: v for k, v in read("%s/staged_pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE").items() if k not in w})
    dur = durations("%s/stats5.txt" % d)
    dur.update({k: v for k, v in durations("%s/staged_stats.txt" % d).items() if k not in dur})
    kernels = {}
    B, T = 4096, 499
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        if k in ("k_solve_tile", "k_solve_hex"):  # the headline batch: the matrix-core chains (the quad chain under --route 256)
            kernels[k] = solve_entry(d, "", k, B, T, f, w)
            continue
        fr, wr = f.get(k, (0.0, 1))[0], w.get(k, (0.0, 1))[0]
        e = {"fetch_size_kb": fr, "write_size_kb": wr, "hbm_read_bytes": 2 * fr * 1024, "hbm_write_bytes": wr * 1024,
             "hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024}
        if k in dur:
            e["avg_launch_us"] = dur[k]["avg_us"]
        kernels[k] = e
    if os.path.exists("%s/sat_pmc_FETCH_SIZE.txt" % d):  # the saturated batch (B = 32768: k_solve_tile<2>)
        fs, ws = read("%s/sat_pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE"), read("%s/sat_pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE")
        for key in ("k_solve_wide", "k_solve_tile<2>"):  # what the saturated batch runs: wide tiles (nu = 1), else two tiles per CU
            if key in fs or key in ws:
                kernels[key] = solve_entry(d, "sat_", key, 32768, T, fs, ws)
    # BASELINE configs[3] (fp32, limits +-5): the per-GPU shard and the stated size on one GPU, each with passes of its own
    for prefix, key, Bk in (("f32_", "k_solve_hex_f32", 4096), ("f32sat_", "k_solve_wide_f32", 32768)):
        if os.path.exists("%s/%spmc_FETCH_SIZE.txt" % (d, prefix)):
            ff, wf = read("%s/%spmc_FETCH_SIZE.txt" % (d, prefix), "FETCH_SIZE"), read("%s/%spmc_WRITE_SIZE.txt" % (d, prefix), "WRITE_SIZE")
            if key in ff or key in wf:
                kernels[key] = solve_entry(d, prefix, key, Bk, T, ff, wf)
                kernels[key]["workload"] = "acrobot T=499 B=%d fp32 limits +-5" % Bk
    # the double integrator (m = 2) in the saturated regime: scripts/bench_integrator.py at B = 32768 (3 + 20 iterations, T = 100)
    if os.path.exists("%s/int_pmc_FETCH_SIZE.txt" % d):
        ff, wf = read("%s/int_pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE"), read("%s/int_pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE")
        for key in ("k_solve_wide2", "k_solve_wide2_f32"):
            if key in ff or key in wf:
                kernels[key] = solve_entry(d, "int_", key, 32768, 100, ff, wf, WARMUP=3, STEPS=20, stats="stats")
                kernels[key]["workload"] = "double integrator T=100 B=32768 %s limits +-0.5 (scripts/bench_integrator.py)" % ("fp32" if key.endswith("f32") else "fp64")
    json.dump({"source": "rocprofv3 passes of `bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 3` on MI355X, one counter "
                         "set per run with --kernel-trace only (scripts/collect_profiles.sh; summaries profiles/%s_*.txt): FETCH_SIZE x 2.000, WRITE_SIZE x 1.000 "
                         "as calibrated on 8-byte row loads / stores of known size (profiles/r04_fetch_calibration.txt), KB = 1024 B; k_sweep_backward / "
                         "k_rollout rows from the same workload launched per stage (--flags 32); k_solve_wide (the saturated batch) from the same command with "
                         "--batch 32768" % tag,
               "source_hash": _build._source_hash(),
               "workload": "acrobot T=499 B=4096 fp64 limits +-1.5", "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
