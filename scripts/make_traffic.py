"""profiles/traffic.json from the rocprofv3 pass summaries of scripts/collect_profiles.sh (scripts/prof_summary.py text).

    python scripts/make_traffic.py gpurun_out/prof_r02a r02a > profiles/traffic.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a coalesced stream), WRITE_SIZE is
taken as is; both are in KB = 1024 B.  The persistent kernel k_solve_tile runs a different number of iterations
per launch (warm-up 3, timed 5 in the counter runs), so its figures are normalised PER ITERATION; bench.py scales
them to the launch it times."""
import json
import re
import sys

WARMUP, STEPS = 3, 5   # iterations of the two k_solve_tile launches of the counter runs (collect_profiles.sh)


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(\w+)(<.*?>)?\(.*?\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            name, targs, n, avg = m.group(1), m.group(2) or "", int(m.group(3)), float(m.group(4))
            if name == "k_rollout" and "false, false" in targs:
                name = "k_rollout_init"
            if "float" in targs:
                name += "_f32"
            out[name] = (avg, n)
    return out


def durations(path):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(\w+)(<.*?>)?\(.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and m.group(1).startswith("k_"):
            name = m.group(1) + ("_f32" if "float" in (m.group(2) or "") else "")
            if name == "k_rollout" and "false, false" in (m.group(2) or ""):
                name = "k_rollout_init"
            out[name] = dict(calls=int(m.group(3)), total_us=float(m.group(4)), avg_us=float(m.group(5)))
    return out


def main(d, tag):
    f, w = read("%s/pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE"), read("%s/pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE")
    f.update({k: v for k, v in read("%s/staged_pmc_FETCH_SIZE.txt" % d, "FETCH_SIZE").items() if k not in f})
    w.update({k: v for k, v in read("%s/staged_pmc_WRITE_SIZE.txt" % d, "WRITE_SIZE").items() if k not in w})
    valu = read("%s/pmc_sq1.txt" % d, "SQ_INSTS_VALU")
    wavecyc = read("%s/pmc_sq3.txt" % d, "SQ_WAVE_CYCLES")
    waitany = read("%s/pmc_sq3.txt" % d, "SQ_WAIT_ANY")
    bank = read("%s/pmc_sq2.txt" % d, "SQ_LDS_BANK_CONFLICT")
    ldsact = read("%s/pmc_sq2.txt" % d, "SQ_LDS_IDX_ACTIVE")
    dur = durations("%s/stats5.txt" % d)
    dur.update({k: v for k, v in durations("%s/staged_stats.txt" % d).items() if k not in dur})
    kernels = {}
    B, T = 4096, 499
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fr, wr = f.get(k, (0.0, 1))[0], w.get(k, (0.0, 1))[0]
        e = {"fetch_size_kb": fr, "write_size_kb": wr, "hbm_read_bytes": 2 * fr * 1024, "hbm_write_bytes": wr * 1024,
             "hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024}
        if k == "k_solve_tile":   # two launches (WARMUP and STEPS iterations): the averages are per (WARMUP+STEPS)/2 iterations
            its = (WARMUP + STEPS) / 2.0
            e["iterations_per_average_launch"] = its
            e["hbm_bytes_per_iteration"] = e["hbm_bytes_per_launch"] / its
            if k in valu and k in dur:
                e["valu_insts_per_iteration"] = valu[k][0] / its
                e["avg_iteration_us"] = dur[k]["total_us"] / (WARMUP + STEPS)
                e["busy_simds"] = 1024            # 256 blocks x 4 wavefronts, one per SIMD
                e["sclk_hz"] = 2.4e9
                e["timesteps_per_iteration"] = B * T
                e["wait_fraction_of_wave_cycles"] = waitany[k][0] / wavecyc[k][0] if k in waitany and k in wavecyc else None
                e["lds_bank_conflict_fraction"] = bank[k][0] / ldsact[k][0] if k in bank and k in ldsact and ldsact[k][0] else None
        elif k in dur:
            e["avg_launch_us"] = dur[k]["avg_us"]
        kernels[k] = e
    json.dump({"source": "rocprofv3 passes of `bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 3` on MI355X, one counter "
                         "set per run with --kernel-trace only (scripts/collect_profiles.sh; summaries profiles/%s_*.txt): FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 reports half of a coalesced stream), WRITE_SIZE as is, KB = 1024 B; k_sweep_backward / "
                         "k_rollout rows from the same workload launched per stage (--flags 32)" % tag,
               "workload": "acrobot T=499 B=4096 fp64 limits +-1.5", "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
