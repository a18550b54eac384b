"""profiles/traffic.json from the two PMC pass summaries (scripts/prof_summary.py output).

    python scripts/make_traffic.py profiles/r01b_pmc_fetch_size.txt profiles/r01b_pmc_write_size.txt > profiles/traffic.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a coalesced
stream), WRITE_SIZE is taken as is; both are in KB = 1024 B."""
import json
import re
import sys


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(\w+)(<.*?>)?\(.*?\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            name, targs, n, avg = m.group(1), m.group(2) or "", int(m.group(3)), float(m.group(4))
            if name == "k_rollout" and "false, false" in targs:
                name = "k_rollout_init"
            out[name] = avg
    return out


def main(fetch_path, write_path):
    f, w = read(fetch_path, "FETCH_SIZE"), read(write_path, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fr, wr = f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"fetch_size_kb": fr, "write_size_kb": wr, "hbm_read_bytes": 2 * fr * 1024,
                      "hbm_write_bytes": wr * 1024, "hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 5 --warmup 3` on MI355X "
                         "(%s, %s); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced stream), "
                         "WRITE_SIZE taken as is; KB = 1024 B; averages over the launches of the run" % (fetch_path, write_path),
               "workload": "acrobot T=499 B=4096 fp64 limits +-1.5", "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:3])
