#!/bin/bash
# Where phase 2 of k_solve_hex goes: the stock library beside experiment builds with -DILQR_HEX_SECTIONS=n, whose "backward" clock runs
# from the end of phase 1 to mark n (1: rollouts + accept done, 2: + the barrier behind them, 3: + this wavefront's commit of the accepted
# candidates) and whose "rollout" clock is the rest of phase 2.
#   for n in 1 2 3: hipcc <FLAGS of ilqr_amd/_build.py> -DILQR_HEX_SECTIONS=$n -o ilqr_amd/lib/libilqr_amd_hexsec$n.so ilqr_amd/csrc/capi.hip
#   gpurun --timeout 600 -- 'bash scripts/hex_sections.sh ilqr_amd/lib/libilqr_amd_hexsec1.so ...'
# Result (r05d): rollouts + accept 178 us (499 steps x 140 instructions x 5.13 cycles at 2.19 GHz = 164), barrier 1, commit 17.5, barrier 5.
mkdir -p gpurun_out
for L in ilqr_amd/lib/libilqr_amd.so "$@"; do
  for rep in 1 2; do
    ILQR_AMD_LIB=$PWD/$L timeout 200 python bench.py --no-cpu-baseline --extras-out /tmp/ab_extras.json > /tmp/ab.json 2>/tmp/ab.err
    python - "$L" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    x=json.load(open("/tmp/ab_extras.json"))
    print(sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v["ms_per_launch"],4) for k,v in x["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
  done
done 2>&1 | tee gpurun_out/hex_sections.txt
