#!/bin/bash
# Where phase 2 of k_solve_hex goes: the stock library beside an experiment build with -DILQR_HEX_SECTIONS, whose "backward" clock is
# rollout_tile (rollouts + accept) and whose "rollout" clock is what follows it (commit of the accepted candidates + barriers).
#   hipcc <FLAGS of ilqr_amd/_build.py> -DILQR_HEX_SECTIONS -o ilqr_amd/lib/libilqr_amd_hexsec.so ilqr_amd/csrc/capi.hip
#   gpurun --timeout 600 -- 'bash scripts/hex_sections.sh'
mkdir -p gpurun_out
for L in ilqr_amd/lib/libilqr_amd.so "$@"; do
  for rep in 1 2; do
    ILQR_AMD_LIB=$PWD/$L timeout 200 python bench.py --no-cpu-baseline --no-extra-configs > /tmp/ab.json 2>/tmp/ab.err
    python - "$L" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v["ms_per_launch"],4) for k,v in d["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
  done
done 2>&1 | tee gpurun_out/hex_sections.txt
