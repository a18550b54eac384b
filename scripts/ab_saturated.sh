#!/bin/bash
# A/B experiment builds of the library on the saturated lines of the bench (B = 32768, fp64 +-1.5 and fp32 +-5):
#   scripts/ab_saturated.sh lib1.so lib2.so ...      (alternating, two rounds)
mkdir -p gpurun_out
for rep in 1 2; do
for L in "$@"; do
  ILQR_AMD_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --extra-configs --extras-out /tmp/ab_extras.json > /tmp/ab.json 2>/tmp/ab.err
  python - "$L" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "headline %.4f ms"%d["ms_per_step"])
    for k in ("saturated", "acrobot_T500_B32768_lim5_f32_one_gpu"):
        v=json.load(open("/tmp/ab_extras.json"))["configs"][k]
        print("   ", k, "%.4g ts/s"%v["value"], "%.4f ms"%v["ms_per_step"], {s:round(x["ms_per_launch"],4) for s,x in v["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
done
done 2>&1 | tee gpurun_out/ab_saturated.txt
