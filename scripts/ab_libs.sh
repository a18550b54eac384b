#!/bin/bash
# A/B several experiment builds of the library on the bench workload: scripts/ab_libs.sh lib1.so lib2.so ...
for L in "$@"; do
  ILQR_AMD_LIB=$L timeout 200 python bench.py --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
  python - "$L" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v["ms_per_launch"],4) for k,v in d["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
done
