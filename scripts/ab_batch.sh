#!/bin/bash
# bench at several per-GPU batch sizes: scripts/ab_batch.sh 4096 8192 ...   (env passes through)
for B in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --extras-out /tmp/ab_extras.json --batch $B > /tmp/ab.json 2>/tmp/ab.err
  python - "$B" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    x=json.load(open("/tmp/ab_extras.json"))  # (the per-stage table lives in the extras file since round 6)
    print("B=%s"%sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], d["roofline"]["kernel"], {k:round(v["ms_per_launch"],4) for k,v in x["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
done
