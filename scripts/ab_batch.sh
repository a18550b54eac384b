#!/bin/bash
# bench at several per-GPU batch sizes: scripts/ab_batch.sh 4096 8192 ...   (env passes through)
for B in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --batch $B > /tmp/ab.json 2>/tmp/ab.err
  python - "$B" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print("B=%s"%sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], d["roofline"]["kernel"], {k:round(v["ms_per_launch"],4) for k,v in d["stages"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
done
