#!/bin/bash
# A/B several experiment builds of the library on the bench workload: scripts/ab_libs.sh [-r REPS] [-c] lib1.so lib2.so ...
#   -c: also one rocprofv3 counter pass per library (LDS bank conflicts, FETCH_SIZE, WRITE_SIZE of k_solve_hex; --steps 5 --warmup 3)
# (bench.py prints the compact record; the per-stage table comes from its extras file)
REPS=2; PMC=0
while getopts "r:c" o; do case $o in r) REPS=$OPTARG;; c) PMC=1;; esac; done; shift $((OPTIND-1))
ROOT=$(pwd)
for rep in $(seq 1 $REPS); do
for L in "$@"; do
  ILQR_AMD_LIB=$ROOT/$L timeout 200 python bench.py --no-cpu-baseline --extras-out /tmp/ab_extras.json > /tmp/ab.json 2>/tmp/ab.err
  python - "$L" <<PY
import json,sys
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    x=json.load(open("/tmp/ab_extras.json"))
    print(sys.argv[1], "%.4g ts/s"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v["ms_per_launch"],4) for k,v in x["stages"].items()}, "sclk %s" % x["roofline_issue"].get("sclk_mhz_measured"))
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab.err").read()[-600:])
PY
done
done
if [ $PMC = 1 ]; then
  cd /tmp && export TMPDIR=/tmp
  for L in "$@"; do
    for C in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
      rm -rf /tmp/abp
      ILQR_AMD_LIB=$ROOT/$L timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/abp -o x -- python $ROOT/bench.py --no-cpu-baseline --extras-out '' --steps 5 --warmup 3 > /dev/null 2> /tmp/abp.err
      f=$(find /tmp/abp -name "*.db" | head -1)
      echo "== $L  [$C]"
      [ -n "$f" ] && python $ROOT/scripts/prof_summary.py $f 2>&1 | grep -E "k_solve_hex|k_solve_tile|k_solve_wide" 
    done
  done
fi
