#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts in this path's access shapes (scripts/ubench/fetchcal.hip).
# usage (GPU box, repo root): scripts/fetch_calibration.sh OUTDIR   -> OUTDIR/fetch_calibration.txt
set -u
OUT=${1:-gpurun_out/cal}
ROOT=$(pwd)
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/fetchcal scripts/ubench/fetchcal.hip 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $ROOT/$OUT/cal_$C -o cal -- $ROOT/$OUT/fetchcal > /dev/null 2> $ROOT/$OUT/cal_$C.err
done
cd $ROOT
python - $OUT <<'PY' > $OUT/fetch_calibration.txt
import glob, sqlite3, sys
out = sys.argv[1]
known = 1 << 30
print("FETCH_SIZE / WRITE_SIZE (KB = 1024 B) of kernels that move exactly %d bytes each (scripts/ubench/fetchcal.hip), MI355X:" % known)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/cal_%s/**/*.db" % (out, c), recursive=True):
        con = sqlite3.connect(f)
        for k, cn, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            if cn == c and (("read" in k) == (c == "FETCH_SIZE")):
                print("%-14s %-12s %14.0f KB  -> true bytes / counter bytes = %.3f" % (k.split("(")[0], cn, v, known / (v * 1024.0)))
PY
cat $OUT/fetch_calibration.txt
find $OUT -name "*.db" -delete; rm -f $OUT/fetchcal
