#!/bin/bash
# rocprofv3 evidence for the LQ path (configs[4]: n=32, m=16, T=200, B=8192), both derivative modes.
# usage (through gpurun, from the repo root): scripts/collect_profiles_lq.sh rNN
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lq_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FD="python $ROOT/scripts/bench_lq.py 8192 3"
EX="python $ROOT/scripts/bench_lq.py 8192 3 16"
rocprofv3 --kernel-trace --stats -d $OUT/stats_fd -o $R -- $FD > $OUT/bench_lq_fd.txt 2> $OUT/stats_fd.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_exact -o $R -- $EX > $OUT/bench_lq_exact.txt 2> $OUT/stats_exact.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o $R -- $EX > /dev/null 2> $OUT/pmc_$C.err
done
cd $ROOT
for d in stats_fd stats_exact pmc_FETCH_SIZE pmc_WRITE_SIZE; do
  f=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > $OUT/$d.txt 2>&1
done
ls -la $OUT
