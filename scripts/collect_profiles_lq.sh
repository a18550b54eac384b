#!/bin/bash
# The generic path's counter passes alone (the LQ block of scripts/collect_profiles.sh): configs[4] with exact derivatives on
# k_backward_w3 (default), k_backward_w2 (route 1024), and the finite-difference mode.   usage: scripts/collect_profiles_lq.sh rNN
set -u
R=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lq_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LQ="python $ROOT/scripts/bench_lq.py 8192 2 16"
SQLQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/lq_stats -o $R -- $LQ > $OUT/lq_bench.txt 2> $OUT/lq_stats.err
timeout 300 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lq_pmc_sq -o $R -- $LQ > /dev/null 2> $OUT/lq_pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/lq_pmc_sq2 -o $R -- $LQ > /dev/null 2> $OUT/lq_pmc_sq2.err
timeout 300 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lq_w2_pmc_sq -o $R -- $LQ 1024 > /dev/null 2> $OUT/lq_w2_pmc_sq.err
cd $ROOT
for d in lq_stats lq_pmc_sq lq_pmc_sq2 lq_w2_pmc_sq; do
  f=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > $OUT/$d.txt 2>&1
done
find $OUT -name "*.db" -delete
cat $OUT/lq_bench.txt; grep -h "k_backward_w\|k_rollout_lq" $OUT/lq_stats.txt $OUT/lq_pmc_sq.txt $OUT/lq_pmc_sq2.txt $OUT/lq_w2_pmc_sq.txt
