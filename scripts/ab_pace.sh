#!/bin/bash
# A/B of an experiment build against the stock library on the headline workload: per-phase times (alternating runs) and the
# FETCH_SIZE / WRITE_SIZE passes of both.   gpurun --timeout 900 -- 'bash scripts/ab_pace.sh ilqr_amd/lib/libilqr_amd_pace.so'
ROOT=$(pwd)
mkdir -p gpurun_out
bash scripts/hex_sections.sh "$@" > /dev/null 2>&1
cp gpurun_out/hex_sections.txt gpurun_out/ab_pace_times.txt
cd /tmp && export TMPDIR=/tmp
for L in ilqr_amd/lib/libilqr_amd.so "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abp
    ILQR_AMD_LIB=$ROOT/$L timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/abp -o x -- python $ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 3 > /dev/null 2> /tmp/abp.err
    f=$(find /tmp/abp -name "*.db" | head -1)
    echo "== $L $C"
    [ -n "$f" ] && python $ROOT/scripts/prof_summary.py $f | grep -E "k_solve_hex|kernel|name" | head -6
  done
done > $ROOT/gpurun_out/ab_pace_pmc.txt 2>&1
cd $ROOT
cat gpurun_out/ab_pace_times.txt gpurun_out/ab_pace_pmc.txt
