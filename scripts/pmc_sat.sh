#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sat; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SAT="python $ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 3 --batch 32768"
i=0
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/s$i -o s -- $SAT > /dev/null 2> $OUT/s$i.err
  f=$(find $OUT/s$i -name "*.db" | head -1)
  [ -n "$f" ] && python $ROOT/scripts/prof_summary.py $f > $OUT/s$i.txt 2>&1
  find $OUT/s$i -name "*.db" -delete
done
cd $ROOT; grep -h "k_solve_wide" -A8 $OUT/s*.txt | head -120
