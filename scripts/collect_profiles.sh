#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root).
# usage: scripts/collect_profiles.sh rNN
# Kernel-trace + stats in one run; every PMC counter set in a run of its own with --kernel-trace only
# (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -u
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --extras-out $OUT/bench_extras_under_rocprof.json"
SHORT="$BENCH --no-extra-configs --steps 5 --warmup 3"
# the whole default record (headline + configs) under the tracer
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o $R -- $BENCH --extra-configs --steps 20 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# the headline alone, 5 iterations per launch: durations in the units of the counter passes below
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats5 -o $R -- $SHORT > /dev/null 2> $OUT/stats5.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o $R -- $SHORT > /dev/null 2> $OUT/pmc_$C.err
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/pmc_sq1 -o $R -- $SHORT > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o $R -- $SHORT > /dev/null 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq3 -o $R -- $SHORT > /dev/null 2> $OUT/pmc_sq3.err
# the matrix unit in the headline kernel (k_solve_hex: nine v_mfma_f64_4x4x4 per chain step)
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $OUT/pmc_sq4 -o $R -- $SHORT > /dev/null 2> $OUT/pmc_sq4.err
# the same launches on round 3's route (ILQR_ROUTE_QUAD_CHAIN = 256: k_solve_tile<..,1>), for the comparison of the two kernels
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/quad_stats5 -o $R -- $SHORT --route 256 > /dev/null 2> $OUT/quad_stats5.err
# the saturated batch (B = 32768: two 64-trajectory wide tiles per CU, k_solve_wide<.., 2>)
SAT="$SHORT --batch 32768"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/sat_stats5 -o $R -- $SAT > /dev/null 2> $OUT/sat_stats5.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/sat_pmc_$C -o $R -- $SAT > /dev/null 2> $OUT/sat_pmc_$C.err
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/sat_pmc_sq1 -o $R -- $SAT > /dev/null 2> $OUT/sat_pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/sat_pmc_sq2 -o $R -- $SAT > /dev/null 2> $OUT/sat_pmc_sq2.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/sat_pmc_sq3 -o $R -- $SAT > /dev/null 2> $OUT/sat_pmc_sq3.err
# BASELINE configs[3] (fp32, limits +-5): its per-GPU shard (B = 4096: k_solve_hex<float>) and its stated size on one GPU (B = 32768: k_solve_wide<float>)
for P in "f32_:--dtype f32 --limit 5" "f32sat_:--dtype f32 --limit 5 --batch 32768"; do
  PRE=${P%%:*}; ARGS=${P#*:}
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${PRE}stats5 -o $R -- $SHORT $ARGS > /dev/null 2> $OUT/${PRE}stats5.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/${PRE}pmc_$C -o $R -- $SHORT $ARGS > /dev/null 2> $OUT/${PRE}pmc_$C.err
  done
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/${PRE}pmc_sq1 -o $R -- $SHORT $ARGS > /dev/null 2> $OUT/${PRE}pmc_sq1.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/${PRE}pmc_sq2 -o $R -- $SHORT $ARGS > /dev/null 2> $OUT/${PRE}pmc_sq2.err
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/${PRE}pmc_sq3 -o $R -- $SHORT $ARGS > /dev/null 2> $OUT/${PRE}pmc_sq3.err
done
# the per-stage route of the same workload (ILQR_FLAG_STAGED = 32): one launch per phase, for the per-phase traffic
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/staged_pmc_$C -o $R -- $SHORT --flags 32 > /dev/null 2> $OUT/staged_pmc_$C.err
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/staged_stats -o $R -- $SHORT --flags 32 > /dev/null 2> $OUT/staged_stats.err
# the generic path (configs[4]: LQ n=32, m=16, T=200, B=8192, exact derivatives): k_backward_w3 (fused: no sweep, no record array); forced, round 2's
# k_backward_w2 (route 1024); and the finite-difference mode (k_derivatives_g + k_backward_w3 on per-knot records)
LQ="python $ROOT/scripts/bench_lq.py 8192 2 16"
SQLQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/lq_stats -o $R -- $LQ > $OUT/lq_bench.txt 2> $OUT/lq_stats.err
timeout 300 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lq_pmc_sq -o $R -- $LQ > /dev/null 2> $OUT/lq_pmc_sq.err
timeout 300 rocprofv3 --pmc MeanOccupancyPerCU MeanOccupancyPerActiveCU --kernel-trace -d $OUT/lq_pmc_occ -o $R -- $LQ > /dev/null 2> $OUT/lq_pmc_occ.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/lq_pmc_$C -o $R -- $LQ > /dev/null 2> $OUT/lq_pmc_$C.err
done
timeout 300 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lq_w2_pmc_sq -o $R -- $LQ 1024 > /dev/null 2> $OUT/lq_w2_pmc_sq.err
LQFD="python $ROOT/scripts/bench_lq.py 8192 1 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/lqfd_stats -o $R -- $LQFD > $OUT/lqfd_bench.txt 2> $OUT/lqfd_stats.err
timeout 600 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lqfd_pmc_sq -o $R -- $LQFD > /dev/null 2> $OUT/lqfd_pmc_sq.err
timeout 600 rocprofv3 --pmc MeanOccupancyPerCU MeanOccupancyPerActiveCU --kernel-trace -d $OUT/lqfd_pmc_occ -o $R -- $LQFD > /dev/null 2> $OUT/lqfd_pmc_occ.err
# ... and with every perturbed point's forms evaluated densely on the matrix cores (ILQR_ROUTE_LQ_DENSE_FD = 2048: k_derivatives_g, the sweep of rounds 1-4)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/lqfd_dense_stats -o $R -- $LQFD 2048 > $OUT/lqfd_dense_bench.txt 2> $OUT/lqfd_dense_stats.err
timeout 600 rocprofv3 --pmc $SQLQ --kernel-trace -d $OUT/lqfd_dense_pmc_sq -o $R -- $LQFD 2048 > /dev/null 2> $OUT/lqfd_dense_pmc_sq.err
# the double integrator (m = 2) in the saturated regime: k_solve_wide2 (64-trajectory tiles, thread-per-trajectory chain with the 2 x 2 box-QP)
INT="python $ROOT/scripts/bench_integrator.py"
B=32768 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/int_stats -o $R -- $INT > $OUT/int_bench.txt 2> $OUT/int_stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  B=32768 timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/int_pmc_$C -o $R -- $INT > /dev/null 2> $OUT/int_pmc_$C.err
done
B=32768 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/int_pmc_sq1 -o $R -- $INT > /dev/null 2> $OUT/int_pmc_sq1.err
cd $ROOT
for U in lat ldsmix; do  # microbenchmarks quoted in DESIGN.md, re-run on this box
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/$U scripts/ubench/$U.hip 2> /dev/null && $OUT/$U > $OUT/ubench_$U.txt 2>/dev/null
  rm -f $OUT/$U
done
for d in stats stats5 quad_stats5 pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq1 pmc_sq2 pmc_sq3 pmc_sq4 sat_stats5 sat_pmc_FETCH_SIZE sat_pmc_WRITE_SIZE sat_pmc_sq1 sat_pmc_sq2 sat_pmc_sq3 \
         f32_stats5 f32_pmc_FETCH_SIZE f32_pmc_WRITE_SIZE f32_pmc_sq1 f32_pmc_sq2 f32_pmc_sq3 f32sat_stats5 f32sat_pmc_FETCH_SIZE f32sat_pmc_WRITE_SIZE f32sat_pmc_sq1 f32sat_pmc_sq2 f32sat_pmc_sq3 \
         staged_pmc_FETCH_SIZE staged_pmc_WRITE_SIZE staged_stats lq_stats lq_pmc_sq lq_pmc_occ lq_pmc_FETCH_SIZE lq_pmc_WRITE_SIZE lq_w2_pmc_sq lqfd_stats lqfd_pmc_sq lqfd_pmc_occ lqfd_dense_stats lqfd_dense_pmc_sq int_stats int_pmc_FETCH_SIZE int_pmc_WRITE_SIZE int_pmc_sq1; do
  f=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f > $OUT/$d.txt 2>&1
done
python scripts/make_traffic.py $OUT $R > $OUT/traffic.json 2> $OUT/traffic.err
cp $OUT/traffic.json profiles/traffic.json   # (on the GPU box only; the caller copies $OUT/traffic.json back)
python bench.py --extras-out $OUT/bench_default_extras.json > $OUT/bench_default.json 2> $OUT/bench_default.err   # the default record, now quoting counters of THIS code
timeout 1200 python bench.py --extra-configs --extras-out $OUT/bench_extra_configs.json > $OUT/bench_extra_configs_line.json 2> $OUT/bench_extra_configs.err   # the other configurations, each with its CPU baseline
find $OUT -name "*.db" -delete
ls -la $OUT
