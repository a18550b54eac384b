ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_b8192; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o r01e -- python $ROOT/bench.py --no-cpu-baseline --batch 8192 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT; f=$(find $OUT/stats -name "*.db" | head -1); python scripts/prof_summary.py $f > $OUT/stats.txt 2>&1; head -8 $OUT/stats.txt | cut -c1-170
