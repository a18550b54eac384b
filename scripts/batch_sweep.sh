#!/bin/bash
# bench.py at several per-GPU batch sizes (headline workload otherwise): one JSON line per size into gpurun_out/sweep_$1.txt
out=gpurun_out/sweep_${1:-x}.txt
: > $out
for b in ${BATCHES:-1024 4096 8192 16384 32768}; do
  python bench.py --batch $b --no-extra-configs --no-cpu-baseline ${EXTRA} 2>>gpurun_out/sweep_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stages']
print('B', d['config']['batch_per_gpu'], 'ms/iter %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'phase1 %.4f phase2 %.4f' % (s.get('backward', {}).get('ms_per_launch', 0), s.get('rollout', {}).get('ms_per_launch', 0)), 'kernel', s.get('solve', s.get('backward'))['kernel'])
" >> $out
done
cat $out
