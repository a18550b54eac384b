#!/bin/bash
# bench.py at several per-GPU batch sizes (headline workload otherwise): one line per size into gpurun_out/sweep_$1.txt
# (the compact record on stdout, the per-phase clocks and the kernel name from the extras file)
out=gpurun_out/sweep_${1:-x}.txt
: > $out
for b in ${BATCHES:-1024 4096 8192 16384 32768}; do
  python bench.py --batch $b --no-cpu-baseline --extras-out /tmp/sweep_extras.json ${EXTRA} 2>>gpurun_out/sweep_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = json.load(open('/tmp/sweep_extras.json'))['stages']
print('B', d['config']['batch_per_gpu'], 'ms/iter %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'phase1 %.4f phase2 %.4f' % (s.get('backward', {}).get('ms_per_launch', 0), s.get('rollout', {}).get('ms_per_launch', 0)), 'kernel', s.get('solve', s.get('backward'))['kernel'], 'hbm frac %.3f' % d['roofline']['frac'])
" >> $out
done
cat $out
