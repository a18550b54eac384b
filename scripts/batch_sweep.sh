#!/bin/bash
# bench.py at other batch sizes per GPU: run through gpurun from the repo root
for b in 1024 2048 8192 16384 65536; do
  python bench.py --no-cpu-baseline --batch $b 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=%d: %.3e trajectory-timesteps/s, %.3f ms per iteration, backward stage kernel %s' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['kernel']))"
done
