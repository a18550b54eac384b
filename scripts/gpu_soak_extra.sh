#!/bin/bash
# More soak evidence on the final sources with seeds of their own:  gpurun --timeout 2400 -- 'bash scripts/gpu_soak_extra.sh r05d'
R=${1:-r05d}
mkdir -p gpurun_out
{
  echo "== scripts/soak_hex.py 300 11"; python scripts/soak_hex.py 300 11
  echo "== scripts/soak_hex.py 300 12"; python scripts/soak_hex.py 300 12
  echo "== scripts/soak.py 240 91"; python scripts/soak.py 240 91
  echo "== scripts/soak_lq_iter.py 200 9"; python scripts/soak_lq_iter.py 200 9
  echo "== scripts/long_walk.py 64 100 1.5 1.0"; python scripts/long_walk.py 64 100 1.5 1.0
  echo "== scripts/long_walk.py 64 100 5.0 0.01"; python scripts/long_walk.py 64 100 5.0 0.01
} > gpurun_out/${R}_soak2.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_soak2.txt | tail -30
