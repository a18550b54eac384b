#!/bin/bash
# Round 5 soaks + the B = 32768 walks:  gpurun --timeout 2400 -- 'bash scripts/gpu_soak_r05.sh'
mkdir -p gpurun_out
python -m pytest "tests/test_gpu_parity.py::test_bench_saturated_batch_two_wide_tiles_per_cu_10_iterations" -x -q -m gpu > gpurun_out/walk32768.txt 2>&1
tail -4 gpurun_out/walk32768.txt
{
  echo "== scripts/soak_lq.py 240 71"; python scripts/soak_lq.py 240 71
  echo "== scripts/soak_lq.py 240 72"; python scripts/soak_lq.py 240 72
  echo "== scripts/soak_lq_iter.py 200 5"; python scripts/soak_lq_iter.py 200 5
  echo "== scripts/soak.py 150 81"; python scripts/soak.py 150 81
} > gpurun_out/r05_soak.txt 2>&1
cat gpurun_out/r05_soak.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
