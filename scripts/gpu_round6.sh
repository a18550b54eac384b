#!/bin/bash
# Round 6, one gpurun call on the final sources: soaks with seeds of their own (every nx = 4 route bit-identical + oracle; the matrix-core chains
# against the quad chain on long horizons; the generic kernels against the oracle and k_backward_w3 against k_backward_w2; whole generic
# iterations; REGULARIZE_VXX), two 100-iteration device-driven walks, the batch sweep, whole-solve times.
#   gpurun --timeout 3000 -- 'bash scripts/gpu_round6.sh r06c'
R=${1:-r06c}
mkdir -p gpurun_out
{
  echo "== scripts/soak.py 200 111"; python scripts/soak.py 200 111
  echo "== scripts/soak_hex.py 300 21"; python scripts/soak_hex.py 300 21
  echo "== scripts/soak_lq.py 200 81"; python scripts/soak_lq.py 200 81
  echo "== scripts/soak_lq_iter.py 150 15"; python scripts/soak_lq_iter.py 150 15
  echo "== scripts/soak_regv.py 100 5"; python scripts/soak_regv.py 100 5
  echo "== scripts/long_walk.py 64 100 1.5 1.0"; python scripts/long_walk.py 64 100 1.5 1.0
  echo "== scripts/long_walk.py 64 100 5.0 0.01"; python scripts/long_walk.py 64 100 5.0 0.01
} > gpurun_out/${R}_soak.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_soak.txt | tail -30
BATCHES="1024 2048 4096 8192 16384 32768" bash scripts/batch_sweep.sh $R > /dev/null 2>&1
cp gpurun_out/sweep_$R.txt gpurun_out/${R}_batch_sweep.txt; cat gpurun_out/${R}_batch_sweep.txt
python scripts/full_solve_time.py > gpurun_out/${R}_full_solve.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_full_solve.txt | tail -4
