#!/bin/bash
# The generic path's GPU tests + stage times of configs[4].   gpurun --timeout 1800 -- 'bash scripts/gpu_generic_check.sh'
mkdir -p gpurun_out
python -m pytest tests/test_gpu_generic_backward.py tests/test_gpu_lq_end_to_end.py tests/test_gpu_analytic.py tests/test_gpu_user_model.py tests/test_gpu_fixes.py "tests/test_gpu_parity.py::test_warm_start_rollout_equals_the_oracles" -x -q -m gpu > gpurun_out/generic_tests.txt 2>&1
tail -25 gpurun_out/generic_tests.txt
{
  echo "== exact derivatives, default route (k_backward_w3 fused)"; python scripts/bench_lq.py 8192 3 16
  echo "== finite differences, default route (k_derivatives_lq)"; python scripts/bench_lq.py 8192 2 0
  echo "== finite differences, dense sweep (ILQR_ROUTE_LQ_DENSE_FD = 2048)"; python scripts/bench_lq.py 8192 2 0 2048
  echo "== scripts/soak_lq.py 120 71"; python scripts/soak_lq.py 120 71
  echo "== scripts/soak_lq_iter.py 120 5"; python scripts/soak_lq_iter.py 120 5
} > gpurun_out/generic_bench.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/generic_bench.txt
