#!/bin/bash
# The generic path's GPU tests + stage times of configs[4] (stock library and the experiment build with cycle marks).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_generic_check.sh'
mkdir -p gpurun_out
python -m pytest tests/test_gpu_generic_backward.py tests/test_gpu_lq_end_to_end.py tests/test_gpu_analytic.py tests/test_gpu_user_model.py tests/test_gpu_fixes.py "tests/test_gpu_parity.py::test_warm_start_rollout_equals_the_oracles" -x -q -m gpu > gpurun_out/generic_tests.txt 2>&1
tail -25 gpurun_out/generic_tests.txt
{
  echo "== exact derivatives, default route (k_backward_w3 fused)"; python scripts/bench_lq.py 8192 3 16
  echo "== finite differences, default route"; python scripts/bench_lq.py 8192 2 0
  echo "== timing build, exact derivatives"
  ILQR_AMD_LIB=$PWD/ilqr_amd/lib/libilqr_amd_timing.so python scripts/bench_lq.py 8192 2 16
} > gpurun_out/generic_bench.txt 2>&1
cat gpurun_out/generic_bench.txt
