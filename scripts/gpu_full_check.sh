#!/bin/bash
# The whole GPU suite, then the default bench line and the LQ stage times.   gpurun --timeout 2400 -- 'bash scripts/gpu_full_check.sh'
mkdir -p gpurun_out
python -m pytest tests/test_gpu_generic_backward.py tests/test_gpu_lq_end_to_end.py tests/test_gpu_analytic.py tests/test_gpu_user_model.py -x -q -m gpu > gpurun_out/generic_tests.txt 2>&1
tail -15 gpurun_out/generic_tests.txt
python -m pytest tests -q -m gpu --deselect tests/test_gpu_generic_backward.py --deselect tests/test_gpu_lq_end_to_end.py --deselect tests/test_gpu_analytic.py --deselect tests/test_gpu_user_model.py > gpurun_out/gpu_tests.txt 2>&1
tail -15 gpurun_out/gpu_tests.txt
{
  echo "== exact derivatives, default route (k_backward_w3 fused)"; python scripts/bench_lq.py 8192 3 16
  echo "== finite differences, default route"; python scripts/bench_lq.py 8192 2 0
} > gpurun_out/generic_bench.txt 2>&1
cat gpurun_out/generic_bench.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json
