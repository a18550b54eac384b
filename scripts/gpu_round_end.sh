#!/bin/bash
# What the driver runs at round end, plus this round's evidence, in one gpurun call:
#   the whole GPU suite, the smoke entry, every rocprofv3 pass behind profiles/ (collect_profiles.sh), the section clocks of the generic
#   backward pass, the soaks.     gpurun --timeout 5400 -- 'bash scripts/gpu_round_end.sh r05'
R=${1:-r05}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_$R.txt 2>&1
grep -E "passed|failed" gpurun_out/gpu_tests_$R.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$R.txt 2>&1; tail -1 gpurun_out/smoke_$R.txt
bash scripts/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
tail -2 gpurun_out/collect_$R.log
bash scripts/w2_sections.sh > /dev/null 2>&1
{
  echo "== scripts/soak_lq.py 240 71"; python scripts/soak_lq.py 240 71
  echo "== scripts/soak_lq.py 240 72"; python scripts/soak_lq.py 240 72
  echo "== scripts/soak_lq_iter.py 240 5"; python scripts/soak_lq_iter.py 240 5
  echo "== scripts/soak.py 200 81"; python scripts/soak.py 200 81
} > gpurun_out/${R}_soak.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_soak.txt
