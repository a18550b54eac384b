#!/bin/bash
# What the driver runs at round end, plus this round's evidence, in one gpurun call:
#   the whole GPU suite, the smoke entry, every rocprofv3 pass behind profiles/ (collect_profiles.sh), the section clocks of the generic backward pass.
#   gpurun --timeout 3600 -- 'bash scripts/gpu_round_end.sh r05'
R=${1:-r05}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_$R.txt 2>&1
tail -5 gpurun_out/gpu_tests_$R.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$R.txt 2>&1; tail -2 gpurun_out/smoke_$R.txt
bash scripts/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
tail -3 gpurun_out/collect_$R.log
bash scripts/w2_sections.sh > /dev/null 2>&1
tail -12 gpurun_out/w2_sections.txt
