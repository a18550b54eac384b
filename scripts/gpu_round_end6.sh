#!/bin/bash
# Round 6, the final evidence in one gpurun call on the final sources: the GPU suite, smoke(), every rocprofv3 pass behind profiles/
# (collect_profiles.sh: traffic.json re-stamped, the default bench record, bench.py --extra-configs), the PCIe-inclusive rates.
#   gpurun --timeout 3000 -- 'bash scripts/gpu_round_end6.sh r06d'
R=${1:-r06d}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_$R.txt 2>&1
grep -E "passed|failed" gpurun_out/gpu_tests_$R.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$R.txt 2>&1; tail -1 gpurun_out/smoke_$R.txt
bash scripts/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
tail -2 gpurun_out/collect_$R.log
python scripts/pcie_inclusive.py > gpurun_out/${R}_pcie.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_pcie.txt | tail -9
cat gpurun_out/prof_$R/bench_default.json
