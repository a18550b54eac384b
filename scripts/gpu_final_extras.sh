#!/bin/bash
# Round-end extras in one gpurun call: the batch sweep, the configs[4] walks into the tracked parity file, the whole-solve and PCIe-inclusive
# figures, the bench line again (now with the counters of k_solve_wide2).   gpurun --timeout 1800 -- 'bash scripts/gpu_final_extras.sh r05d'
R=${1:-r05d}
mkdir -p gpurun_out
BATCHES="1024 2048 4096 8192 16384 32768" bash scripts/batch_sweep.sh $R > /dev/null 2>&1
cat gpurun_out/sweep_$R.txt
cp profiles/parity_r05.json gpurun_out/parity_r05.json
ILQR_PARITY_JSON=$PWD/gpurun_out/parity_r05.json python -m pytest tests/test_gpu_lq_end_to_end.py -q -m gpu 2>&1 | tail -2
python scripts/full_solve_time.py > gpurun_out/${R}_full_solve.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_full_solve.txt | tail -4
python scripts/pcie_inclusive.py > gpurun_out/${R}_pcie.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${R}_pcie.txt | tail -4
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; tail -c 600 gpurun_out/${R}_bench_default.json
