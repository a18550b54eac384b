#!/bin/bash
# Section times of the generic backward step (k_backward_w3 by default, k_backward_w2 with route 1024) from an experiment build with cycle
# marks, and the MFMA / VALU co-issue microbenchmark.  Run through gpurun:  gpurun --timeout 900 -- 'bash scripts/w2_sections.sh'
#   (build first, in the container:  hipcc <FLAGS of ilqr_amd/_build.py> -DILQR_W2_TIMING -o ilqr_amd/lib/libilqr_amd_timing.so ilqr_amd/csrc/capi.hip
#    and  hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/coissue scripts/ubench/coissue.hip)
mkdir -p gpurun_out
{
  echo "== coissue"; ./scripts/ubench/coissue
  echo "== stock library, exact derivatives"; python scripts/bench_lq.py 8192 2 16
  echo "== timing build, exact derivatives, k_backward_w3 (marks perturb the schedule: use the proportions; the step sections are printed per LITERAL box-QP)"
  ILQR_AMD_LIB=$PWD/ilqr_amd/lib/libilqr_amd_timing.so python scripts/bench_lq.py 8192 2 16
  echo "== timing build, exact derivatives, k_backward_w2 (route 1024)"
  ILQR_AMD_LIB=$PWD/ilqr_amd/lib/libilqr_amd_timing.so python scripts/bench_lq.py 8192 2 16 1024
  echo "== timing build, n=12 m=4, k_backward_w3"
  LQ_N=12 LQ_M=4 ILQR_AMD_LIB=$PWD/ilqr_amd/lib/libilqr_amd_timing.so python scripts/bench_lq.py 8192 2 16
} > gpurun_out/w2_sections.txt 2>&1
tail -40 gpurun_out/w2_sections.txt
