#!/bin/bash
# round 5, fifteenth GPU call: random-sweep campaigns on the final build (a fresh seed; a second one with the two-tile
# first solve pass switched on)
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
( echo "# round 5, final build: PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=808"
  PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=808 timeout 1500 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | cut -c1-900
  echo
  echo "# the same build with the first solve pass two tiles per step (PAROPT_AMD_S2D_TWO=1): PAROPT_SWEEP_CASES=400 PAROPT_SWEEP_SEED=909"
  PAROPT_AMD_S2D_TWO=1 PAROPT_SWEEP_CASES=400 PAROPT_SWEEP_SEED=909 timeout 1200 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | cut -c1-900
) > gpurun_out/r05_sweep_campaigns.txt
tail -30 gpurun_out/r05_sweep_campaigns.txt | cut -c1-400
