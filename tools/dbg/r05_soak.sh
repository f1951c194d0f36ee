#!/bin/bash
# round 5, final build: the GPU suite twice in a row on one box, then a 2000-draw campaign (fresh seed) with 300
# quasi-Newton sequences and 40 large-n draws
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
( for k in 1 2; do python -m pytest tests -m gpu -q --no-header 2>&1 | grep -E "passed|failed"; done
  F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
  echo "# PAROPT_SWEEP_CASES=2000 PAROPT_SWEEP_SEED=7777 PAROPT_SWEEP_QN_CASES=300 PAROPT_SWEEP_LARGE_CASES=40"
  PAROPT_SWEEP_CASES=2000 PAROPT_SWEEP_SEED=7777 PAROPT_SWEEP_QN_CASES=300 PAROPT_SWEEP_LARGE_CASES=40 timeout 2400 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1000
) > gpurun_out/r05_soak.txt 2>&1
grep -E "passed|failed|differ|ERROR|^CASE|^LARGE|^HOST|^FACADE|^QN" gpurun_out/r05_soak.txt | cut -c1-220
