#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
PAROPT_TR_SWEEP_FIXTURE=1 PAROPT_TR_SWEEP_CASES=1 timeout 1500 python3 tests/test_gpu_tr_sweep.py 2>&1 | grep -v "$F" | cut -c1-1500 > gpurun_out/r05_tr_sweep_reference_fixture.txt
grep "differ\|ERROR\|^TR FIXTURE" gpurun_out/r05_tr_sweep_reference_fixture.txt | cut -c1-200
