#!/bin/bash
# round 5: the device against the compiled reference's fixture (400 draws)
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
PAROPT_SWEEP_FIXTURE=1 PAROPT_SWEEP_CASES=1 PAROPT_SWEEP_LARGE_CASES=0 PAROPT_SWEEP_QN_CASES=1 timeout 1500 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1100 > gpurun_out/r05_sweep_reference_fixture.txt
grep "differ\|ERROR\|^FIXTURE" gpurun_out/r05_sweep_reference_fixture.txt | cut -c1-250
