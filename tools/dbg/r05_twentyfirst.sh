#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
timeout 1500 python3 tests/test_gpu_tr_sweep.py 2>&1 | grep -v "$F" | cut -c1-2500 > gpurun_out/r05_tr_sweep_default.txt
cat gpurun_out/r05_tr_sweep_default.txt | cut -c1-1600
