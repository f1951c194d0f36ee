#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
( for i in 160 328 714 1045; do echo "=== seed 1111 draw $i"; PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=1111 python3 tools/dbg/sweep_case_detail2.py $i 2>&1 | grep -v "$F" | cut -c1-330; done
  for i in 560 736 1163; do echo "=== seed 2222 draw $i"; PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=2222 python3 tools/dbg/sweep_case_detail2.py $i 2>&1 | grep -v "$F" | cut -c1-330; done
  echo "=== seed 3333 draw 527 (RCCL)"; PAROPT_SWEEP_RCCL=1 PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=3333 python3 tools/dbg/sweep_case_detail2.py 527 2>&1 | grep -v "$F" | cut -c1-330
  echo "=== seed 3333 draw 527 (self communicator)"; PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=3333 python3 tools/dbg/sweep_case_detail2.py 527 2>&1 | grep -v "$F" | cut -c1-330
) > gpurun_out/r05_sweep_details.txt
wc -l gpurun_out/r05_sweep_details.txt
