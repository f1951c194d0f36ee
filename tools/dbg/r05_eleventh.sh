#!/bin/bash
# round 5, eleventh GPU call: the C++ facade tests (ParOptInfeasSubproblem over both subproblems), cycle stamps of the
# Gram kernels at 43 / 56 / 64 / 73 columns
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_cpp_facade.py -m gpu -q --no-header 2>&1 | tail -5
for cols in 43 56 64 73 80; do
  rows=128; if [ $cols -gt 64 ]; then rows=64; fi
  echo "== $cols columns"
  STAMP_COLS=$cols STAMP_ROWS=$rows PAROPT_AMD_WGRAM_ABLATE=16 python3 tools/dbg/wgram_stamps.py 2>&1 | tail -4
done | tee gpurun_out/r05_wgram_stamps.txt
