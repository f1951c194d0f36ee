#!/bin/bash
# round 5, sixth GPU call: remaining tests, lines of record of configs 3, 2 and 5 (kernel stats + PMC)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_cpp_facade.py tests/test_gpu_tr.py -q --no-header 2>&1 | tail -30 > gpurun_out/r05_tests6.log
tail -5 gpurun_out/r05_tests6.log
bash tools/collect_r05.sh c3 > gpurun_out/r05_collect_c3.log 2>&1
tail -1 gpurun_out/r05_collect_c3.log | cut -c1-300
bash tools/collect_r05.sh c2 > gpurun_out/r05_collect_c2.log 2>&1
tail -1 gpurun_out/r05_collect_c2.log | cut -c1-300
bash tools/collect_r05.sh c5 > gpurun_out/r05_collect_c5.log 2>&1
tail -1 gpurun_out/r05_collect_c5.log | cut -c1-300
