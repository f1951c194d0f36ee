#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_cpp_facade.py -k user_written -q --no-header -x 2>&1 | tail -60 > gpurun_out/r05_tests5.log
python -m pytest tests/test_gpu_tr.py -k user_written tests/test_gpu_user_problem.py -q --no-header 2>&1 | tail -15 >> gpurun_out/r05_tests5.log
tail -30 gpurun_out/r05_tests5.log
bash tools/collect_r05.sh c4 > gpurun_out/r05_collect_c4.log 2>&1
tail -2 gpurun_out/r05_collect_c4.log | cut -c1-300
