#!/bin/bash
# round 5: sparse-constraint tests on the final group kernels, then config 4's line of record again (bench + boundary +
# CPU reference, kernel statistics, PMC) in one call
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_vec.py tests/test_gpu_user_problem.py tests/test_gpu_ip.py tests/test_gpu_random_sweep.py tests/test_gpu_tr.py -m gpu -q --no-header -x 2>&1 | tail -3
bash tools/collect_r05.sh c4 > gpurun_out/r05_collect_c4.log 2>&1
tail -1 gpurun_out/r05_collect_c4.log | cut -c1-400
head -16 gpurun_out/r05_pmc_c4.txt | cut -c1-180
