#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/collect_r05.sh c3 > gpurun_out/r05_collect_c3.log 2>&1
tail -1 gpurun_out/r05_collect_c3.log | cut -c1-300
head -10 gpurun_out/r05_pmc_c3.txt | cut -c1-170
