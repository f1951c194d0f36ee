#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/collect_r05.sh c2 > gpurun_out/r05_collect_c2.log 2>&1
tail -1 gpurun_out/r05_collect_c2.log | cut -c1-300
head -8 gpurun_out/r05_pmc_c2.txt | cut -c1-170
