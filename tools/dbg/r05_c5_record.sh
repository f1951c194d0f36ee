#!/bin/bash
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/collect_r05.sh c5 > gpurun_out/r05_collect_c5.log 2>&1
tail -1 gpurun_out/r05_collect_c5.log | cut -c1-500
