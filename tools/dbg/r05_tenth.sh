#!/bin/bash
# round 5, tenth GPU call: line of record + kernel statistics + PMC of config 3 with L-BFGS(20) (73-column panel)
mkdir -p gpurun_out
bash tools/collect_r05.sh c3l > gpurun_out/r05_collect_c3l.log 2>&1
tail -2 gpurun_out/r05_collect_c3l.log | cut -c1-400
head -30 gpurun_out/r05_pmc_c3l.txt | cut -c1-200
