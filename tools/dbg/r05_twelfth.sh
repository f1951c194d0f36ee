#!/bin/bash
# round 5, twelfth GPU call: whole GPU suite on the final build of the wide-panel Gram, then the convergent comparison
# (config 3's data with L-BFGS(20) to the solver's stop) with the 64-row producer/consumer Gram in place
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -15 > gpurun_out/r05_tests12.log
tail -3 gpurun_out/r05_tests12.log
python3 tools/bench_convergent.py --cpu-n 2500000 --repeats 1 --max-iters 8000 > gpurun_out/r05_convergent_c3_lbfgs20.json 2> gpurun_out/r05_convergent.err
cut -c1-1500 gpurun_out/r05_convergent_c3_lbfgs20.json
