#!/bin/bash
# round 5: final verification on one box -- the whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -6 > gpurun_out/r05_tests_final.log
grep -E "passed|failed" gpurun_out/r05_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
cut -c1-700 gpurun_out/r05_bench_default.json
