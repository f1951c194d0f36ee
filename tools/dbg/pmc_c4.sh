set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 6 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_c4_fetch -o fetch --output-format csv -- python3 bench.py $args > gpurun_out/pmc_c4_fetch.out 2> gpurun_out/pmc_c4_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_c4_write -o write --output-format csv -- python3 bench.py $args > gpurun_out/pmc_c4_write.out 2> gpurun_out/pmc_c4_write.err
python3 tools/pmc_summary.py gpurun_out/pmc_c4.json gpurun_out/pmc_c4_fetch gpurun_out/pmc_c4_write
find gpurun_out/pmc_c4_* -name '*.csv' -size +2M -delete 2>/dev/null
