set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03t
rocprofv3 --kernel-trace --stats -d gpurun_out/r03t/prof -o c3 --output-format csv -- python3 bench.py --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin > gpurun_out/r03t/bench.json 2> gpurun_out/r03t/err
rm -f gpurun_out/r03t/prof/*_kernel_trace.csv
head -30 gpurun_out/r03t/prof/*kernel_stats.csv | cut -c1-180
