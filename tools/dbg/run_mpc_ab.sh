cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -8
PAROPT_AMD_HOST_TRACE=1 python tools/bench_tr.py --no-cpu-baseline --repeats 3 2>&1 | grep "host trace"
PAROPT_AMD_HOST_TRACE=1 python bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --boundary builtin --no-cpu-baseline --repeats 2 --skip-extension-variant 2>&1 | grep "host trace"
PAROPT_AMD_HOST_TRACE=1 python bench.py --nglobal 6250000 --steps 20 --warmup 5 --boundary builtin --no-cpu-baseline --repeats 2 --skip-extension-variant 2>&1 | grep "host trace\|\"value\"" | cut -c1-300
