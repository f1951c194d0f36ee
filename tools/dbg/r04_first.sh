set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_a.log 2>&1; echo "pytest rc $?" 
tail -3 gpurun_out/r04_gputests_a.log
args="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 4 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin"
rm -rf gpurun_out/trace_c4
rocprofv3 --kernel-trace -d gpurun_out/trace_c4 -o t --output-format csv -- python3 bench.py $args > gpurun_out/trace_c4.json 2> gpurun_out/trace_c4.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c4 260 > gpurun_out/r04_launch_seq_c4_before.txt
rm -rf gpurun_out/trace_c5
rocprofv3 --kernel-trace -d gpurun_out/trace_c5 -o t --output-format csv -- python3 tools/bench_tr.py --no-cpu-baseline --tr-iters 3 > gpurun_out/trace_c5.json 2> gpurun_out/trace_c5.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c5 200 > gpurun_out/r04_launch_seq_c5_before.txt
find gpurun_out/trace_c4 gpurun_out/trace_c5 -name '*.csv' -size +1M -delete
python3 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin > gpurun_out/r04_c4_before.json 2>/dev/null
python3 bench.py --nglobal 5000000 --ncon 4 --nwcon 250000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin > gpurun_out/r04_c4_before_quarter.json 2>/dev/null
python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_c5_before.json 2>/dev/null
head -c 400 gpurun_out/r04_c4_before.json; echo; head -c 600 gpurun_out/r04_c5_before.json
