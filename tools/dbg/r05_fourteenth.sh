#!/bin/bash
# round 5, fourteenth GPU call: ordered kernel dispatches of one iteration at configs 4 and 5 (which small launches sit
# next to which)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/tr14_c4 gpurun_out/tr14_c5
rocprofv3 --kernel-trace -d gpurun_out/tr14_c4 -o c4 --output-format csv -- python3 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin > /dev/null 2> gpurun_out/tr14_c4.err
python3 tools/dbg/launch_seq.py gpurun_out/tr14_c4 1500 > gpurun_out/r05_launch_seq_c4.txt
rocprofv3 --kernel-trace -d gpurun_out/tr14_c5 -o c5 --output-format csv -- python3 tools/bench_tr.py --no-cpu-baseline --repeats 1 > /dev/null 2> gpurun_out/tr14_c5.err
python3 tools/dbg/launch_seq.py gpurun_out/tr14_c5 1500 > gpurun_out/r05_launch_seq_c5.txt
rm -rf gpurun_out/tr14_c4 gpurun_out/tr14_c5
wc -l gpurun_out/r05_launch_seq_c4.txt
