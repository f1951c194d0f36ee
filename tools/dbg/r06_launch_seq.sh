# the launches of config 5's inner iterations in order, with the gaps between them (round 6 sequence)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/tr6_c5
rocprofv3 --kernel-trace -d gpurun_out/tr6_c5 -o c5 --output-format csv -- python3 tools/bench_tr.py --no-cpu-baseline --repeats 1 > /dev/null 2>&1
python3 tools/dbg/launch_seq.py gpurun_out/tr6_c5 1500 > gpurun_out/r06_launch_seq_c5.txt
rm -rf gpurun_out/tr6_c5
wc -l gpurun_out/r06_launch_seq_c5.txt
