cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bench_launcher.py tests/test_gpu_ip.py tests/test_gpu_tr.py tests/test_gpu_user_problem.py -m gpu -q 2>&1 | tail -3
bash tools/collect_r06.sh c5 > /dev/null 2>&1
head -c 300 gpurun_out/r06_bench_c5.json
