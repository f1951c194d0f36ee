cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PAROPT_SWEEP_RCCL=1 PAROPT_SWEEP_SEED=616 PAROPT_SWEEP_CASES=300 timeout 1200 python tests/test_gpu_random_sweep.py > gpurun_out/r06_sweep_rccl_616.txt 2>&1
grep -v "^paropt_amd: the sparse\|^ParOpt" gpurun_out/r06_sweep_rccl_616.txt | grep "differ\|CASE" | cut -c1-1200
PAROPT_SWEEP_SEED=616 PAROPT_SWEEP_CASES=300 timeout 1200 python tests/test_gpu_random_sweep.py > gpurun_out/r06_sweep_616.txt 2>&1
grep -v "^paropt_amd: the sparse\|^ParOpt" gpurun_out/r06_sweep_616.txt | grep "differ\|CASE" | cut -c1-1200
