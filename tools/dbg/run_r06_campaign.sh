# Round-6 soak on the GPU box: the suite, then drawn-case campaigns with fresh seeds (interior point; trust region +
# the compiled reference's fixtures) on the build with the fused corrector solve.  Output under gpurun_out/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_suite.txt
PAROPT_SWEEP_SEED=606 PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_FIXTURE=1 timeout 1500 python tests/test_gpu_random_sweep.py > gpurun_out/r06_sweep_campaign.txt 2>&1
PAROPT_TR_SWEEP_SEED=707 PAROPT_TR_SWEEP_CASES=120 PAROPT_TR_SWEEP_FIXTURE=1 timeout 1200 python tests/test_gpu_tr_sweep.py > gpurun_out/r06_tr_sweep_campaign.txt 2>&1
tail -3 gpurun_out/r06_suite.txt; tail -4 gpurun_out/r06_sweep_campaign.txt; tail -4 gpurun_out/r06_tr_sweep_campaign.txt
