# Round-6 second soak (final build): large-n interior-point draws against the oracle, a second trust-region campaign
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PAROPT_SWEEP_SEED=626 PAROPT_SWEEP_CASES=200 PAROPT_SWEEP_LARGE_CASES=60 timeout 1500 python tests/test_gpu_random_sweep.py > gpurun_out/r06_sweep_campaign2.txt 2>&1
PAROPT_TR_SWEEP_SEED=727 PAROPT_TR_SWEEP_CASES=200 timeout 1500 python tests/test_gpu_tr_sweep.py > gpurun_out/r06_tr_sweep_campaign2.txt 2>&1
grep "differ\|^CASE\|^LARGE\|^HOST\|^TR " gpurun_out/r06_sweep_campaign2.txt gpurun_out/r06_tr_sweep_campaign2.txt | cut -c1-700
