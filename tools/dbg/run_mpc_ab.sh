cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ip.py -m gpu -q -k "predictor_corrector_fused" 2>&1 | grep -E "^E  |passed|failed" | head
