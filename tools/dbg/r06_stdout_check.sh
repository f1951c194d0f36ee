# stdout of bench.py is the ONE JSON line, also when RCCL prints its banner (forced single-rank communicator; the same
# under torchrun as the driver launches it)
cd $GRAFT_REPO_ROOT
PAROPT_AMD_FORCE_RCCL=1 python3 bench.py --nglobal 2000000 --steps 3 --warmup 3 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 1 > /tmp/o1.txt 2> /tmp/e1.txt
echo "direct: stdout lines $(wc -l < /tmp/o1.txt), starts with $(head -c 12 /tmp/o1.txt); RCCL banner lines on stderr: $(grep -c 'RCCL version' /tmp/e1.txt)"
PAROPT_AMD_FORCE_RCCL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --nglobal 2000000 --steps 3 --warmup 3 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 1 > /tmp/o2.txt 2> /tmp/e2.txt
echo "torchrun: stdout lines $(wc -l < /tmp/o2.txt), starts with $(head -c 12 /tmp/o2.txt); RCCL banner lines on stderr: $(grep -c 'RCCL version' /tmp/e2.txt)"
python -m pytest tests/test_bench_launcher.py tests/test_gpu_multirank.py -q -m gpu 2>&1 | tail -3
