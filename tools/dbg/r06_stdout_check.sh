# stdout of bench.py is the ONE JSON line, also when RCCL prints its banner (forced single-rank communicator) and in the
# driver's own launch form (torch.distributed.run around bench.py; two ranks sharing GPU 0 over gloo on this box)
cd $GRAFT_REPO_ROOT
PAROPT_AMD_FORCE_RCCL=1 python3 bench.py --nglobal 2000000 --steps 3 --warmup 3 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 1 > /tmp/o1.txt 2> /tmp/e1.txt
echo "direct: stdout lines $(wc -l < /tmp/o1.txt), starts with $(head -c 12 /tmp/o1.txt); RCCL banner lines on stderr: $(grep -c 'RCCL version' /tmp/e1.txt)"
PAROPT_AMD_FORCE_RCCL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --nglobal 2000000 --steps 3 --warmup 3 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 1 > /tmp/o2.txt 2> /tmp/e2.txt
echo "torchrun N=1: stdout lines $(wc -l < /tmp/o2.txt), starts with $(head -c 12 /tmp/o2.txt); RCCL banner lines on stderr: $(grep -c 'RCCL version' /tmp/e2.txt)"
PAROPT_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --nglobal 2000000 --steps 3 --warmup 3 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 1 > /tmp/o3.txt 2> /tmp/e3.txt
echo "torchrun N=2 (shared GPU, gloo): rc $?, stdout lines $(wc -l < /tmp/o3.txt), starts with $(head -c 12 /tmp/o3.txt)"
python3 -c "
import json
d = json.loads(open('/tmp/o3.txt').read())
print({k: d[k] for k in ('n_gpus', 'value', 'ms_per_step', 'scaling')}, d['config']['collective'], d['config']['n_local'])"
