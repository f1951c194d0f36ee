# Round-6 end-of-round check on the GPU box: the driver's own sequence (suite, smoke, bench line), the suite a second
# time, and a drawn-case campaign with every reduction through the RCCL communicator (single rank).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -7 | head -2
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_driver_form.json 2> gpurun_out/r06_bench_driver_form.err
tail -4 gpurun_out/r06_bench_driver_form.err
python -c "
import json
d = json.loads(open('gpurun_out/r06_bench_driver_form.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps', 'warmup')}, d['roofline']['frac'], d['roofline']['traffic_source'][:40], d['cpu_baseline']['value'])
"
python -m pytest tests -q -m gpu 2>&1 | tail -7 | head -2
PAROPT_SWEEP_RCCL=1 PAROPT_SWEEP_SEED=616 PAROPT_SWEEP_CASES=300 timeout 1200 python tests/test_gpu_random_sweep.py 2>&1 | grep -v "^paropt_amd: the sparse\|^ParOpt" | tail -8
