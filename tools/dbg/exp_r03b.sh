set -u
mkdir -p gpurun_out/r03b
python tools/tr_inexact_rows.py > gpurun_out/r03b/tr_inexact.jsonl 2> gpurun_out/r03b/tr_inexact.err
for rs in 1 0; do for l2 in 1 0; do
PAROPT_AMD_WGRAM_RS=$rs PAROPT_AMD_LINCOMB_2D=$l2 python tools/microbench.py --tag rs${rs}_l2d${l2} --reps 5 2>/dev/null | grep -E "wgram|lincomb" ; done; done > gpurun_out/r03b/micro.jsonl
PAROPT_AMD_WGRAM_RS=1 python tools/microbench.py --tag rs1_c8k40 --n 10000000 --c 8 --k 0 --reps 5 2>/dev/null | grep -E "wgram" >> gpurun_out/r03b/micro.jsonl
PAROPT_AMD_WGRAM_RS=0 python tools/microbench.py --tag rs0_c8k40 --n 10000000 --c 8 --k 0 --reps 5 2>/dev/null | grep -E "wgram" >> gpurun_out/r03b/micro.jsonl
python -m pytest tests/test_gpu_vec.py -q -m gpu -k wgram 2>&1 | tail -3 > gpurun_out/r03b/wgram_tests.log
python bench.py --steps 20 --warmup 12 --repeats 3 --no-cpu-baseline --boundary builtin --skip-extension-variant > gpurun_out/r03b/bench_rs1.json 2>/dev/null
PAROPT_AMD_WGRAM_RS=0 PAROPT_AMD_LINCOMB_2D=0 python bench.py --steps 20 --warmup 12 --repeats 3 --no-cpu-baseline --boundary builtin --skip-extension-variant > gpurun_out/r03b/bench_rs0.json 2>/dev/null
cat gpurun_out/r03b/micro.jsonl | cut -c1-200
