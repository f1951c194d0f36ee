set -u
mkdir -p gpurun_out/r03c
python tools/ab_switch.py --variants "0=0,1=0;0=1,1=0;0=0,1=1;0=1,1=1" --rounds 3 --what iter > gpurun_out/r03c/ab_iter.jsonl 2> gpurun_out/r03c/ab_iter.err
python tools/ab_switch.py --variants "0=0;0=1" --rounds 4 --what micro --filter wgram > gpurun_out/r03c/ab_micro.jsonl 2> gpurun_out/r03c/ab_micro.err
for rs in 0 1; do PAROPT_AMD_WGRAM_RS=$rs PAROPT_AMD_WGRAM_ABLATE=16 python tools/dbg/wgram_stamps.py; done > gpurun_out/r03c/stamps.txt 2>&1
python tools/tr_inexact_rows.py > gpurun_out/r03c/tr_inexact.jsonl 2> gpurun_out/r03c/tr_inexact.err
python -m pytest tests/test_gpu_tr.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r03c/tr_tests.log
grep -h median gpurun_out/r03c/ab_iter.jsonl | cut -c1-220
cut -c1-220 gpurun_out/r03c/ab_micro.jsonl
cat gpurun_out/r03c/stamps.txt; cat gpurun_out/r03c/tr_tests.log
