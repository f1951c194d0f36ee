set -u
mkdir -p gpurun_out/r03i
python tools/ab_switch.py --variants "0=0,4=2;0=1,4=2;0=1,4=0" --rounds 6 --what micro --filter "wgram" > gpurun_out/r03i/ab_micro.jsonl 2> gpurun_out/r03i/ab_micro.err
python tools/ab_switch.py --variants "0=0,4=2;0=1,4=2" --rounds 4 --what iter > gpurun_out/r03i/ab_iter.jsonl 2> gpurun_out/r03i/ab_iter.err
for pr in 2; do PAROPT_AMD_WGRAM_PRIO=$pr PAROPT_AMD_WGRAM_RS=1 PAROPT_AMD_WGRAM_ABLATE=16 python tools/dbg/wgram_stamps.py | tail -4; done > gpurun_out/r03i/stamps.txt 2>&1
cut -c1-200 gpurun_out/r03i/ab_micro.jsonl; grep -h "ms_per_iter\|wgram_launch\|setup_kkt" gpurun_out/r03i/ab_iter.jsonl | cut -c1-200; cat gpurun_out/r03i/stamps.txt
python -m pytest tests/test_gpu_vec.py -q -m gpu -k wgram 2>&1 | tail -2
