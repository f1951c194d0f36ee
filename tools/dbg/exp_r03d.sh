set -u
mkdir -p gpurun_out/r03d
python tools/ab_switch.py --variants "0=0,4=0;0=0,4=2;0=1,4=0;0=1,4=1;0=1,4=3" --rounds 4 --what micro --filter wgram > gpurun_out/r03d/ab_micro.jsonl 2> gpurun_out/r03d/ab_micro.err
for pr in 0 3; do PAROPT_AMD_WGRAM_PRIO=$pr PAROPT_AMD_WGRAM_RS=1 PAROPT_AMD_WGRAM_ABLATE=16 python tools/dbg/wgram_stamps.py | tail -4; done > gpurun_out/r03d/stamps.txt 2>&1
cut -c1-200 gpurun_out/r03d/ab_micro.jsonl; cat gpurun_out/r03d/stamps.txt
