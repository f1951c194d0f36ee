set -u
mkdir -p gpurun_out/r03f
for ab in 16 17 19; do for pr in 0 2; do echo "== ablate $ab prio $pr (16 stamps only, 17 no matrix work, 19 no loads)"; PAROPT_AMD_WGRAM_PRIO=$pr PAROPT_AMD_WGRAM_RS=0 PAROPT_AMD_WGRAM_ABLATE=$ab python tools/dbg/wgram_stamps.py | tail -4; done; done > gpurun_out/r03f/stamps.txt 2>&1
cat gpurun_out/r03f/stamps.txt
