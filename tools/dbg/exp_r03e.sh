set -u
mkdir -p gpurun_out/r03e
python tools/ab_switch.py --variants "0=0,4=0;0=0,4=1;0=0,4=2;0=0,4=3" --rounds 8 --what micro --filter "wgram" > gpurun_out/r03e/ab_micro.jsonl 2> gpurun_out/r03e/ab_micro.err
python tools/ab_switch.py --variants "0=0,4=0;0=0,4=2;0=1,4=2" --rounds 4 --what iter > gpurun_out/r03e/ab_iter.jsonl 2> gpurun_out/r03e/ab_iter.err
cut -c1-220 gpurun_out/r03e/ab_micro.jsonl; grep -h "ms_per_iter\|wgram_launch\|setup_kkt" gpurun_out/r03e/ab_iter.jsonl | cut -c1-220
