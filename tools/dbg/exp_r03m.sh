set -u
mkdir -p gpurun_out/r03m
python tools/ab_switch.py --variants "0=0;0=1" --rounds 6 --what micro --filter "wgram" --n 10000000 --c 48 --k 0 > gpurun_out/r03m/ab_micro_c2.jsonl 2> gpurun_out/r03m/err1
python tools/ab_switch.py --variants "0=0;0=1" --rounds 4 --what micro --filter "wgram" --n 50000000 --c 52 --k 0 > gpurun_out/r03m/ab_micro_ng14.jsonl 2> gpurun_out/r03m/err2
python tools/ab_switch.py --variants "0=0;0=1" --rounds 4 --what iter --n 10000000 --c 8 --k 20 --qn bfgs > gpurun_out/r03m/ab_iter_c2.jsonl 2> gpurun_out/r03m/err3
cut -c1-180 gpurun_out/r03m/ab_micro_c2.jsonl gpurun_out/r03m/ab_micro_ng14.jsonl; grep -h "ms_per_iter\|wgram_launch\|setup_kkt" gpurun_out/r03m/ab_iter_c2.jsonl | cut -c1-180
python -m pytest tests/test_gpu_vec.py -q -m gpu -k wgram 2>&1 | tail -2
