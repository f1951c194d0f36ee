set -u
mkdir -p gpurun_out/r03n
python tools/ab_switch.py --variants "0=0,4=0;0=0,4=2;0=1,4=2" --rounds 3 --what iter --n 20000000 --c 4 --k 10 --qn bfgs --nwcon 1000000 --nw 20 > gpurun_out/r03n/ab_iter_c4.jsonl 2> gpurun_out/r03n/err1
grep -h "ms_per_iter\|wgram_launch\|setup_kkt" gpurun_out/r03n/ab_iter_c4.jsonl | cut -c1-180
python tools/ab_switch.py --variants "0=0,4=0;0=1,4=2" --rounds 3 --what iter --n 10000000 --c 8 --k 20 --qn bfgs --problem quadratic > gpurun_out/r03n/ab_iter_c2.jsonl 2> gpurun_out/r03n/err2
grep -h "ms_per_iter\|wgram_launch" gpurun_out/r03n/ab_iter_c2.jsonl | cut -c1-180
tail -3 gpurun_out/r03n/err1
