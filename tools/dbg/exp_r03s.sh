set -u
mkdir -p gpurun_out/r03s
python tools/ab_switch.py --variants "8=3,9=4;8=2,9=4;8=1,9=4;8=2,9=3;8=2,9=2;8=1,9=2" --rounds 4 --what iter > gpurun_out/r03s/ab.jsonl 2> gpurun_out/r03s/err1
grep -h "ms_per_iter" gpurun_out/r03s/ab.jsonl | cut -c1-200
python tools/ab_switch.py --variants "8=3,9=4;8=2,9=3;8=1,9=2" --rounds 3 --what iter --n 10000000 --c 8 --k 20 --qn bfgs --problem quadratic > gpurun_out/r03s/ab_c2.jsonl 2> gpurun_out/r03s/err2
grep -h "ms_per_iter" gpurun_out/r03s/ab_c2.jsonl | cut -c1-200
