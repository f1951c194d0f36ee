set -u
mkdir -p gpurun_out/r03o
python -m pytest tests/test_gpu_ip.py -q -m gpu -k "write_saving or golden" -x 2>&1 | tail -4
python tools/ab_switch.py --variants "2=1;2=0" --rounds 4 --what iter > gpurun_out/r03o/ab_iter.jsonl 2> gpurun_out/r03o/err
grep -h "ms_per_iter\|kkt_step" gpurun_out/r03o/ab_iter.jsonl | cut -c1-200
