set -u
mkdir -p gpurun_out/r03q
python -m pytest tests/test_gpu_ip.py -q -m gpu -k "write_saving or golden" -x 2>&1 | tail -3
python tools/ab_switch.py --variants "7=1;7=0" --rounds 4 --what iter > gpurun_out/r03q/ab_iter.jsonl 2> gpurun_out/r03q/err
grep -h "ms_per_iter\|kkt_step" gpurun_out/r03q/ab_iter.jsonl | cut -c1-200
