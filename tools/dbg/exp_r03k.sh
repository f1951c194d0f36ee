set -u
mkdir -p gpurun_out/r03k
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03k/gputests.log
python tools/ab_switch.py --variants "3=1;3=0" --rounds 4 --what iter > gpurun_out/r03k/ab_iter.jsonl 2> gpurun_out/r03k/ab_iter.err
cat gpurun_out/r03k/gputests.log
grep -h "ms_per_iter\|kkt_step\|step_update" gpurun_out/r03k/ab_iter.jsonl | cut -c1-200
