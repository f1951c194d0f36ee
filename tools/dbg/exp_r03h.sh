set -u
mkdir -p gpurun_out/r03h
python tools/ab_switch.py --variants "6=1;6=0" --rounds 4 --what iter > gpurun_out/r03h/ab_iter.jsonl 2> gpurun_out/r03h/ab_iter.err
python tools/ab_switch.py --variants "6=1" --rounds 4 --what micro --filter "solve2r\|comp_merit" > gpurun_out/r03h/ab_micro.jsonl 2> gpurun_out/r03h/ab_micro.err
python tools/microbench.py --reps 5 2>/dev/null | grep -E "solve2r|comp_merit|trial" | cut -c1-150
grep -h "ms_per_iter\|kkt_step\|scale_step\|user_eval" gpurun_out/r03h/ab_iter.jsonl | cut -c1-200
