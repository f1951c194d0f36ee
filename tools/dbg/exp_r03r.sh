set -u
mkdir -p gpurun_out/r03r
python tools/ab_switch.py --variants "8=3;8=2;8=4;8=5;8=6;8=8" --rounds 3 --what iter > gpurun_out/r03r/ab_bpc3.jsonl 2> gpurun_out/r03r/err1
grep -h "ms_per_iter\|kkt_step\|step_update" gpurun_out/r03r/ab_bpc3.jsonl | cut -c1-150
python tools/ab_switch.py --variants "9=4;9=3;9=5;9=6;9=8" --rounds 3 --what iter > gpurun_out/r03r/ab_bpc4.jsonl 2> gpurun_out/r03r/err2
grep -h "ms_per_iter\|setup_kkt\|line_search" gpurun_out/r03r/ab_bpc4.jsonl | cut -c1-150
