set -u
mkdir -p gpurun_out/r03j
python tools/ab_switch.py --variants "1=0;1=1;1=2;1=1,7=4;1=1,7=16;1=1,7=32" --rounds 5 --what micro --filter "lincomb" > gpurun_out/r03j/ab_micro.jsonl 2> gpurun_out/r03j/ab_micro.err
cut -c1-200 gpurun_out/r03j/ab_micro.jsonl
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import paropt_amd as pa
ctx = pa.Context(0)
n = 50_000_000
x = pa.PVec(ctx, n).fill_hash(0, 1, 0, 1.0, 0.0); y = pa.PVec(ctx, n)
for k in range(3):
    print("stream ceilings: read-only %.0f GB/s copy %.0f GB/s" % (16e-9 * n / (pa.bench_stream(x, y, 0, 20) * 1e-3), 16e-9 * n / (pa.bench_stream(x, y, 1, 20) * 1e-3)))
PY
