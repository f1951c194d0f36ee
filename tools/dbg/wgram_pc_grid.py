"""Tuning aid: the weighted Gram in its producer/consumer and single-role forms (PAROPT_AMD_WGRAM_PC=1/0, one process
each) over panel widths and vector lengths: average kernel ms (HIP events) and fraction of the HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
ctx = pa.Context(0)
for n in (5_000_000, 10_000_000, 20_000_000):
    d = pa.PVec(ctx, n); d.fill_hash(1, 9, 0, 1.0, 0.5)
    V = []
    for j in range(34):
        v = pa.PVec(ctx, n); v.fill_hash(1, 20 + j, 0, 2.0, -1.0); V.append(v)
    for nv in (5, 9, 13, 17, 25, 33):
        ms = pa.bench_wgram(d, V[:nv], reps=20)
        print("PC=%s n=%d nv=%d  %.1f us  %.2f of 8 TB/s" % (os.environ.get("PAROPT_AMD_WGRAM_PC", "1"), n, nv, ms * 1e3, 8.0 * (nv + 1) * n / (ms * 1e-3) / 8e12), flush=True)
    del V, d
