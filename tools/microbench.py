#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on one GPU: mdot and wgram GB/s vs the HBM roofline."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--nvecs", type=str, default="8,32,40")
    ap.add_argument("--wgram", type=str, default="42,48")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import paropt_amd as pa

    ctx = pa.Context(0)
    n = a.n
    nvmax = max([int(v) for v in a.nvecs.split(",")] + [int(v) for v in a.wgram.split(",") if v])
    x = pa.PVec(ctx, n).fill_hash(0, 10, 0, 2.0, -1.0)
    V = [pa.PVec(ctx, n).fill_hash(0, 20 + j, 0, 2.0, -1.0) for j in range(nvmax)]
    for nv in [int(v) for v in a.nvecs.split(",")]:
        ms, out = pa.bench_mdot(x, V[:nv], a.reps)
        gbs = 8.0 * (nv + 1) * n / (ms * 1e-3) * 1e-9
        print(json.dumps(dict(kernel="mdot", n=n, nvecs=nv, ms=ms, GBps=gbs, frac_of_8TBps=gbs / 8000.0)))
    d = pa.PVec(ctx, n).fill_hash(0, 9, 0, 1.0, 0.5)
    for nv in [int(v) for v in a.wgram.split(",") if v]:
        ms = pa.bench_wgram(d, V[:nv], a.reps)
        gbs = 8.0 * (nv + 1) * n / (ms * 1e-3) * 1e-9
        tf = nv * (nv + 1) * n / (ms * 1e-3) * 1e-12
        print(json.dumps(dict(kernel="wgram", n=n, nvecs=nv, ms=ms, GBps=gbs, TFLOPs=tf)))


if __name__ == "__main__":
    main()
