#!/usr/bin/env python3
"""Micro-benchmarks on one GPU: every hot kernel of an interior-point iteration in isolation
(po_bench_kernels), as JSON lines with GB/s against the HBM roofline."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--c", type=int, default=32, help="dense constraints")
    ap.add_argument("--k", type=int, default=10, help="quasi-Newton panel columns (<= 12)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tag", type=str, default="")
    ap.add_argument("--vec-api", action="store_true",
                    help="roofline rows of the ParOptVec operations and ParOptQuasiNewton::mult (with the ceiling of "
                         "each stream mix) instead of the iteration kernels")
    a = ap.parse_args()
    import paropt_amd as pa

    ctx = pa.Context(0)
    if a.vec_api:
        for r in pa.bench_vec_api(ctx, a.n, a.reps):
            r.update(n=a.n, tag=a.tag)
            print(json.dumps(r), flush=True)
        return
    for r in pa.bench_kernels(ctx, a.n, a.c, a.k, a.reps):
        r.update(n=a.n, c=a.c, k=a.k, tag=a.tag)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
