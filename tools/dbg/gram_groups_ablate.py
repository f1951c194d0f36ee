"""Tuning aid: the Gram pass with the structured panel image riding in it, with pieces cut (PAROPT_AMD_WGRAM_ABLATE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import paropt_amd as pa

n, nw, nv = 20_000_000, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 25
ctx = pa.Context(0)
d = pa.PVec(ctx, n); d.fill_hash(1, 9, 0, 1.0, 0.5)
V = []
for j in range(nv):
    v = pa.PVec(ctx, n); v.fill_hash(1, 20 + j, 0, 2.0, -1.0); V.append(v)
U = [pa.PVec(ctx, n // nw) for _ in range(nv - 1)]
for rep in range(2):
    for name, fn in (("fused", lambda: pa.wgram_with_groups(d, V, n // nw, nw, 0, -1.0, U, rhs_last=True)),
                     ("plain", lambda: pa.wgram(d, V, rhs_last=True)),
                     ("panel", lambda: pa.group_panel(d, V[:nv - 1], n // nw, nw, 0, -1.0, U))):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        ctx.synchronize()
        print("ablate=%s nv=%d %s %.1f us" % (os.environ.get("PAROPT_AMD_WGRAM_ABLATE", "0"), nv, name, (time.perf_counter() - t0) * 1e5), flush=True)
