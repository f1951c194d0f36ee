"""What one host-synchronising reduction costs on this box (single rank: final-stage values -> device-to-host copy
-> stream sync), and what a whole tiny reduction (kernel + final stage + copy + sync) costs end to end."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
ctx = pa.Context(0)
for cnt in (2, 64, 2628):
    print(json.dumps(ctx.bench_collective(cnt, True, 200)))
x = pa.PVec(ctx, 1000).fill_hash(0, 1, 0, 1.0, 0.0)
y = pa.PVec(ctx, 1000).fill_hash(0, 2, 0, 1.0, 0.0)
for _ in range(50):
    x.dot(y)
ts = []
for _ in range(400):
    t0 = time.perf_counter(); x.dot(y); ts.append(1e6 * (time.perf_counter() - t0))
ts.sort()
print(json.dumps({"what": "po_vec_dot on 1000 elements end to end (2 launches + copy + sync), us", "median": ts[len(ts)//2], "min": ts[0], "p90": ts[int(0.9*len(ts))]}))
