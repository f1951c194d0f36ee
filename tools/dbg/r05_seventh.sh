#!/bin/bash
# round 5, seventh GPU call: the whole GPU suite on the current build (GMRES helper refactor), then the experiment
# "final stage inside the first-stage kernel" on dot / norm (PAROPT_AMD_FUSED_FINAL=1): call and kernel times, bits
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -15 > gpurun_out/r05_tests7.log
tail -5 gpurun_out/r05_tests7.log
for n in 1000000 50000000; do
  python tools/microbench.py --vec-api --n $n --reps 200 --tag two_launch > gpurun_out/r05_ff_two_$n.jsonl 2>&1
  PAROPT_AMD_FUSED_FINAL=1 python tools/microbench.py --vec-api --n $n --reps 200 --tag fused > gpurun_out/r05_ff_fused_$n.jsonl 2>&1
done
python - <<'PY'
import json
for n in (1000000, 50000000):
    a = {r["op"]: r for r in map(json.loads, open("gpurun_out/r05_ff_two_%d.jsonl" % n)) if "op" in r}
    b = {r["op"]: r for r in map(json.loads, open("gpurun_out/r05_ff_fused_%d.jsonl" % n)) if "op" in r}
    for op in a:
        if any(k in op for k in ("dot", "norm", "Norm")):
            print(n, op, {k: (a[op].get(k), b[op].get(k)) for k in a[op] if k.endswith("_ms") or k.endswith("_us")})
PY
python - <<'PY'
import os, numpy as np
def run(env):
    import subprocess, sys
    code = ("import paropt_amd as pa, numpy as np\n"
            "ctx=pa.Context(0)\n"
            "rng=np.random.default_rng(5)\n"
            "out=[]\n"
            "for n in (1,63,1000,123457,5000001):\n"
            "    x=pa.Vec(ctx,n); y=pa.Vec(ctx,n)\n"
            "    x.from_numpy(rng.standard_normal(n)); y.from_numpy(rng.standard_normal(n))\n"
            "    out += [x.dot(y).hex(), x.norm().hex(), x.l1norm().hex(), x.maxabs().hex()]\n"
            "print(' '.join(out))\n")
    e = dict(os.environ); e.update(env)
    return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
a = run({}); b = run({"PAROPT_AMD_FUSED_FINAL": "1"})
print("bits identical:", a.stdout == b.stdout and len(a.stdout) > 10, a.stderr[-300:], b.stderr[-300:])
PY
