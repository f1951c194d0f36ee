#!/bin/bash
# A/B of kernel-variant builds on the GPU box: tools/microbench.py once per library (PAROPT_AMD_LIB), twice around
# to see the run-to-run noise.   usage: bash tools/ab_libs.sh <tag>=<lib.so> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in ${AB_ROUNDS:-1 2 3 4}; do
  for spec in "$@"; do
    tag=${spec%%=*}
    lib=${spec#*=}
    PAROPT_AMD_LIB=$PWD/$lib python3 tools/microbench.py --tag $tag.$round >> gpurun_out/ab_microbench.jsonl 2>> gpurun_out/ab.err
  done
done
python3 - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/ab_microbench.jsonl")]
tags = sorted({r["tag"].split(".")[0] for r in rows})
kern = []
for r in rows:
    if r["kernel"] not in kern:
        kern.append(r["kernel"])
print("GB/s from the best launch of every round: median over rounds [min..max]")
print("%-34s" % "kernel" + "".join("%24s" % t for t in tags))
for k in kern:
    line = "%-34s" % k[:34]
    for t in tags:
        v = [r["alg_GB"] / r["min_ms"] * 1e3 for r in rows if r["kernel"] == k and r["tag"].split(".")[0] == t]
        line += "%9.0f [%5.0f..%5.0f]" % (statistics.median(v), min(v), max(v)) if v else "%24s" % "-"
    print(line)
PY
