#!/usr/bin/env python3
"""Which call sites synchronise with the host?  Runs a few interior-point iterations with PAROPT_AMD_SYNC_TRACE=1
(context.cpp prints a backtrace at every host-synchronising exchange) in a child process and prints, per steady
iteration, the ordered list of sites (innermost non-plumbing frames).

    python tools/dbg/sync_sites.py --n 2000000 --c 4 --nwcon 100000 --qn bfgs        # config 4 shaped
    python tools/dbg/sync_sites.py --n 2000000 --c 32 --qn sr1                        # config 3 shaped"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import paropt_amd as pa
ctx = pa.Context(0)
prob = pa.SeparableProblem(ctx, %(problem)r, %(n)d, %(c)d, 0)
if %(nwcon)d > 0:
    prob.setWeighting(%(nwcon)d, %(nw)d, 0, 0)
prob.setLinearConstraints(False)
opts = {"qn_type": %(qn)r, "qn_subspace_size": %(k)d, "abs_res_tol": 1e-30, "start_affine_multiplier_min": 0.01,
        "max_major_iters": %(iters)d, "write_output_frequency": 0}
opts.update(%(extra)r)
ip = pa.InteriorPoint(prob, opts)
def cb(k):
    sys.stderr.write("== iteration %%d\n" %% k)
    sys.stderr.flush()
ip.setIterationCallback(cb)
ip.optimize()
"""


CHILD_TR = r"""
import sys
sys.path.insert(0, %(root)r)
import paropt_amd as pa
ctx = pa.Context(0)
prob = pa.SeparableProblem(ctx, "quadratic", %(n)d, %(c)d, 0)
tr = pa.TrustRegion(prob, {"qn_subspace_size": %(k)d, "tr_max_iterations": %(iters)d, "max_major_iters": 200})
tr.setEigenModelSynthetic(10, 0, 0, 2.0)
inner = [0]
def cb(i):
    if i > 0:
        s = tr.getState()
        inner[0] += s["subproblem_iters"] + s["adaptive_subproblem_iters"]
tr.setIterationCallback(cb)
tr.optimize()
sys.stderr.write("== inner iterations %%d\n" %% inner[0])
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tr", action="store_true", help="config 5 shaped: histogram of sites per inner iteration")
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--c", type=int, default=4)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--qn", default="bfgs")
    ap.add_argument("--problem", default="convex")
    ap.add_argument("--nwcon", type=int, default=0)
    ap.add_argument("--nw", type=int, default=20)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--show", type=int, default=14, help="iteration to list")
    ap.add_argument("--window", type=int, default=0, help="--tr: also list this many consecutive syncs")
    ap.add_argument("--opt", action="append", default=[], help="name=value option overrides")
    a = ap.parse_args()
    extra = {}
    for o in a.opt:
        k, v = o.split("=", 1)
        try:
            v = int(v)
        except ValueError:
            try:
                v = float(v)
            except ValueError:
                pass
        extra[k] = v
    if a.tr:
        code = CHILD_TR % dict(root=ROOT, n=a.n, c=a.c, k=a.k, iters=a.iters)
        env = dict(os.environ, PAROPT_AMD_SYNC_TRACE="1")
        p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-3000:])
            sys.exit(p.returncode)
        syms = subprocess.run(["c++filt"], input=p.stderr, stdout=subprocess.PIPE, text=True).stdout
        skip = re.compile(r"exchange_reduced|batch_flush|reduce_finish|BatchScope|batch_end|launch_reduce|po::reduce|"
                          r"libamdhip|libc\.so|python|\[0x")
        hist, cur, inner = {}, None, 1
        evs = []
        for line in syms.splitlines():
            if line.startswith("== inner iterations"):
                inner = max(1, int(line.split()[3]))
            elif line.startswith("== host sync"):
                cur = []
                evs.append(cur)
            elif cur is not None and "(" in line:
                m = re.search(r"\((.*)\+0x[0-9a-f]+\)", line)
                if m:
                    cur.append(m.group(1))
        for fr in evs:
            fr = [re.sub(r"\(.*", "", f) for f in fr if f and not skip.search(f)]
            key = " <- ".join(fr[:4])
            hist[key] = hist.get(key, 0) + 1
        print("inner iterations %d, host syncs %d = %.2f per inner iteration" % (inner, len(evs), len(evs) / inner))
        for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
            print("  %6.2f  %s" % (v / inner, k))
        if a.window > 0:  # the raw sequence somewhere in the middle of the run
            hdrs = [l for l in syms.splitlines() if l.startswith("== host sync")]
            lo = len(evs) // 2
            print("sequence of %d syncs from the middle of the run:" % a.window)
            for idx in range(lo, min(len(evs), lo + a.window)):
                fr = [re.sub(r"\(.*", "", f) for f in evs[idx] if f and not skip.search(f)]
                print("  %-14s %s" % (hdrs[idx].split("(")[1].rstrip(")"), " <- ".join(fr[:3])))
        return
    code = CHILD % dict(root=ROOT, problem=a.problem, n=a.n, c=a.c, nwcon=a.nwcon, nw=a.nw, qn=a.qn, k=a.k,
                        iters=a.iters, extra=extra)
    env = dict(os.environ, PAROPT_AMD_SYNC_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        sys.exit(p.returncode)
    syms = subprocess.run(["c++filt"], input=p.stderr, stdout=subprocess.PIPE, text=True).stdout
    skip = re.compile(r"exchange_reduced|batch_flush|reduce_finish|BatchScope|batch_end|launch_reduce|po::reduce|"
                      r"libamdhip|libc\.so|python|\[0x")
    per_iter, cur, it = {}, None, -1
    events = []
    for line in syms.splitlines():
        if line.startswith("== iteration"):
            it = int(line.split()[2])
            continue
        if line.startswith("== host sync"):
            cur = {"it": it, "hdr": line, "frames": []}
            events.append(cur)
            continue
        if cur is not None and "(" in line:
            m = re.search(r"\((.*)\+0x[0-9a-f]+\)", line)
            name = m.group(1) if m else line
            cur["frames"].append(name)
    for e in events:
        fr = [f for f in e["frames"] if f and not skip.search(f)]
        short = [re.sub(r"\(.*", "", f) for f in fr[:3]]
        per_iter.setdefault(e["it"], []).append((e["hdr"].split("(")[1].rstrip(")"), " <- ".join(short)))
    # callbacks run at the END of iteration k: syncs printed after "== iteration k" belong to iteration k + 1
    tgt = a.show
    rows = per_iter.get(tgt, [])
    print("host syncs between the callbacks of iterations %d and %d: %d" % (tgt, tgt + 1, len(rows)))
    for vals, site in rows:
        print("  %-12s %s" % (vals, site))
    counts = {k: len(v) for k, v in sorted(per_iter.items())}
    print("per iteration:", counts)


if __name__ == "__main__":
    main()
