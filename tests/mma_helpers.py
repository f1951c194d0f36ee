"""Shared helpers of the MMA parity tests (oracle and GPU)."""
import numpy as np

from conftest import ip_options_from_case

MMA_COLS = ("fobj", "l1", "linfty", "l1_lambda", "infeas")


def parse_mma_table(text):
    """{mma iteration: (sub-iteration count, [fobj, l1-opt, linft-opt, l1-lambd, infeas])} (:584-592)."""
    rows = {}
    for ln in str(text).splitlines():
        p = ln.split()
        if len(p) == 7 and p[0].isdigit():
            rows[int(p[0])] = (int(p[1]), [float(v) for v in p[2:]])
    return rows


def mma_options_from_case(case):
    a = case["args"]
    return ip_options_from_case(case), {k[4:]: v for k, v in a.items() if k.startswith("mma.")}


def run_oracle_mma(case):
    from oracle import mma_oracle as mo
    from oracle import paropt_oracle as po

    a = case["args"]
    prob = po.SepProblem(a["problem"], a["n"], a.get("c", 2), seed=a.get("seed", 0),
                         nwcon=a.get("nwcon", 0), nw=a.get("nw", 0), nwstart=a.get("nwstart", 0),
                         nwskip=a.get("nwskip", 0), nwineq=a.get("nwineq", -1),
                         chain=(a["chain_span"], a.get("chain_stride", 1)) if a.get("chain_span", 0) else None)
    opts, mopts = mma_options_from_case(case)
    mma = mo.MMA(prob, mopts)
    ip = po.InteriorPoint(mma, opts)
    mma.optimize(ip)
    rows = [(t["sub_iter"], [t[c] for c in MMA_COLS]) for t in mma.trace]
    final = dict(iters=(mma.mma_iter, mma.subproblem_iter), fobj=mma.fobj, x=mma.x.copy(), z=mma.z.copy(),
                 norms=(po.VecOps(prob.comm).norm(mma.x), po.VecOps(prob.comm).norm(mma.L),
                        po.VecOps(prob.comm).norm(mma.U)))
    return rows, final


def compare_mma(g, rows, final, window, exact_frac=0.9):
    ref = parse_mma_table(g["paropt_mma"])
    ncmp = min(window, len(ref), len(rows))
    assert ncmp >= min(window, len(ref)), (ncmp, len(ref), len(rows))
    exact = 0
    for k in range(ncmp):
        sub, vals = rows[k]
        rsub, rvals = ref[k]
        # interior-point iterations of THIS subproblem solve (the table holds the running total); one
        # solve that crawls for dozens of iterations with steps ~1e-5 leaves at a round-off dependent
        # iteration, which must not count against every later row
        inc = sub - (rows[k - 1][0] if k else 0)
        rinc = rsub - (ref[k - 1][0] if k else 0)
        exact += int(inc == rinc)
        for name, a, b in zip(MMA_COLS, vals, rvals):
            # printed with 7 / 4 significant digits; the optimality norms are sums of many terms that
            # cancel to ~1e-2 of the gradient near the optimum, hence the absolute floor
            tol = (2e-6 if name == "fobj" else 2e-3) * abs(b) + (1e-9 if name == "fobj" else 2e-4)
            assert abs(a - b) <= tol, "%s @%d: %r vs %r" % (name, k, a, b)
    # interior-point iteration counts of the subproblem solves: bit-exact bookkeeping
    assert exact >= exact_frac * ncmp, "only %d of %d sub-iteration counts identical" % (exact, ncmp)
    return ncmp
