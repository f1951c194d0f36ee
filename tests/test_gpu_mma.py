"""
GPU parity of the method of moving asymptotes (through the C ABI) against trajectories of the compiled
reference (tests/golden/mma_*.npz): the iteration table to its print precision, the interior-point
iteration counts of the subproblem solves, asymptotes and the final point.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from mma_helpers import compare_mma, mma_options_from_case, parse_mma_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def run_gpu_mma(ctx, case):
    import paropt_amd as pa

    a = case["args"]
    prob = pa.SeparableProblem(ctx, a["problem"], a["n"], a.get("c", 2), a.get("seed", 0))
    if a.get("nwcon", 0) > 0:
        prob.setWeighting(a["nwcon"], a["nw"], a.get("nwstart", 0), a.get("nwskip", 0), a.get("nwineq", a["nwcon"]))
    if a.get("chain_span", 0) > 0:
        prob.setChain(a["chain_span"], a.get("chain_stride", 1), a.get("chain_reverse", 0))
    opts, mopts = mma_options_from_case(case)
    mma = pa.MMA(prob, dict(opts, **mopts))
    rows = []
    mma.setIterationCallback(lambda k: rows.append((mma.getState()["subproblem_iter"], mma.getLastRow())))
    mma.optimize()
    st = mma.getState()
    x, z, zw, zl, zu = mma.getOptimizedPoint()
    lo, up = mma.getAsymptotes()
    final = dict(iters=(st["mma_iter"], st["subproblem_iter"]), fobj=st["fobj"], x=x.to_numpy(), z=z,
                 norms=(x.norm(), lo.norm(), up.norm()))
    return mma, rows, final


@pytest.mark.parametrize("name", golden_names("mma_"))
def test_mma_trajectory_golden(ctx, name):
    g, case = load_golden(name)
    mma, rows, final = run_gpu_mma(ctx, case)
    n = compare_mma(g, rows, final, 40)
    assert n >= 15
    assert final["iters"][0] == int(g["final/iters"][0])
    assert abs(final["fobj"] - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
    np.testing.assert_allclose(final["x"], g["final/x"], rtol=0, atol=1e-5 * max(1.0, np.abs(g["final/x"]).max()))
    np.testing.assert_allclose(final["norms"], g["final/norms"], rtol=1e-6)
    assert len(parse_mma_table(mma.getHistory())) == final["iters"][0]
