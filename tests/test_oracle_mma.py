"""
The numpy restatement of ParOptMMA (oracle/mma_oracle.py) pinned against trajectories of the
compiled reference (tests/golden/mma_*.npz from oracle/ref_driver.cpp mode "mma"): the iteration table
to its print precision, the cumulative subproblem iteration counts, the final point.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from mma_helpers import compare_mma, run_oracle_mma

MMA_CASES = golden_names("mma_")


@pytest.mark.parametrize("name", MMA_CASES)
def test_mma_trajectory(name):
    g, case = load_golden(name)
    rows, final = run_oracle_mma(case)
    n = compare_mma(g, rows, final, 40)
    assert n >= 15
    np.testing.assert_array_equal(np.array(final["iters"])[:1], g["final/iters"][:1])
    assert abs(final["fobj"] - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
    np.testing.assert_allclose(final["x"], g["final/x"], rtol=0, atol=1e-5 * max(1.0, np.abs(g["final/x"]).max()))
    np.testing.assert_allclose(final["norms"], g["final/norms"], rtol=1e-6)
