"""
The numpy restatement of the trust-region driver (oracle/tr_oracle.py: quadratic / infeasibility /
compact-eigenvalue subproblems, SL1QP with the adaptive penalty update) pinned against trajectories
of the compiled reference (tests/golden/tr_*.npz from oracle/ref_driver.cpp mode "tr"): the
iteration table to its print precision, accept/reject and quasi-Newton flags exactly, the
interior-point iteration counts of both subproblem solves, trust-region radius, penalty parameters,
model values and the iterate itself.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from tr_helpers import TR_REFERENCE_IRREPRODUCIBLE, compare_tr, run_oracle_tr

TR_CASES = golden_names("tr_")


@pytest.mark.parametrize("name", TR_CASES)
def test_tr_trajectory(name):
    g, case = load_golden(name)
    window = 40 if "sr1" in name else 60
    # filter method with the restoration phase on: the reference's compatibility test looks at the
    # last constraint only and with |c| instead of max(0,-c) (:1826-1832), so every step of this case
    # is a restoration LP step and the iterates hop between LP vertices: compare the first 10
    unstable = name == "tr_filter_quadratic_n200_c3"
    if unstable:
        window = 10
    if name in TR_REFERENCE_IRREPRODUCIBLE:  # only the compared rows are run
        case["args"]["tr.tr_max_iterations"] = TR_REFERENCE_IRREPRODUCIBLE[name]["rows"]
    rows, snaps, final = run_oracle_tr(case)
    if name in TR_REFERENCE_IRREPRODUCIBLE:  # the reference run itself depends on the rank count past these rows
        nr = TR_REFERENCE_IRREPRODUCIBLE[name]["rows"]
        assert compare_tr(g, rows, snaps, final, nr, check_snaps=False, check_counts=False) == nr
        return
    n = compare_tr(g, rows, snaps, final, window, check_snaps=not unstable)
    if unstable:
        return
    if "sr1" not in name:
        assert final["iter_count"] == int(g["final/iter_count"][0])
        assert abs(final["fk"] - g["final/fk"][0]) <= 1e-6 * max(1.0, abs(g["final/fk"][0]))
        np.testing.assert_allclose(final["x"], g["final/x"], rtol=0, atol=1e-5 * max(1.0, np.abs(g["final/x"]).max()))
    assert n >= (12 if "tr_rand_" in name else 20)  # (the drawn-option goldens run 12 iterations)
