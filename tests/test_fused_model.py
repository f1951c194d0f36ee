"""
The fused KKT algebra (one weighted Gram -> both Schur complements; dot-pass / host algebra /
axpy-pass solves) used by the HIP product is the same map as the reference's
setUpKKTSystem/computeKKTStep sequence: checked here on the CPU, against the golden
trajectories of the compiled reference, before any GPU is involved.
"""
import numpy as np
import pytest

from conftest import golden_names, ip_options_from_case, load_golden
from oracle import paropt_oracle as po
from oracle.fused_model import FusedInteriorPoint

CASES = ["ip_quadratic_n257_c3_bfgs", "ip_convex_n300_c5_bfgs", "ip_convex_n300_c5_sr1",
         "ip_quadratic_illcond_n500_c4", "ip_rosenbrock_n100", "ip_convex_n2000_c32_bfgs"]


@pytest.mark.parametrize("name", CASES)
def test_fused_matches_reference(name):
    g, case = load_golden(name)
    a = case["args"]
    prob = po.SepProblem(a["problem"], a["n"], a.get("c", 2), eig_min=a.get("eig_min", 1.0),
                         eig_max=a.get("eig_max", 100.0))
    opts = ip_options_from_case(case)
    opts.pop("write_output_frequency", None)
    ip = FusedInteriorPoint(prob, opts)
    snaps = []
    ip.hook = lambda s, k: snaps.append(s.snapshot())
    ip.optimize()
    window = 8 if "sr1" in name else 25
    for k in range(min(window, len(snaps))):
        p = "it%03d/" % k
        s = snaps[k]
        np.testing.assert_array_equal(s["counters"], g[p + "counters"])
        assert s.get("qn_size", 0) == int(g[p + "qn_size"][0])
        assert abs(s["mu"] - g[p + "mu"][0]) <= 1e-6 * abs(g[p + "mu"][0])
        np.testing.assert_allclose(s["norms"], g[p + "norms"], rtol=1e-6)
        np.testing.assert_allclose(s["z"], g[p + "z"], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(g[p + "z"]).max()))
    if "sr1" not in name:
        np.testing.assert_array_equal(np.array([ip.niter, ip.neval, ip.ngeval]), g["final/counters"])
        assert abs(ip.fobj - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
