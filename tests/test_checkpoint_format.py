"""CPU: the solution-file layout (src/ParOptInteriorPoint.cpp:883-972) decoded by paropt_amd.ParOpt.unpack_checkpoint
from the files the REFERENCE itself wrote (tests/golden/*checkpoint*: one and two MPI ranks, with and without
sparse constraints), against the state the reference dumped at the same iteration."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

CASES = [n for n in golden_names("ip") if "checkpoint" in n]


@pytest.mark.parametrize("name", CASES)
def test_unpack_reference_checkpoint(name, tmp_path):
    from paropt_amd import ParOpt

    g, case = load_golden(name)
    path = str(tmp_path / "ckpt.bin")
    g["checkpoint_bytes"].tofile(path)
    d = ParOpt.unpack_checkpoint(path, full=True)
    a = case["args"]
    assert d["nvars"] == a["n"] and d["ncon"] == a["c"]
    barrier, s, z, x, zl, zu = ParOpt.unpack_checkpoint(path)
    assert barrier == d["barrier"] and len(x) == a["n"] and len(z) == a["c"]
    # the file left behind is the state of iteration 10 (write_output_frequency = 10, 12 iterations)
    p = "it010/"
    assert d["barrier"] == g[p + "mu"][0]
    for key in ("s", "t", "z", "zs", "zt"):
        np.testing.assert_array_equal(d[key], g[p + key])
    # (the two-rank goldens record rank 0's block of each vector: the file must START with it, then hold rank 1's)
    for key in ("x", "zl", "zu"):
        ref = g[p + key]
        assert len(d[key]) == a["n"] and len(ref) in (a["n"], (a["n"] + 1) // 2)
        np.testing.assert_array_equal(d[key][:len(ref)], ref)
    if d["nwcon"] > 0:
        for key in ("zw", "sw"):
            ref = g[p + key]
            np.testing.assert_array_equal(d[key][:len(ref)], ref)
        assert np.all(d["sw"] > 0.0)  # slacks of every rank's block are interior
    assert len(g["checkpoint_bytes"]) == 12 + 8 * (1 + 5 * d["ncon"] + 3 * d["nvars"] + 2 * d["nwcon"])
