"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol that
include/paropt_amd.h declares, the ctypes table covers the header, and the product fails loudly
(no CPU fallback) when no GPU is present."""
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "paropt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(po_[a-z0-9_]+)\s*\(", text)) - {"po_allgather_fn", "po_ip_iteration_fn"})


def test_library_exports_every_declared_symbol():
    import paropt_amd.lib as L

    syms = header_symbols()
    assert len(syms) > 60
    for s in syms:
        assert hasattr(L.lib, s), "libparopt_amd.so does not export %s" % s
    assert set(L.SIGNATURES) == set(syms), set(L.SIGNATURES) ^ set(syms)


def test_version_string():
    import paropt_amd.lib as L

    assert b"gfx950" in L.lib.po_version()


def test_no_cpu_fallback():
    """Without a GPU the context cannot be created and nothing computes on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import paropt_amd as pa

    with pytest.raises(pa.ParOptAMDError) as e:
        pa.Context(0)
    assert e.value.code in (1, 4)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under paropt_amd/ may use it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "paropt_amd")):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_null_handles_are_refused_not_dereferenced():
    """Entry points added in round 5 (state injection, user-written subproblems, ParOptInfeasSubproblem) answer a NULL
    handle with an error code and a message -- no GPU needed to check that they do not dereference it."""
    import ctypes as C

    import paropt_amd.lib as L

    lib = L.lib
    out = L.po_problem()
    assert lib.po_infeas_create(None, 1, 2, C.byref(out)) != 0
    assert lib.po_infeas_set_objective_scaling(None, 1.0) != 0
    assert lib.po_trsub_sync_linear_model(None) != 0
    assert lib.po_trsub_problem(None, C.byref(out)) != 0
    assert lib.po_trsub_destroy(None) == 0  # (like free(NULL))
    assert lib.po_problem_destroy(None) == 0
    assert len(lib.po_last_error()) > 0
