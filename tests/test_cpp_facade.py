"""
The C++ facade (include/ParOptAMD.hpp: ParOptVec / ParOptProblem / ParOptLBFGS / ParOptInteriorPoint
with the reference's method names over the C ABI) compiles and links against libparopt_amd.so with
a plain g++; a user problem written the reference's way (host arrays through getArray) reproduces the
reference's Rosenbrock trajectory on the GPU.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


def build(tmp_path, name="rosenbrock_amd"):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".cpp"), "-L" + os.path.join(ROOT, "paropt_amd"),
           "-lparopt_amd", "-Wl,-rpath," + os.path.join(ROOT, "paropt_amd"), "-Wl,-rpath-link,/opt/rocm/lib",
           "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch

    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 2 and "no CPU fallback" in res.stderr


@pytest.mark.gpu
def test_cpp_rosenbrock_matches_reference(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([exe, "nvars=100"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    # the C ABI keeps the reference's default output file name
    assert "iter nobj ngrd nhvc" in open(str(tmp_path / "paropt.out")).read()
    g, _ = load_golden("ip_rosenbrock_n100")
    np.testing.assert_array_equal(np.array([out["niter"], out["neval"], out["ngeval"]]), g["final/counters"])
    assert abs(out["fobj"] - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
    np.testing.assert_allclose(out["xnorm"], g["final/norms"][0], rtol=1e-7)
    np.testing.assert_allclose([out["z0"], out["z1"]], g["final/z"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_cpp_rosenbrock_sparse_constraints_match_reference(tmp_path):
    """The reference example WITH its sparse constraints (nwcon = 5), host callbacks through the
    facade's evalSparseCon / addSparseJacobian / addSparseJacobianTranspose / addSparseInnerProduct."""
    exe = build(tmp_path)
    res = subprocess.run([exe, "nvars=100", "nwcon=5"], capture_output=True, text=True, timeout=300,
                         cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    g, _ = load_golden("ipw_rosenbrock_n100_w5")
    np.testing.assert_array_equal(np.array([out["niter"], out["neval"], out["ngeval"]]), g["final/counters"])
    assert abs(out["fobj"] - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
    np.testing.assert_allclose(out["xnorm"], g["final/norms"][0], rtol=1e-7)
    np.testing.assert_allclose([out["z0"], out["z1"]], g["final/z"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_cpp_optimizer_trust_region(tmp_path):
    """ParOptOptimizer with algorithm = "tr" through the facade: the reference's trust-region
    trajectory on the same problem ends at the same point (golden tr_rosenbrock_n60_bfgs)."""
    exe = build(tmp_path)
    res = subprocess.run([exe, "nvars=60", "algorithm=tr"], capture_output=True, text=True, timeout=600,
                         cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    g, _ = load_golden("tr_rosenbrock_n60_bfgs")
    assert abs(out["fobj"] - g["final/fk"][0]) <= 1e-6 * max(1.0, abs(g["final/fk"][0]))
    np.testing.assert_allclose(out["xnorm"], g["final/xnorm"][0], rtol=1e-6)
    np.testing.assert_allclose([out["z0"], out["z1"]], g["final/z"], rtol=1e-4, atol=1e-6)


def test_sparse_facade_compiles(tmp_path):
    import torch

    exe = build(tmp_path, "sparse_rosenbrock_amd")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 2 and "no MI355X available" in res.stderr


@pytest.mark.gpu
def test_cpp_sparse_rosenbrock_csr_matches_reference(tmp_path):
    """examples/rosenbrock/sparse_rosenbrock.cpp on the facade's ParOptSparseProblem: the reference's own
    trajectory on the same problem (golden ipcsr_rosenbrock_n100_chain2, reference interior point on
    ParOptSparseProblem) ends at the same point with the same counters."""
    exe = build(tmp_path, "sparse_rosenbrock_amd")
    res = subprocess.run([exe, "nvars=100"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    g, _ = load_golden("ipcsr_rosenbrock_n100_chain2")
    np.testing.assert_array_equal(np.array([out["niter"], out["neval"], out["ngeval"]]), g["final/counters"])
    assert abs(out["fobj"] - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
    np.testing.assert_allclose(out["xnorm"], g["final/norms"][0], rtol=1e-7)
    np.testing.assert_allclose([out["z0"], out["z1"]], g["final/z"], rtol=1e-5, atol=1e-6)
    assert "nnz(L)" in out["factor_info"]


def build_c(tmp_path):
    exe = str(tmp_path / "c_abi_quadratic")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_quadratic.c"), "-L" + os.path.join(ROOT, "paropt_amd"),
           "-lparopt_amd", "-Wl,-rpath," + os.path.join(ROOT, "paropt_amd"), "-Wl,-rpath-link,/opt/rocm/lib",
           "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_abi_header_is_c99_and_links(tmp_path):
    """include/paropt_amd.h is plain C (no C++ in the boundary) and every call of the example links."""
    import torch

    exe = build_c(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    res = subprocess.run([exe, "1000"], capture_output=True, text=True)
    assert res.returncode == 2 and "no MI355X available" in res.stderr


@pytest.mark.gpu
def test_c_abi_three_algorithms_agree(tmp_path):
    """ip, tr and mma from plain C on the same convex quadratic reach the same optimum."""
    exe = build_c(tmp_path)
    res = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    f = out["ip"]["fobj"]
    assert abs(out["tr"]["fobj"] - f) <= 1e-4 * max(1.0, abs(f))
    assert abs(out["mma"]["fobj"] - f) <= 1e-3 * max(1.0, abs(f))
    assert out["ip"]["niter"] > 5 and out["tr"]["iters"] > 3 and out["mma"]["sub_iters"] > out["mma"]["iters"]
