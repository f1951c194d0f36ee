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


REF_EXAMPLE = "/root/reference/examples/rosenbrock/rosenbrock.cpp"
EXAMPLE_AMD = os.path.join(ROOT, "oracle", "_ref", "rosenbrock_example_amd")


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="the reference tree is only present in the build container")
def test_reference_example_recompiles_unchanged(tmp_path):
    """The reference's own example program -- /root/reference/examples/rosenbrock/rosenbrock.cpp, not a copy and not
    edited -- compiles with -Werror-free g++ against include/paropt_compat/ (the reference's header names, MPI_Comm
    communicators) and links against libparopt_amd.so + MPI: the problem class (getArray-style host code, MPI
    reductions, sparse constraints, createQuasiDefMat), ParOptOptions, ParOptOptimizer::addDefaultOptions and the
    whole main() are used as they are."""
    exe = str(tmp_path / "rosenbrock_example_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "paropt_compat"),
           "-I/opt/conda/include", REF_EXAMPLE, "-o", exe, "-static-libstdc++", "-static-libgcc",
           "-L" + os.path.join(ROOT, "paropt_amd"), "-lparopt_amd", "-Wl,-rpath," + os.path.join(ROOT, "paropt_amd"),
           "/opt/conda/lib/libmpi.so", "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath,/opt/conda/lib"]
    subprocess.check_call(cmd)
    import torch

    if not torch.cuda.is_available():  # no GPU here: the program must say so instead of computing on the CPU
        res = subprocess.run([exe], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
        assert "no CPU fallback" in res.stderr or "no HIP device" in res.stderr


def parse_tr_table(text):
    rows, infos = [], []
    for ln in text.splitlines():
        parts = ln.split()
        if len(parts) >= 14 and parts[0].isdigit():
            rows.append([float(v) for v in parts[:13]])
            infos.append(" ".join(parts[14:]))
    return np.array(rows), infos


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXAMPLE_AMD), reason="prebuilt by oracle/Makefile in the build container")
def test_reference_example_program_reproduces_reference_run(tmp_path):
    """The unchanged reference example built against the product (oracle/Makefile target rosenbrock_example_amd)
    runs on the GPU under MPI_Init / MPI_COMM_WORLD and writes the same trust-region table as the same source file
    built against the reference library (tests/golden/example_rosenbrock.npz): iteration count, accept/reject
    pattern and the printed columns to print precision."""
    res = subprocess.run([EXAMPLE_AMD], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "ParOpt time" in res.stdout
    rows, infos = parse_tr_table(open(str(tmp_path / "paropt.tr")).read())
    g = np.load(os.path.join(ROOT, "tests", "golden", "example_rosenbrock.npz"))
    ref = g["table"]
    assert rows.shape == ref.shape, (rows.shape, ref.shape)
    # iter, fobj, infeas, l1, linfty, |x - xk|, tr, rho, mod red., avg z, max z, avg pen., max pen.
    np.testing.assert_array_equal(rows[:, 0], ref[:, 0])
    np.testing.assert_allclose(rows[:, 1], ref[:, 1], rtol=2e-5)            # fobj, 6 printed digits
    np.testing.assert_allclose(rows[:, 6], ref[:, 6], rtol=1e-2)            # trust-region radius
    np.testing.assert_allclose(rows[:, 9:13], ref[:, 9:13], rtol=2e-2, atol=1e-2)
    for col in (2, 3, 4, 5, 7, 8):  # three printed digits; tiny late values are round-off
        np.testing.assert_allclose(rows[:, col], ref[:, col], rtol=3e-2, atol=1e-5)
    assert os.path.exists(str(tmp_path / "paropt.out"))


def test_getarray_pointer_is_the_data(tmp_path):
    """src/ParOptVec.cpp:212-217: no sync call between writes through the getArray pointer and the vector
    operations, in either direction (compile-time check of the abstract ParOptVec interface as well)."""
    src = tmp_path / "live.cpp"
    src.write_text(r"""
#include "ParOptAMD.hpp"
struct P : public ParOptProblem {
  P(po_ctx c) : ParOptProblem(c) { setProblemSizes(1000, 0, 0); }
  void getVarsAndBounds(ParOptVec *, ParOptVec *, ParOptVec *) {}
  int evalObjCon(ParOptVec *, ParOptScalar *, ParOptScalar *) { return 0; }
  int evalObjConGradient(ParOptVec *, ParOptVec *, ParOptVec **) { return 0; }
};
int main() {
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) return 2;
  P *p = new P(ctx);
  p->incref();
  ParOptVec *x = p->createDesignVec(), *y = p->createDesignVec();  // the abstract type with its 11 virtuals
  x->incref();
  y->incref();
  ParOptScalar *xa, *ya;
  int n = x->getArray(&xa);
  y->getArray(&ya);
  for (int i = 0; i < n; i++) xa[i] = 1.0 + i;          // written through the pointer ...
  double s = 0.0;
  for (int i = 0; i < n; i++) s += (1.0 + i) * (1.0 + i);
  int bad = 0;
  bad |= fabs(x->norm() - sqrt(s)) > 1e-9;              // ... seen by the device reduction without a sync
  y->copyValues(x);                                      // device result ...
  bad |= (ya[10] != 11.0) << 1;                          // ... visible through the pointer taken before
  y->scale(2.0);
  xa[0] = 100.0;
  y->axpy(1.0, x);
  bad |= (ya[0] != 102.0 || ya[999] != 3000.0) << 2;
  x->zeroEntries();
  bad |= (xa[5] != 0.0) << 3;
  ParOptVec *vs[2] = {x, y};
  ParOptScalar d[2];
  xa[1] = 1.0;
  y->mdot(vs, 2, d);
  bad |= (d[0] != ya[1]) << 4;
  printf("bad=%d\n", bad);
  return bad;
}
""")
    exe = str(tmp_path / "live")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src),
                           "-L" + os.path.join(ROOT, "paropt_amd"), "-lparopt_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "paropt_amd"), "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    import torch

    if torch.cuda.is_available():
        res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stdout + res.stderr


def test_options_registry_semantics(tmp_path):
    """ParOptOptions of the facade (src/ParOptOptions.h:9-61): typed getters, immediate validation, ranges, the
    three addDefaultOptions entry points -- host only, runs without a GPU."""
    src = tmp_path / "opts.cpp"
    src.write_text(r"""
#include "ParOptAMD.hpp"
int main() {
  ParOptOptions *o = new ParOptOptions();
  o->incref();
  ParOptOptimizer::addDefaultOptions(o);
  int bad = 0, k = 0;
  bad |= (o->getOptionType("algorithm") != ParOptOptions::PAROPT_ENUM_OPTION) << k++;
  bad |= (strcmp(o->getEnumOption("algorithm"), "tr") != 0) << k++;
  bad |= (o->getStringOption("ip_checkpoint_file") != NULL) << k++;
  bad |= (o->getIntOption("qn_subspace_size") != 10) << k++;
  bad |= (o->getFloatOption("abs_res_tol") != 1e-6) << k++;
  bad |= (o->getBoolOption("use_line_search") != 1) << k++;
  bad |= (o->setOption("qn_subspace_size", 100000) == 0) << k++;      // out of range: refused at once
  bad |= (o->getIntOption("qn_subspace_size") != 10) << k++;
  bad |= (o->setOption("abs_res_tol", 3) == 0) << k++;                // wrong type
  bad |= (o->setOption("qn_type", "newton") == 0) << k++;             // not in the enumeration
  bad |= (o->setOption("no_such_option", 1.0) == 0) << k++;
  bad |= (o->setOption("qn_type", "sr1") != 0 || strcmp(o->getEnumOption("qn_type"), "sr1") != 0) << k++;
  bad |= (o->setOption("tr_max_size", 2.0) != 0 || o->getFloatOption("tr_max_size") != 2.0) << k++;
  bad |= (o->setOption("mma_max_iterations", 7) != 0 || o->getIntOption("mma_max_iterations") != 7) << k++;
  int lo = 0, hi = 0, ne = 0;
  const char *const *vals = NULL;
  bad |= (o->getIntRange("max_line_iters", &lo, &hi) != 0 || lo != 1 || hi != 100) << k++;
  bad |= (o->getEnumRange("barrier_strategy", &ne, &vals) != 0 || ne != 4) << k++;
  bad |= (o->isOption("penalty_gamma") != 1 || o->isOption("nope") != 0) << k++;
  int count = 0;
  o->begin();
  do { count += o->getName() != NULL; } while (o->next());
  bad |= (count < 80) << k++;
  ParOptOptions *ip = new ParOptOptions();
  ip->incref();
  ParOptInteriorPoint::addDefaultOptions(ip);
  bad |= (ip->isOption("tr_max_size") != 0 || ip->isOption("barrier_strategy") != 1) << k++;
  printf("bad=%d count=%d\n", bad, count);
  ip->decref();
  o->decref();
  return bad != 0;
}
""")
    exe = str(tmp_path / "opts")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src),
                           "-L" + os.path.join(ROOT, "paropt_amd"), "-lparopt_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "paropt_amd"), "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr


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


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["objects", "optimizer"])
def test_cpp_eigenvalue_example_reproduces_reference_rows(tmp_path, driver):
    """BASELINE config 5 from the user's side in C++: examples/eigenvalue_amd.cpp assembles ParOptLBFGS,
    ParOptCompactEigenApprox, ParOptEigenQuasiNewton, ParOptEigenSubproblem (+ setEigenModelUpdate with a callback
    that fills the directions through getArray) and runs ParOptTrustRegion::optimize(ip) -- or ParOptOptimizer with
    setTrustRegionSubproblem -- the way the reference's examples/eigenvalue/eigenvalue_opt.py:298-308 does; the
    table it prints is the compiled reference's (golden tr_eig_quadratic_n200_c2_N4), row for row."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tr_helpers import compare_tr
    from tr_helpers import parse_tr_table as parse_rows

    exe = build(tmp_path, "eigenvalue_amd")
    g, case = load_golden("tr_eig_quadratic_n200_c2_N4")
    a = case["args"]
    cmd = [exe, "n=%d" % a["n"], "c=%d" % a["c"], "N=%d" % a["eig_N"], "index=%d" % a["eig_index"],
           "curv=%g" % a["eig_curv"], "driver=" + driver, "opt.qn_subspace_size=%d" % a["opt.qn_subspace_size"],
           "opt.tr_max_iterations=%d" % a["tr.tr_max_iterations"]]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    table = parse_rows(res.stdout)
    rows = [table[k] for k in sorted(table)]
    n = compare_tr(g, rows, [], None, 60, check_snaps=False)
    assert n >= 20 and len(rows) == int(g["final/iter_count"][0])
    last = [ln for ln in res.stdout.splitlines() if ln.startswith("final:")][0].split()
    assert abs(float(last[2]) - g["final/fk"][0]) <= 1e-6 * max(1.0, abs(g["final/fk"][0]))
    np.testing.assert_allclose(float(last[4]), g["final/xnorm"][0], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["objects", "optimizer"])
def test_cpp_user_written_trust_region_subproblem(tmp_path, driver):
    """The extension point SURVEY 8b names: a ParOptTrustRegionSubproblem subclass WRITTEN BY THE USER
    (examples/user_subproblem_amd.cpp restates the reference's quadratic model, src/ParOptTrustRegion.cpp:27-466, on the
    facade's vector and quasi-Newton classes) under ParOptTrustRegion(subproblem)->optimize(ip) and under
    ParOptOptimizer::setTrustRegionSubproblem (src/ParOptOptimizer.cpp:226-237): the table it prints is the compiled
    reference's (golden tr_quadratic_n200_c3_bfgs), row for row."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tr_helpers import compare_tr
    from tr_helpers import parse_tr_table as parse_rows
    from test_gpu_tr import TR_INEXACT_ROWS

    exe = build(tmp_path, "user_subproblem_amd")
    name = "tr_quadratic_n200_c3_bfgs"
    g, case = load_golden(name)
    a = case["args"]
    cmd = [exe, "n=%d" % a["n"], "c=%d" % a["c"], "driver=" + driver]
    for k, v in a.items():
        if k.startswith("opt.") and k != "opt.write_output_frequency":
            cmd.append("%s=%s" % (k, v))
        if k.startswith("tr."):
            cmd.append("opt.%s=%s" % (k[3:], v))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    table = parse_rows(res.stdout)
    rows = [table[k] for k in sorted(table)]
    n = compare_tr(g, rows, [], None, 60, check_snaps=False, inexact_rows=TR_INEXACT_ROWS.get(name, set()))
    assert n >= 12 and len(rows) == int(g["final/iter_count"][0])
    last = [ln for ln in res.stdout.splitlines() if ln.startswith("final:")][0].split()
    assert abs(float(last[2]) - g["final/fk"][0]) <= 1e-6 * max(1.0, abs(g["final/fk"][0]))


@pytest.mark.gpu
def test_cpp_infeas_subproblem_over_user_written_and_library_subproblems(tmp_path):
    """ParOptInfeasSubproblem (src/ParOptTrustRegion.h:293-374) as a facade class: built by hand over the USER-WRITTEN
    quadratic subproblem and over the library's ParOptQuadraticSubproblem of the same problem (driver=infeas of
    examples/user_subproblem_amd.cpp), the steering LP of the first trust-region iteration comes out the same."""
    exe = build(tmp_path, "user_subproblem_amd")
    res = subprocess.run([exe, "n=200", "c=3", "driver=infeas", "opt.abs_res_tol=1e-9", "opt.tr_init_size=0.1"],
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr
    lines = {ln.split()[1].rstrip(":"): ln.split() for ln in res.stdout.splitlines() if ln.startswith("infeas ")}
    assert set(lines) == {"user", "library"}, res.stdout + res.stderr

    def nums(tok):
        return [float(v) for v in tok if v[0] in "-0123456789" and v[-1] in "0123456789"]

    u, lb = np.array(nums(lines["user"])), np.array(nums(lines["library"]))
    assert u.shape == lb.shape and u.size == 3 + 3 + 3
    assert abs(u[1]) > 1e-3 and abs(u[2] - 0.1) < 1e-6  # a real step, at the trust-region bound somewhere
    np.testing.assert_allclose(u, lb, rtol=1e-8, atol=1e-10)


def test_eigenvalue_example_compiles_with_the_reference_header_names(tmp_path):
    """The class set of src/ParOptTrustRegion.h / src/ParOptCompactEigenvalueApprox.h is there under the reference's
    header names (include/paropt_compat, MPI_Comm communicators): a translation unit that includes only those headers
    and touches every class of VERDICT r3 missing #1 compiles."""
    src = tmp_path / "uses_tr_classes.cpp"
    src.write_text(
        '#include "ParOptCompactEigenvalueApprox.h"\n#include "ParOptTrustRegion.h"\n#include "ParOptOptimizer.h"\n'
        "static void upd(void *, ParOptVec *, ParOptCompactEigenApprox *a) { ParOptScalar *c0; a->getApproximation(&c0, "
        "NULL, NULL, NULL, NULL, NULL); }\n"
        "void assemble(ParOptProblem *p, ParOptOptions *o) {\n"
        "  ParOptLBFGS *qn = new ParOptLBFGS(p, 10);\n"
        "  ParOptCompactEigenApprox *ap = new ParOptCompactEigenApprox(p, 4);\n"
        "  ParOptEigenQuasiNewton *eq = new ParOptEigenQuasiNewton(qn, ap, 0);\n"
        "  ParOptEigenSubproblem *es = new ParOptEigenSubproblem(p, eq);\n"
        "  es->setEigenModelUpdate(NULL, upd);\n"
        "  ParOptTrustRegionSubproblem *sub = es;\n"
        "  ParOptQuadraticSubproblem *qs = new ParOptQuadraticSubproblem(p, qn);\n"
        "  ParOptInteriorPoint *ip = new ParOptInteriorPoint(sub, o);\n"
        "  ParOptTrustRegion *tr = new ParOptTrustRegion(sub, o);\n"
        "  tr->setPenaltyGamma(10.0); tr->initialize(); tr->optimize(ip);\n"
        "  ParOptVec *x; tr->getOptimizedPoint(&x);\n"
        "  ParOptOptimizer *opt = new ParOptOptimizer(p, o); opt->setTrustRegionSubproblem(qs); opt->optimize();\n"
        "  ip->checkGradients(1e-6); ip->setBFGSUpdateType(PAROPT_DAMPED_UPDATE); ip->setUseDiagHessian(0);\n"
        "  ip->checkMeritFuncGradient(NULL, 1e-6);\n"
        "  ParOptInfeasSubproblem *inf = new ParOptInfeasSubproblem(sub, ParOptInfeasSubproblem::PAROPT_LINEAR_OBJECTIVE,\n"
        "      ParOptInfeasSubproblem::PAROPT_LINEAR_CONSTRAINT);\n"
        "  inf->setObjectiveScaling(0.5); ParOptProblem *as_problem = inf; (void)as_problem;\n"
        "}\n")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-c", "-I" + os.path.join(ROOT, "include", "paropt_compat"),
                           "-I/opt/conda/include", str(src), "-o", str(tmp_path / "uses_tr_classes.o")])


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
