#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REAL reference.

Runs oracle/_ref/ref_driver (the unmodified reference sources of /root/reference compiled
by oracle/Makefile) on small instances of the BASELINE.json workloads and stores inputs
(the case parameters; all data are a pure function of them through the counter hash) and
expected outputs as compressed .npz fixtures.  Only this container can run it (the
reference does not travel to the GPU box); the fixtures are committed.

usage: python oracle/make_golden.py [--only NAME]
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
MPIEXEC = "/opt/conda/bin/mpiexec"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_rec(path):
    """Parse the record format documented at the top of oracle/ref_driver.cpp."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from("<i", data, pos)
        pos += 4
        name = data[pos : pos + ln].decode()
        pos += ln
        dt, cnt = struct.unpack_from("<iq", data, pos)
        pos += 12
        if dt == 0:
            arr = np.frombuffer(data, dtype="<f8", count=cnt, offset=pos).copy()
            pos += 8 * cnt
        else:
            arr = np.frombuffer(data, dtype="<i4", count=cnt, offset=pos).copy()
            pos += 4 * cnt
        out[name] = arr
    return out


def run_driver(mode, args, ranks=1):
    env = dict(os.environ)
    env["MKL_NUM_THREADS"] = "1"
    env["PATH"] = "/opt/conda/bin:" + env.get("PATH", "")
    cmd = [DRIVER, mode] + ["%s=%s" % (k, v) for k, v in args.items()]
    if ranks > 1:
        cmd = [MPIEXEC, "-n", str(ranks)] + cmd
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir())
    if res.returncode not in (0,):
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("ref_driver failed: %s" % " ".join(cmd))
    return res.stdout


# The option sets of the reference examples (examples/random_quadratic/random_quadratic.py:92-102,
# examples/random_convex/random_convex.py:116-126)
EXAMPLE_OPTS = {
    "opt.abs_res_tol": 1e-8,
    "opt.starting_point_strategy": "affine_step",
    "opt.barrier_strategy": "monotone",
    "opt.start_affine_multiplier_min": 0.01,
    "opt.penalty_gamma": 1000.0,
    "opt.write_output_frequency": 1,
}

CASES = {}


def case(name, mode, ranks=1, **args):
    CASES[name] = (mode, ranks, args)


# --- vector reductions ---
for n in (1, 63, 1000, 4097):
    for nv in (1, 8, 33):
        case("vecops_n%d_nv%d" % (n, nv), "vecops", n=n, nvecs=nv)
case("vecops_n4097_nv8_r2", "vecops", ranks=2, n=4097, nvecs=8)
case("vecops_n1000_nv33_r4", "vecops", ranks=4, n=1000, nvecs=33)

# --- quasi-Newton sequences (item 2) ---
for typ, upd in (("bfgs", "skip"), ("bfgs", "damped"), ("sr1", "skip")):
    for msub in (3, 10):
        case("qn_%s_%s_m%d" % (typ, upd, msub), "qn", type=typ, update=upd, msub=msub, n=200, steps=25)
case("qn_bfgs_damped_m3_sts", "qn", type="bfgs", update="damped", msub=3, n=200, steps=25, diag="yts_over_sts")

# --- interior-point trajectories (items 3-5) ---
ip_common = dict(EXAMPLE_OPTS)
case(
    "ip_quadratic_n257_c3_bfgs",
    "ip",
    problem="quadratic",
    n=257,
    c=3,
    dump_vecs_every=5,
    kat_iter=12,
    **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}),
)
case(
    "ip_quadratic_n1000_c8_bfgs20",
    "ip",
    problem="quadratic",
    n=1000,
    c=8,
    dump_vecs_every=20,
    kat_iter=30,
    **dict(ip_common, **{"opt.qn_subspace_size": 20, "opt.qn_type": "bfgs", "opt.max_major_iters": 150}),
)
case(
    "ip_quadratic_illcond_n500_c4",
    "ip",
    problem="quadratic",
    n=500,
    c=4,
    eig_max=1e5,
    dump_vecs_every=25,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 80}),
)
case(
    "ip_quadratic_damped_n300_c2",
    "ip",
    problem="quadratic",
    n=300,
    c=2,
    dump_vecs_every=10,
    **dict(
        ip_common,
        **{
            "opt.qn_subspace_size": 6,
            "opt.qn_type": "bfgs",
            "opt.qn_update_type": "damped_update",
            "opt.max_major_iters": 60,
        },
    ),
)
case(
    "ip_convex_n300_c5_bfgs",
    "ip",
    problem="convex",
    n=300,
    c=5,
    dump_vecs_every=10,
    kat_iter=8,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 80}),
)
case(
    "ip_convex_n300_c5_sr1",
    "ip",
    problem="convex",
    n=300,
    c=5,
    dump_vecs_every=5,
    kat_iter=8,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 25}),
)
case(
    "ip_convex_n2000_c32_sr1",
    "ip",
    problem="convex",
    n=2000,
    c=32,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 20}),
)
case(
    "ip_convex_n2000_c32_bfgs",
    "ip",
    problem="convex",
    n=2000,
    c=32,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}),
)
case(
    "ip_convex_n2000_c32_bfgs_r2",
    "ip",
    ranks=2,
    problem="convex",
    n=2000,
    c=32,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}),
)
# panels wider than one launch of the product's kernels (c + k > 80: blocked Gram; > 96: collapsed panel sums) --
# the reference has no limit on the number of dense constraints or on the quasi-Newton width
case(
    "ip_convex_n2000_c100_bfgs10",
    "ip",
    problem="convex",
    n=2000,
    c=100,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 40}),
)
case(
    "ip_quadratic_n1500_c70_bfgs10",
    "ip",
    problem="quadratic",
    n=1500,
    c=70,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 40}),
)
case(
    "ip_convex_n2000_c90_sr1",
    "ip",
    problem="convex",
    n=2000,
    c=90,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 20}),
)
# config 1: examples/rosenbrock/rosenbrock.cpp with algorithm=ip, w=0
case(
    "ip_rosenbrock_n100",
    "ip",
    problem="rosenbrock",
    n=100,
    dump_vecs_every=10,
    **{
        "opt.qn_subspace_size": 10,
        "opt.qn_type": "bfgs",
        "opt.abs_res_tol": 1e-6,
        "opt.barrier_strategy": "monotone",
        "opt.write_output_frequency": 1,
        "opt.max_major_iters": 120,
    },
)
case(
    "ip_quadratic_compfrac_n200_c2",
    "ip",
    problem="quadratic",
    n=200,
    c=2,
    dump_vecs_every=0,
    **dict(
        ip_common,
        **{
            "opt.qn_subspace_size": 5,
            "opt.barrier_strategy": "complementarity_fraction",
            "opt.max_major_iters": 60,
        },
    ),
)
case(
    "ip_quadratic_nolinesearch_n200_c2",
    "ip",
    problem="quadratic",
    n=200,
    c=2,
    dump_vecs_every=0,
    **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.use_line_search": 0, "opt.max_major_iters": 40}),
)


# solution-file format: written at iterations 0 and 10, the file left behind is iteration 10's
case(
    "ip_quadratic_checkpoint_n130_c3",
    "ip",
    problem="quadratic",
    n=130,
    c=3,
    dump_vecs_every=10,
    checkpoint=1,
    **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.write_output_frequency": 10, "opt.max_major_iters": 12}),
)
case(
    "ipw_convex_checkpoint_n120_c2_w20",
    "ip",
    problem="convex",
    n=120,
    c=2,
    nwcon=20,
    nw=5,
    nwstart=4,
    nwskip=0,
    dump_vecs_every=10,
    checkpoint=1,
    **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.write_output_frequency": 10, "opt.max_major_iters": 12}),
)
# the same file written by TWO MPI ranks (MPI_File_write_at_all at var_range / wcon_range offsets, :929-968):
# one file in the layout of the concatenated problem; the CSR case has rank-local sparse constraints
case("ip_quadratic_checkpoint_n131_c3_r2", "ip", ranks=2, problem="quadratic", n=131, c=3, dump_vecs_every=10,
     checkpoint=1,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.write_output_frequency": 10, "opt.max_major_iters": 12}))
case("ipcsr_convex_checkpoint_n121_c2_chain2_r2", "ip", ranks=2, problem="convex", n=121, c=2, chain_span=2,
     chain_stride=1, dump_vecs_every=10, checkpoint=1,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.write_output_frequency": 10, "opt.max_major_iters": 12}))
# --- weighting (sparse, block-diagonal) constraints: SURVEY 8f rank 1 / config 4 ---
# the reference example itself: examples/rosenbrock/rosenbrock.cpp:219-222 (nwcon=5, nw=5, start 1, skip 1)
case(
    "ipw_rosenbrock_n100_w5",
    "ip",
    problem="rosenbrock",
    n=100,
    nwcon=5,
    nw=5,
    nwstart=1,
    nwskip=1,
    dump_vecs_every=10,
    **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-6,
       "opt.barrier_strategy": "monotone", "opt.write_output_frequency": 1, "opt.max_major_iters": 150},
)
# config-4 style: convex objective, dense constraints + one weighting constraint per group of 5
case(
    "ipw_convex_n400_c4_w80",
    "ip",
    problem="convex",
    n=400,
    c=4,
    nwcon=80,
    nw=5,
    nwstart=0,
    nwskip=0,
    dump_vecs_every=10,
    **dict(ip_common, **{"opt.qn_subspace_size": 8, "opt.qn_type": "bfgs", "opt.max_major_iters": 100}),
)
# nwblock > 1: blocks of three constraints on the same group of six variables with different weights
# (oracle/ref_driver.cpp SepProblem::wgt), so that Aw D^-1 Aw^T has dense 3 x 3 diagonal blocks
case("ipw_rosenbrock_n240_w60_nwblock3", "ip", problem="rosenbrock", n=240, nwcon=60, nw=6, nwstart=0, nwskip=2,
     nwblock=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs", "opt.max_major_iters": 120}))
case("ipw_quadratic_n240_c2_w60_nwblock3", "ip", problem="quadratic", n=240, c=2, nwcon=60, nw=6, nwstart=0,
     nwskip=2, nwblock=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs", "opt.max_major_iters": 120}))
# equality weighting constraints (sum of each group = 1), partial coverage with gaps
case(
    "ipw_convex_n300_c2_w30_eq",
    "ip",
    problem="convex",
    n=300,
    c=2,
    nwcon=30,
    nw=4,
    nwstart=3,
    nwskip=5,
    nwineq=0,
    dump_vecs_every=10,
    **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs", "opt.max_major_iters": 100}),
)
# weighting constraints with the other barrier strategies / norms / the SR1 update
for tag, extra in (
    ("mpc", {"opt.barrier_strategy": "mehrotra_predictor_corrector"}),
    ("mehrotra", {"opt.barrier_strategy": "mehrotra"}),
    ("l2", {"opt.norm_type": "l2"}),
    ("l1_compfrac", {"opt.norm_type": "l1", "opt.barrier_strategy": "complementarity_fraction"}),
    ("sr1", {"opt.qn_type": "sr1"}),
):
    case(
        "ipw_convex_n240_c3_w40_%s" % tag,
        "ip",
        problem="convex",
        n=240,
        c=3,
        nwcon=40,
        nw=4,
        nwstart=2,
        nwskip=2,
        nwineq=25,
        dump_vecs_every=10,
        **dict(ip_common, **dict({"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs",
                                  "opt.max_major_iters": 60}, **extra)),
    )
# --- CSR form of the sparse constraints: the reference's ParOptSparseProblem with the
# overlapping chain constraints of examples/rosenbrock/sparse_rosenbrock.cpp generalised to span / stride;
# the S-solve is oracle/ref_driver.cpp's dense LAPACK quasi-definite matrix (the reference's sparse Cholesky
# needs METIS, absent here), everything else is the reference.
case("ipcsr_rosenbrock_n100_chain2", "ip", problem="rosenbrock", n=100, chain_span=2, chain_stride=1,
     dump_vecs_every=10,
     **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-6,
        "opt.barrier_strategy": "monotone", "opt.write_output_frequency": 1, "opt.max_major_iters": 150})
case("ipcsr_convex_n240_c3_chain3s2_rev", "ip", problem="convex", n=240, c=3, chain_span=3, chain_stride=2,
     chain_reverse=1, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs", "opt.max_major_iters": 80}))
case("ipcsr_quadratic_n200_c2_chain4s4", "ip", problem="quadratic", n=200, c=2, chain_span=4, chain_stride=4,
     dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.qn_type": "bfgs", "opt.max_major_iters": 80}))
for tag, extra in (
    ("mpc", {"opt.barrier_strategy": "mehrotra_predictor_corrector"}),
    ("sr1", {"opt.qn_type": "sr1"}),
    ("l2_least_squares", {"opt.norm_type": "l2", "opt.starting_point_strategy": "least_squares_multipliers"}),
):
    case("ipcsr_convex_n200_c2_chain5s3_%s" % tag, "ip", problem="convex", n=200, c=2, chain_span=5,
         chain_stride=3, dump_vecs_every=10,
         **dict(ip_common, **dict({"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs",
                                   "opt.max_major_iters": 60}, **extra)))
case("ipcsr_convex_n240_c2_chain2_r2", "ip", ranks=2, problem="convex", n=240, c=2, chain_span=2, chain_stride=1,
     dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}))
# second-order branches WITH sparse constraints: block form (weighting, linear) and CSR form (chain, the
# constraint Hessian 2 zw_i enters the Lagrangian)
hvw = {"opt.use_hvec_product": 1, "opt.gmres_subspace_size": 15, "opt.nk_switch_tol": 1e3, "opt.max_gmres_rtol": 1.0}
case("ipw_quadratic_hvec_n300_c3_w40", "ip", problem="quadratic", n=300, c=3, nwcon=40, nw=5, nwstart=2, nwskip=1,
     dump_vecs_every=10,
     **dict(ip_common, **dict(hvw, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80})))
case("ipw_rosenbrock_hvec_n100_w5", "ip", problem="rosenbrock", n=100, nwcon=5, nw=5, nwstart=1, nwskip=1,
     dump_vecs_every=10,
     **dict({"opt.qn_subspace_size": 10, "opt.abs_res_tol": 1e-6, "opt.write_output_frequency": 1,
             "opt.max_major_iters": 150}, **dict(hvw, **{"opt.nk_switch_tol": 1.0, "opt.max_gmres_rtol": 0.5})))
case("ipcsr_convex_hvec_n200_c2_chain3s2", "ip", problem="convex", n=200, c=2, chain_span=3, chain_stride=2,
     dump_vecs_every=10,
     **dict(ip_common, **dict(hvw, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80})))
case("ipcsr_quadratic_hvec_n150_c2_chain2", "ip", problem="quadratic", n=150, c=2, chain_span=2, chain_stride=1,
     dump_vecs_every=10,
     **dict(ip_common, **dict(hvw, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 80})))
case("ipcsr_convex_diaghess_n200_c2_chain3s2", "ip", problem="convex", n=200, c=2, chain_span=3, chain_stride=2,
     dump_vecs_every=10,
     **dict(ip_common, **{"opt.use_diag_hessian": 1, "opt.qn_subspace_size": 6, "opt.max_major_iters": 80}))
# problems that declare no upper / no lower bounds (useUpperBounds() / useLowerBounds() = 0)
case("ip_quadratic_noupper_n200_c2", "ip", problem="quadratic", n=200, c=2, use_upper=0, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 80}))
case("ip_convex_nolower_n200_c2", "ip", problem="convex", n=200, c=2, use_lower=0, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 25}))
# --- Hessian-vector products: inexact Newton-Krylov steps (computeKKTGMRESStep :5796-6191) and the
# diagonal-Hessian variant.  nk_switch_tol / max_gmres_rtol are opened up so that
# the GMRES branch is taken early and often.
hv = {"opt.use_hvec_product": 1, "opt.gmres_subspace_size": 15, "opt.nk_switch_tol": 1e3, "opt.max_gmres_rtol": 1.0}
case("ip_convex_hvec_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **dict(hv, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80})))
case("ip_quadratic_hvec_n257_c3", "ip", problem="quadratic", n=257, c=3, dump_vecs_every=10,
     **dict(ip_common, **dict(hv, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 80})))
case("ip_rosenbrock_hvec_n100", "ip", problem="rosenbrock", n=100, dump_vecs_every=10,
     **dict({"opt.qn_subspace_size": 10, "opt.abs_res_tol": 1e-6, "opt.write_output_frequency": 1,
             "opt.max_major_iters": 150}, **dict(hv, **{"opt.nk_switch_tol": 1.0, "opt.max_gmres_rtol": 0.5})))
case("ip_convex_hvec_noprecon_n200_c2", "ip", problem="convex", n=200, c=2, dump_vecs_every=10,
     **dict(ip_common, **dict(hv, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80,
                                     "opt.use_qn_gmres_precon": 0, "opt.gmres_subspace_size": 30})))
case("ip_convex_diaghess_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.use_diag_hessian": 1, "opt.qn_subspace_size": 6, "opt.max_major_iters": 80}))
case("ip_rosenbrock_diaghess_n100", "ip", problem="rosenbrock", n=100, dump_vecs_every=10,
     **{"opt.use_diag_hessian": 1, "opt.qn_subspace_size": 10, "opt.abs_res_tol": 1e-6,
        "opt.write_output_frequency": 1, "opt.max_major_iters": 150})
# the two defects the random sweep of round 3 found, pinned against the reference itself: an odd n under the sequential
# linear method (zero diagonal: the pad element of the re-formed Dinv), and the sequential linear method together with
# use_diag_hessian (the diagonal is never evaluated, :4920-4949) on a problem with sparse constraints
case("ip_convex_n2049_c3_seqlin", "ip", problem="convex", n=2049, c=3, dump_vecs_every=6,
     **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-8, "opt.start_affine_multiplier_min": 0.01,
        "opt.max_major_iters": 12, "opt.barrier_strategy": "monotone", "opt.sequential_linear_method": 1,
        "opt.write_output_frequency": 1})
case("ip_quadratic_n129_c17_w20_seqlin_diaghess", "ip", problem="quadratic", n=129, c=17, nwcon=20, nw=2, nwstart=5,
     nwskip=1, nwineq=20, dump_vecs_every=4,
     **{"opt.qn_subspace_size": 2, "opt.qn_type": "sr1", "opt.abs_res_tol": 1e-8, "opt.start_affine_multiplier_min": 0.01,
        "opt.max_major_iters": 8, "opt.norm_type": "l2", "opt.sequential_linear_method": 1, "opt.use_diag_hessian": 1,
        "opt.qn_sigma": 1.0, "opt.starting_point_strategy": "affine_step", "opt.penalty_gamma": 1000.0,
        "opt.write_output_frequency": 1})
case("ip_quadratic_n511_c2_seqlin_mpc", "ip", problem="quadratic", n=511, c=2, dump_vecs_every=6,
     **{"opt.qn_subspace_size": 5, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-8, "opt.start_affine_multiplier_min": 0.01,
        "opt.max_major_iters": 10, "opt.barrier_strategy": "mehrotra_predictor_corrector",
        "opt.sequential_linear_method": 1, "opt.write_output_frequency": 1})
# --- method of moving asymptotes (ParOptOptimizer algorithm = "mma", src/ParOptMMA.cpp) ---
case("mma_convex_n300_c3", "mma", problem="convex", n=300, c=3, **{"mma.mma_max_iterations": 30})
case("mma_quadratic_n200_c2", "mma", problem="quadratic", n=200, c=2, **{"mma.mma_max_iterations": 25})
case("mma_rosenbrock_n60", "mma", problem="rosenbrock", n=60, **{"mma.mma_max_iterations": 25, "opt.abs_res_tol": 1e-7})
case("mma_convex_n200_c2_linearized", "mma", problem="convex", n=200, c=2,
     **{"mma.mma_max_iterations": 25, "mma.mma_use_constraint_linearization": 1, "mma.mma_bound_relax": 1e-4})
case("mma_convex_n200_c2_w40", "mma", problem="convex", n=200, c=2, nwcon=40, nw=5, nwstart=0, nwskip=0,
     **{"mma.mma_max_iterations": 20})
case("mma_csr_convex_n150_c2_chain2s2", "mma", problem="convex", n=150, c=2, chain_span=2, chain_stride=2,
     **{"mma.mma_max_iterations": 15})
# --- trust-region driver: ParOptOptimizer's algorithm="tr" set-up ---
tr_common = {"opt.qn_subspace_size": 5, "tr.tr_max_iterations": 60}
case("tr_quadratic_n200_c3_bfgs", "tr", problem="quadratic", n=200, c=3, dump_vecs_every=10, **tr_common)
case("tr_convex_n300_c4_bfgs", "tr", problem="convex", n=300, c=4, dump_vecs_every=10,
     **dict(tr_common, **{"tr.tr_max_size": 0.5, "tr.tr_init_size": 0.05}))
case("tr_rosenbrock_n60_bfgs", "tr", problem="rosenbrock", n=60, dump_vecs_every=10,
     **dict(tr_common, **{"opt.qn_subspace_size": 10, "tr.tr_max_iterations": 80}))
# L-SR1: Rosenbrock, whose first steps are not collinear (on the separable quadratic the first three
# steps all sit on the trust-region box in the same direction, which makes the SR1 compact matrix
# exactly rank one and the reference's own trajectory a round-off artefact)
case("tr_rosenbrock_n60_sr1", "tr", problem="rosenbrock", n=60, dump_vecs_every=10,
     **dict(tr_common, **{"opt.qn_type": "sr1", "opt.qn_subspace_size": 4, "tr.tr_max_iterations": 40}))
case("tr_quadratic_n150_c2_fixedgamma", "tr", problem="quadratic", n=150, c=2, dump_vecs_every=10,
     **dict(tr_common, **{"tr.tr_adaptive_gamma_update": 0, "opt.penalty_gamma": 50.0}))
case("tr_convex_n200_c2_w40", "tr", problem="convex", n=200, c=2, nwcon=40, nw=5, nwstart=0, nwskip=0,
     dump_vecs_every=10, **dict(tr_common, **{"tr.tr_max_size": 0.5, "tr.tr_init_size": 0.05}))
# the trust-region driver over a problem in the CSR form: pins the stored-value semantics of
# ParOptSparseProblem::evalSparseCon / addSparseJacobian under trial-point evaluations
case("tr_csr_convex_n120_c2_chain3s2", "tr", problem="convex", n=120, c=2, chain_span=3, chain_stride=2,
     dump_vecs_every=10, **dict(tr_common, **{"tr.tr_max_size": 0.5, "tr.tr_init_size": 0.05, "tr.tr_max_iterations": 40}))
case("tr_csr_rosenbrock_n60_chain2", "tr", problem="rosenbrock", n=60, chain_span=2, chain_stride=1,
     dump_vecs_every=10, **dict(tr_common, **{"opt.qn_subspace_size": 10, "tr.tr_max_iterations": 60}))
# option combinations of the trust-region driver drawn at random (round 3: the random sweep of the interior point
# compares with the oracle; the oracle's trust-region driver is not a reliable comparator on drawn cases -- its
# subproblem solves end on other round-off level tests than the reference's -- so these come from the reference)
tr_rand = {"tr.tr_max_iterations": 12}
case("tr_rand_convex_n257_c2_eta01", "tr", problem="convex", n=257, c=2, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 3, "opt.qn_update_type": "damped_update", "tr.tr_init_size": 0.5,
                        "tr.tr_eta": 0.1, "tr.tr_max_size": 0.5}))
case("tr_rand_quadratic_n129_c4_fixedgamma100", "tr", problem="quadratic", n=129, c=4, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 5, "tr.tr_adaptive_gamma_update": 0, "opt.penalty_gamma": 100.0}))
case("tr_rand_rosenbrock_n127_eta05", "tr", problem="rosenbrock", n=127, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 8, "tr.tr_eta": 0.5, "tr.tr_max_size": 2.0}))
case("tr_rand_quadratic_n300_c8_subcon", "tr", problem="quadratic", n=300, c=8, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 8, "opt.qn_update_type": "damped_update",
                        "tr.tr_adaptive_constraint": "subproblem_constraint"}))
case("tr_rand_convex_n200_c3_constobj", "tr", problem="convex", n=200, c=3, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 2, "tr.tr_adaptive_objective": "constant_objective",
                        "tr.tr_init_size": 0.05}))
case("tr_rand_quadratic_n65_c1_subobj", "tr", problem="quadratic", n=65, c=1, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_subspace_size": 2, "tr.tr_adaptive_objective": "subproblem_objective",
                        "tr.tr_init_size": 0.05, "opt.penalty_gamma": 10.0}))
case("tr_rand_eig_quadratic_n257_c3_N5_subcon", "tr", problem="quadratic", n=257, c=3, eig_N=5, eig_index=2, eig_curv=1.0,
     dump_vecs_every=4, **dict(tr_rand, **{"opt.qn_subspace_size": 3, "tr.tr_adaptive_constraint": "subproblem_constraint",
                                            "tr.tr_eta": 0.1}))
case("tr_rand_convex_n255_c2_w51_fixedgamma", "tr", problem="convex", n=255, c=2, nwcon=51, nw=4, nwstart=0, nwskip=1,
     dump_vecs_every=4, **dict(tr_rand, **{"opt.qn_subspace_size": 4, "tr.tr_adaptive_gamma_update": 0,
                                            "opt.penalty_gamma": 100.0, "tr.tr_init_size": 0.05}))
case("tr_rand_rosenbrock_n129_sr1_eta01", "tr", problem="rosenbrock", n=129, dump_vecs_every=4,
     **dict(tr_rand, **{"opt.qn_type": "sr1", "opt.qn_subspace_size": 5, "tr.tr_eta": 0.1, "tr.tr_max_size": 2.0}))
# the metric's configuration (config 3 shape: convex objective, c = 32, L-SR1(10)) under the trust-region driver,
# where L-SR1 makes progress, at n = 1e5 on four MPI ranks; subproblem solves capped at 200 interior-
# point iterations as the reference's trust-region examples set it
case("tr_convex_n100000_c32_sr1_r4", "tr", ranks=4, problem="convex", n=100000, c=32, dump_vecs_every=4, vec_stride=50,
     **dict(tr_common, **{"opt.qn_type": "sr1", "opt.qn_subspace_size": 10, "tr.tr_max_size": 0.5,
                          "tr.tr_init_size": 0.05, "tr.tr_max_iterations": 12, "opt.max_major_iters": 200}))
# filter globalisation (filterOptimize :1690-2210)
case("tr_filter_quadratic_n200_c3", "tr", problem="quadratic", n=200, c=3, dump_vecs_every=10,
     **dict(tr_common, **{"tr.tr_accept_step_strategy": "filter_method"}))
case("tr_filter_rosenbrock_n60", "tr", problem="rosenbrock", n=60, dump_vecs_every=10,
     **dict(tr_common, **{"tr.tr_accept_step_strategy": "filter_method", "opt.qn_subspace_size": 10,
                          "tr.filter_has_feas_restore_phase": 0}))
# compact eigenvalue subproblem (config 5 shape): constraint 0 modelled with N curvature directions
case("tr_eig_quadratic_n200_c2_N4", "tr", problem="quadratic", n=200, c=2, eig_N=4, eig_index=0, eig_curv=2.0,
     dump_vecs_every=10, **tr_common)
case("tr_eig_convex_n300_c3_N6", "tr", problem="convex", n=300, c=3, eig_N=6, eig_index=1, eig_curv=0.5,
     dump_vecs_every=10, **dict(tr_common, **{"tr.tr_max_size": 0.5, "tr.tr_init_size": 0.05}))
for strat in ("mehrotra", "mehrotra_predictor_corrector"):
    case(
        "ip_quadratic_%s_n300_c3" % ("mpc" if "corrector" in strat else "mehrotra"),
        "ip",
        problem="quadratic",
        n=300,
        c=3,
        dump_vecs_every=10,
        **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.barrier_strategy": strat, "opt.max_major_iters": 80}),
    )
# the option set of examples/rosenbrock/rosenbrock.cpp:234-242 (barrier_strategy = mehrotra)
case(
    "ip_rosenbrock_mehrotra_n100",
    "ip",
    problem="rosenbrock",
    n=100,
    dump_vecs_every=10,
    **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-6,
       "opt.barrier_strategy": "mehrotra", "opt.write_output_frequency": 1, "opt.max_major_iters": 150},
)
for nt in ("l1", "l2"):
    case(
        "ip_quadratic_norm_%s_n300_c3" % nt,
        "ip",
        problem="quadratic",
        n=300,
        c=3,
        dump_vecs_every=0,
        **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.norm_type": nt, "opt.max_major_iters": 80}),
    )

# --- SURVEY 8a' integer bookkeeping: bound repairs of initAndCheckDesignAndBounds (check_flag bits, :4290-4344)
# and clamp events (:3150-3190, 4177-4195).  bounds_mode: see oracle/ref_driver.cpp SepProblem.
case("ip_convex_badbounds7_n300_c3", "ip", problem="convex", n=300, c=3, bounds_mode=7, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 60}))
case("ip_quadratic_badbounds2_n200_c2", "ip", problem="quadratic", n=200, c=2, bounds_mode=2, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 60}))
case("ip_convex_badbounds5_n201_c2_r2", "ip", ranks=2, problem="convex", n=201, c=2, bounds_mode=5, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 60}))
# a coarse design_precision makes the clamps of the trial point and of the multipliers fire every few iterations
case("ip_convex_clamp_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=5,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 40, "opt.design_precision": 3e-2}))
case("ip_convex_clamp1e1_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=5,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 40, "opt.design_precision": 1e-1}))
case("ip_quadratic_clamp_n200_c2", "ip", problem="quadratic", n=200, c=2, dump_vecs_every=5,
     **dict(ip_common, **{"opt.qn_subspace_size": 5, "opt.max_major_iters": 40, "opt.design_precision": 5e-2}))
# --- option branches with real control flow: backtracking line search (:4078), periodic Hessian
# reset (:4611), relative function test (:4644), qn_sigma (:1477, 1871), no starting-point strategy, sequential
# linear method without a quasi-Newton object, abs_step_tol (only ever stored, :4587, 4880, 4996)
case("ip_rosenbrock_backtrack_n100", "ip", problem="rosenbrock", n=100, dump_vecs_every=10,
     **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.abs_res_tol": 1e-6, "opt.write_output_frequency": 1,
        "opt.max_major_iters": 150, "opt.use_backtracking_alpha": 1})
case("ip_convex_backtrack_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 60, "opt.use_backtracking_alpha": 1}))
case("ip_quadratic_resetfreq7_n300_c3", "ip", problem="quadratic", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80, "opt.hessian_reset_freq": 7}))
case("ip_quadratic_relfunc_n300_c3", "ip", problem="quadratic", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80, "opt.rel_func_tol": 1e-6}))
case("ip_quadratic_sigma_n300_c3", "ip", problem="quadratic", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80, "opt.qn_sigma": 0.5}))
case("ip_convex_sigma_sr1_n300_c3", "ip", problem="convex", n=300, c=3, dump_vecs_every=5,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.qn_type": "sr1", "opt.max_major_iters": 25,
                          "opt.qn_sigma": 2.0}))
case("ip_quadratic_nostart_n300_c3", "ip", problem="quadratic", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80,
                          "opt.starting_point_strategy": "no_start_strategy"}))
case("ip_quadratic_slp_noqn_n200_c2", "ip", problem="quadratic", n=200, c=2, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_type": "none", "opt.sequential_linear_method": 1, "opt.max_major_iters": 40}))
case("ip_quadratic_absstep_n300_c3", "ip", problem="quadratic", n=300, c=3, dump_vecs_every=10,
     **dict(ip_common, **{"opt.qn_subspace_size": 6, "opt.max_major_iters": 80, "opt.abs_step_tol": 1e-4}))
# --- the metric's configurations at n = 1e5 on FOUR MPI ranks (state of rank 0's shard, every 25th
# element, every 5th iteration): config-2 shape, config-3 shape with the convergent L-BFGS variant, and the L-SR1
# variant over a short window
case("ip_quadratic_n100000_c8_bfgs20_r4", "ip", ranks=4, problem="quadratic", n=100000, c=8, dump_vecs_every=5,
     vec_stride=25, **dict(ip_common, **{"opt.qn_subspace_size": 20, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}))
case("ip_convex_n100000_c32_bfgs10_r4", "ip", ranks=4, problem="convex", n=100000, c=32, dump_vecs_every=5,
     vec_stride=25, **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}))
case("ip_convex_n100000_c32_sr1_r4", "ip", ranks=4, problem="convex", n=100000, c=32, dump_vecs_every=5,
     vec_stride=25, **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 20}))
case("ip_convex_n100000_c32_bfgs10_r1", "ip", ranks=1, problem="convex", n=100000, c=32, dump_vecs_every=5,
     vec_stride=25, **dict(ip_common, **{"opt.qn_subspace_size": 10, "opt.qn_type": "bfgs", "opt.max_major_iters": 60}))


# --- state-injected single-step known-answer cases: the reference's private-method dumps at one
# iteration with COMPLETE state (x, zl, zu, dense blocks, mu, the limited-memory pairs S / Y and the small matrices
# B, L, D behind them), the Schur complements AS ASSEMBLED (Gmat, Ce before dgetrf: the LAPACK tracing shim of
# ref_driver.cpp), the first step (computeKKTStep) and the step after one refinement (:4985-4991).  Shapes: the
# metric's (convex, c = 32, L-SR1(10): a 43-column Gram) at n = 2000 and at n = 100 003 (odd; compare-only vectors
# stored as every 37th entry, problem data left out: a pure function of the case), config 2's (c = 8, L-BFGS(20)),
# and one with sparse weighting constraints (config 4's form).
kat_opts = dict(ip_common)
case("kat_convex_n2000_c32_sr1", "ip", problem="convex", n=2000, c=32, dump_vecs_every=0, kat_iter=12,
     **dict(kat_opts, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 14}))
case("kat_convex_n100003_c32_sr1", "ip", problem="convex", n=100003, c=32, dump_vecs_every=0, kat_iter=12,
     kat_light=1, kat_out_stride=37,
     **dict(kat_opts, **{"opt.qn_subspace_size": 10, "opt.qn_type": "sr1", "opt.max_major_iters": 13}))
case("kat_quadratic_n2000_c8_bfgs20", "ip", problem="quadratic", n=2000, c=8, dump_vecs_every=0, kat_iter=24,
     **dict(kat_opts, **{"opt.qn_subspace_size": 20, "opt.qn_type": "bfgs", "opt.max_major_iters": 26}))
case("kat_ipw_convex_n400_c4_w80", "ip", problem="convex", n=400, c=4, nwcon=80, nw=5, nwstart=0, nwskip=0,
     dump_vecs_every=0, kat_iter=9,
     **dict(kat_opts, **{"opt.qn_subspace_size": 8, "opt.qn_type": "bfgs", "opt.max_major_iters": 11}))

# --- the predictor-corrector step from the reference's own state (round 6): affine step + one refinement, probe to
# the boundary, complementarity there, the Mehrotra rule, corrector residual (addMehrotraCorrectorResidual) and ONE
# solve -- src/ParOptInteriorPoint.cpp:4956-5045 through the private methods (kat_mpc=1).  Shapes: config 5's steering
# solve (sequential linear method, panel = the c constraint gradients), a quasi-Newton panel of 14 columns (c = 8,
# L-BFGS(3): the widest the one-pass corrector kernels take), and an odd n = 30 011 with strided compare-only vectors.
mpc_opts = dict(kat_opts, **{"opt.barrier_strategy": "mehrotra_predictor_corrector"})
case("kat_mpc_convex_n2000_c4_seqlin", "ip", problem="convex", n=2000, c=4, dump_vecs_every=0, kat_iter=14, kat_mpc=1,
     kat_use_qn=0,
     **dict(mpc_opts, **{"opt.qn_subspace_size": 3, "opt.qn_type": "bfgs", "opt.sequential_linear_method": 1,
                         "opt.max_major_iters": 16}))
case("kat_mpc_quadratic_n2000_c8_bfgs3", "ip", problem="quadratic", n=2000, c=8, dump_vecs_every=0, kat_iter=7, kat_mpc=1,
     **dict(mpc_opts, **{"opt.qn_subspace_size": 3, "opt.qn_type": "bfgs", "opt.max_major_iters": 9}))
case("kat_mpc_quadratic_n30011_c3_bfgs4", "ip", problem="quadratic", n=30011, c=3, dump_vecs_every=0, kat_iter=8,
     kat_mpc=1, kat_light=1, kat_out_stride=11,
     **dict(mpc_opts, **{"opt.qn_subspace_size": 4, "opt.qn_type": "bfgs", "opt.max_major_iters": 10}))


def parse_tr_table(text):
    """Rows of the trust-region iteration table (paropt.tr): 13 numeric columns without the wall time, + info."""
    rows, infos = [], []
    for ln in text.splitlines():
        parts = ln.split()
        if len(parts) >= 14 and parts[0].isdigit():
            rows.append([float(v) for v in parts[:13]])
            infos.append(" ".join(parts[14:]))
    return np.array(rows), infos


def example_golden():
    """The reference's own example program (examples/rosenbrock/rosenbrock.cpp, built UNCHANGED against the
    reference library by oracle/Makefile) run here; its trust-region table is the golden for the same source file
    built against the product (oracle/_ref/rosenbrock_example_amd, tests/test_cpp_facade.py)."""
    exe = os.path.join(HERE, "_ref", "rosenbrock_example")
    env = dict(os.environ, MKL_NUM_THREADS="1", PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([exe], env=env, cwd=td, check=True, capture_output=True)
        tr_text = open(os.path.join(td, "paropt.tr")).read()
    rows, infos = parse_tr_table(tr_text)
    path = os.path.join(GOLDEN, "example_rosenbrock.npz")
    np.savez_compressed(path, table=rows, info=np.array(infos),
                        case_json=np.array(json.dumps(dict(mode="example", ranks=1, args={}))))
    print("%-40s %8d bytes  %d table rows" % ("example_rosenbrock", os.path.getsize(path), len(rows)))
    return dict(mode="example", ranks=1, args={}, bytes=os.path.getsize(path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", HERE])
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {}
    if not a.only or a.only in "example_rosenbrock":
        manifest["example_rosenbrock"] = example_golden()
    for name, (mode, ranks, args) in sorted(CASES.items()):
        if a.only and a.only not in name:
            continue
        with tempfile.TemporaryDirectory() as td:
            rec = os.path.join(td, "out.rec")
            dargs = dict(args)
            dargs["out"] = rec
            if mode == "ip":
                dargs["text"] = os.path.join(td, "paropt.out")
            if mode == "tr":
                dargs["text"] = os.path.join(td, "paropt.tr")
            if mode == "mma":
                dargs["text"] = os.path.join(td, "paropt.mma")
            if dargs.get("checkpoint"):
                dargs["checkpoint"] = os.path.join(td, "checkpoint.bin")
            run_driver(mode, dargs, ranks)
            d = read_rec(rec)
            if dargs.get("checkpoint"):
                # the reference's binary solution file (src/ParOptInteriorPoint.cpp:883-972), verbatim
                d["checkpoint_bytes"] = np.fromfile(dargs["checkpoint"], dtype=np.uint8)
            if mode == "ip":
                with open(dargs["text"]) as f:
                    lines = [ln.rstrip("\n") for ln in f]
                # check_flag of initAndCheckDesignAndBounds (:4290-4344) is a local of the reference: recovered
                # from the three warnings it prints (the checks run in the constructor and again in optimize())
                flag = 0
                for ln in lines:
                    if "Variable bounds are inconsistent" in ln:
                        flag |= 1
                    if "too close to lower bound" in ln:
                        flag |= 2
                    if "too close to upper bound" in ln:
                        flag |= 4
                d["check_flag"] = np.array([flag], dtype=np.int32)
                # keep only the iteration table (drop the options echo)
                start = next((i for i, ln in enumerate(lines) if ln.startswith("iter ")), 0)
                d["paropt_out"] = np.array("\n".join(lines[start:]))
            if mode == "tr":
                with open(dargs["text"]) as f:
                    lines = [ln.rstrip("\n") for ln in f]
                start = next((i for i, ln in enumerate(lines) if ln.strip().startswith("iter ")), 0)
                d["paropt_tr"] = np.array("\n".join(lines[start:]))
            if mode == "mma":
                with open(dargs["text"]) as f:
                    lines = [ln.rstrip("\n") for ln in f]
                start = next((i for i, ln in enumerate(lines) if ln.strip().startswith("MMA ")), 0)
                d["paropt_mma"] = np.array("\n".join(lines[start:]))
        d["case_json"] = np.array(json.dumps(dict(mode=mode, ranks=ranks, args=args)))
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, **d)
        manifest[name] = dict(mode=mode, ranks=ranks, args=args, bytes=os.path.getsize(path))
        print("%-40s %8d bytes  %d records" % (name, manifest[name]["bytes"], len(d)))
    mpath = os.path.join(GOLDEN, "MANIFEST.json")
    if a.only and os.path.exists(mpath):  # partial regeneration: merge into the existing manifest
        with open(mpath) as f:
            old = json.load(f)
        old.update(manifest)
        manifest = old
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
