"""
State-injected single-step known-answer tests (through the C ABI).

The REFERENCE's state at one iteration -- x, zl, zu, the dense blocks, mu, the limited-memory pairs S / Y with the
small matrices B, L, D behind them, dumped from its private members by oracle/ref_driver.cpp -- is loaded into the
device solver (po_ip_debug_set_state, po_qn_debug_load) and every piece of the KKT step is compared with what the
reference's private methods produced FROM THAT SAME STATE (src/ParOptInteriorPoint.cpp):

    Dinv                    setUpKKTDiagSystem :1864-1910              <= 4 ulp
    residual (rx, dense)    computeKKTRes :1337-1446                   <= 1e-13 of the block's largest term
    G, Ce  AS ASSEMBLED     :1932-1970, :2634-2667 (before dgetrf)     <= 1e-11 of the matrix' largest entry
    gpiv, cpiv              LAPACK pivot rows                          exact
    compact matrix M        computeMatUpdate (QuasiNewton.cpp:339-377) exact (same arithmetic on the same B, L, D)
    first step              computeKKTStep :2700-2737                  <= 1e-9 of the block's largest entry
    step after refinement   :4985-4991                                 <= 1e-9   (the fused kernel sequence of optimize())
    fraction to boundary    computeMaxStep :2942-3103                  <= 1e-7 relative (one entry's ratio)

No trajectory is involved: a regression of 1e-9 in the Gram kernel or in a solve pass fails here although every
trajectory golden (tests/test_gpu_ip.py) would still pass -- test_kat_detects_a_perturbed_gram proves it with a
deliberately injected relative perturbation of 1e-9 in one Gram entry.

Shapes: the metric's (convex, c = 32, L-SR1(10): the 43-column Gram) at n = 2000 and n = 100 003, config 2's (c = 8,
L-BFGS(20)), one with sparse weighting constraints (config 4's form), and the four small cases of rounds 1-3.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, ip_options_from_case, load_golden

pytestmark = pytest.mark.gpu

KAT_CASES = ["kat_convex_n2000_c32_sr1", "kat_convex_n100003_c32_sr1", "kat_quadratic_n2000_c8_bfgs20",
             "kat_ipw_convex_n400_c4_w80", "ip_quadratic_n257_c3_bfgs", "ip_quadratic_n1000_c8_bfgs20",
             "ip_convex_n300_c5_bfgs", "ip_convex_n300_c5_sr1"]

TOL_DINV_ULP = 4
TOL_RES = 1e-13
TOL_MAT = 1e-11
TOL_STEP = 1e-9
TOL_MAXSTEP = 1e-7

MEASURED = {}  # achieved error / tolerance per case and piece, written to gpurun_out/ for the record


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()
    out = os.path.join(ROOT, "gpurun_out")
    if MEASURED and os.path.isdir(out):
        with open(os.path.join(out, "kat_measured.json"), "w") as f:
            json.dump(MEASURED, f, indent=1, sort_keys=True)


def colmajor(flat):
    m = int(round(np.sqrt(flat.size)))
    return np.asarray(flat).reshape(m, m).T


def inject(ctx, name):
    """Device solver holding the reference's state of golden `name`; returns (ip, golden, stride of the stored
    compare-only vectors)."""
    import paropt_amd as pa

    g, case = load_golden(name)
    a = case["args"]
    prob = pa.SeparableProblem(ctx, a["problem"], a["n"], a.get("c", 2), a.get("seed", 0), a.get("eig_min", 1.0),
                               a.get("eig_max", 100.0))
    if a.get("nwcon", 0) > 0:
        prob.setWeighting(a["nwcon"], a["nw"], a.get("nwstart", 0), a.get("nwskip", 0), a.get("nwineq", a["nwcon"]))
    opts = ip_options_from_case(case)
    opts["write_output_frequency"] = 0
    opts["max_major_iters"] = 1  # initialises bounds, multipliers and the quasi-Newton object; the state is replaced
    ip = pa.InteriorPoint(prob, opts)
    ip.optimize()
    x, _, zl, zu = ip.getOptimizedPoint()
    x.from_numpy(g["kat/x"])
    zl.from_numpy(g["kat/zl"])
    zu.from_numpy(g["kat/zu"])
    wv = ip.getOptimizedSparse()
    if wv is not None:
        for key, v in zip(("zw", "sw", "tw", "zsw", "ztw"), wv):
            v.from_numpy(g["kat/" + key])
    msub, msub_max = (int(v) for v in g["kat/qn_sizes"])
    qn = ip.getQuasiNewton()
    qn.debugLoad(g["kat/qn_b0"][0], g["kat/qn_B"], g["kat/qn_L"], g["kat/qn_D"],
                 [g["kat/S%d" % j] for j in range(msub)], [g["kat/Y%d" % j] for j in range(msub)])
    assert len(g["kat/qn_D"]) == msub_max
    ip.debugSetState(g["kat/z"], g["kat/s"], g["kat/t"], g["kat/zs"], g["kat/zt"], g["kat/mu"][0])
    stride = int(g["kat/out_stride"][0]) if "kat/out_stride" in g else 1
    return ip, g, stride


def record(name, piece, err, tol):
    MEASURED.setdefault(name, {})[piece] = {"error": float(err), "tolerance": float(tol)}


def check_pieces(name, g, d, stride, first_step):
    """Compare the dump `d` of po_ip_debug_kkt with the reference's private-method records of golden g."""
    c, k = d["c"], d["k"]
    # Dinv: one division and a few additions per entry -- ulp level
    ref = g["kat/Dinv"]
    mine = d["Dinv"][::stride]
    ulp = np.abs(mine - ref) / np.spacing(np.abs(ref))
    record(name, "Dinv_ulp", ulp.max(), TOL_DINV_ULP)
    assert ulp.max() <= TOL_DINV_ULP, "Dinv: %g ulp" % ulp.max()
    # residual: rx = zl - zu - g + A^T z is a sum of c + 3 terms, compared against the size of its largest term
    ref = g["kat/res_x"]
    mine = d["res_x"][::stride]
    scale = max(np.abs(ref).max(), np.abs(g["kat/zl"]).max(), np.abs(g["kat/zu"]).max())
    if "kat/g" in g:
        scale = max(scale, np.abs(g["kat/g"]).max())
    err = np.abs(mine - ref).max() / scale
    record(name, "res_x", err, TOL_RES)
    assert err <= TOL_RES, "res_x: %g" % err
    # res.z = -(c(x) - s + t): c(x) is the problem's own reduction over n terms a_ji x_i (+ offset), evaluated by the
    # device problem in another summation order -- compared against the sum of the magnitudes of those terms
    xabs = np.abs(g["kat/x"])
    if "kat/Ac0" in g:
        cterms = max(float(np.abs(g["kat/Ac%d" % j]) @ xabs) for j in range(c))
    else:  # large-n record without the Jacobian: entries of the separable problems' Jacobians are in [0, 1)
        cterms = 0.5 * float(xabs.sum())
    for key in ("z", "s", "t", "zs", "zt"):
        ref = g["kat/res_" + key]
        sc = max(1.0, np.abs(ref).max(), max(np.abs(g["kat/c"]).max(), cterms) if key == "z" else 0.0)
        err = np.abs(d["res_" + key] - ref).max() / sc
        record(name, "res_" + key, err, TOL_RES)
        assert err <= TOL_RES, "res_%s: %g" % (key, err)
    err = np.abs(d["res_norms"] - g["kat/res_norms"]).max() / max(1.0, np.abs(g["kat/res_norms"]).max())
    record(name, "res_norms", err, TOL_RES)
    assert err <= TOL_RES, "residual norms: %g" % err
    # the Schur complements as assembled
    G = colmajor(g["kat/Gmat"])
    err = np.abs(d["G"] - G).max() / np.abs(G).max()
    record(name, "G" + ("" if first_step else "_fused"), err, TOL_MAT)
    assert err <= TOL_MAT, "G: %g of its largest entry" % err
    # ... and entry by entry in the natural scaling of a Gram matrix, |dG_ij| / sqrt(G_ii G_jj): small rows count too
    dg = np.sqrt(np.abs(np.diag(G)))
    err = (np.abs(d["G"] - G) / np.outer(dg, dg)).max()
    record(name, "G_scaled" + ("" if first_step else "_fused"), err, TOL_MAT)
    assert err <= TOL_MAT, "G: %g in the diagonal scaling" % err
    np.testing.assert_array_equal(d["gpiv"], g["kat/gpiv"], err_msg="gpiv")
    if k > 0:
        Ce = colmajor(g["kat/Ce"])
        assert Ce.shape == (k, k)
        err = np.abs(d["Ce"] - Ce).max() / np.abs(Ce).max()
        record(name, "Ce" + ("" if first_step else "_fused"), err, TOL_MAT)
        assert err <= TOL_MAT, "Ce: %g of its largest entry" % err
        np.testing.assert_array_equal(d["cpiv"], g["kat/cpiv"], err_msg="cpiv")
    # the step
    pre = "kat/step_" if first_step else "kat/rstep_"
    tag = "step_" if first_step else "rstep_"
    keys = ["x", "zl", "zu", "z", "s", "t", "zs", "zt"]
    if "step_zw" in d:
        keys += ["zw", "sw", "tw", "zsw", "ztw"]
    for key in keys:
        ref = g[pre + key]
        mine = d["step_" + key]
        if key in ("x", "zl", "zu"):
            mine = mine[::stride]
        err = np.abs(mine - ref).max() / max(np.abs(ref).max(), 1e-300)
        record(name, tag + key, err, TOL_STEP)
        assert err <= TOL_STEP, "%s%s: %g of its largest entry" % (tag, key, err)
    # fraction to the boundary (tau = 0.95): vector part from the solve pass, dense blocks on the host
    smin = np.minimum(1.0, d["step_mins"])
    for blk, var, i in (("s", "s", 0), ("t", "t", 0), ("zs", "zs", 1), ("zt", "zt", 1)):
        p, v = d["step_" + blk], g["kat/" + var]
        neg = p < 0.0
        if neg.any():
            smin[i] = min(smin[i], (-0.95 * v[neg] / p[neg]).min())
    if "step_zw" in d:
        for blk, var, i in (("sw", "sw", 0), ("tw", "tw", 0), ("zsw", "zsw", 1), ("ztw", "ztw", 1)):
            p, v = d["step_" + blk], g["kat/" + var]
            neg = p < 0.0
            if neg.any():
                smin[i] = min(smin[i], (-0.95 * v[neg] / p[neg]).min())
    # (a ratio -tau v_i / p_i at ONE entry: its relative error is that entry's, which exceeds the step's error relative
    # to its largest entry by max|p| / |p_i| -- hence the wider tolerance)
    ref = g["kat/max_step_tau095" if first_step else "kat/rmax_step_tau095"]
    err = (np.abs(smin - ref) / np.abs(ref)).max()
    record(name, tag + "max_step", err, TOL_MAXSTEP)
    assert err <= TOL_MAXSTEP, "max step: %s vs %s" % (smin, ref)


def compact_matrix(ip):
    """M of the loaded quasi-Newton state without forming Z (po_qn_get_compact with Z = NULL)."""
    import ctypes as C

    import paropt_amd.lib as L

    qn = ip.getQuasiNewton()
    k, b0 = C.c_int(), C.c_double()
    d0, M = L.c_double_p(), L.c_double_p()
    rc = L.lib.po_qn_get_compact(qn._h, C.byref(k), C.byref(b0), C.byref(d0), C.byref(M), None)
    assert rc == 0
    n = k.value
    return b0.value, np.array([d0[i] for i in range(n)]), np.array([M[i] for i in range(n * n)])


@pytest.mark.parametrize("name", KAT_CASES)
def test_kat_first_step_from_reference_state(ctx, name):
    """computeKKTRes + setUpKKTDiagSystem + setUpKKTSystem + computeKKTStep from the reference's own state."""
    ip, g, stride = inject(ctx, name)
    b0, d0, M = compact_matrix(ip)
    assert b0 == g["kat/qn_b0"][0]
    np.testing.assert_array_equal(d0, g["kat/qn_d0"])
    np.testing.assert_array_equal(M, g["kat/qn_M"])  # same arithmetic on the same B, L, D: the same bits
    d = ip.debugKKT(g["kat/mu"][0], 0)
    check_pieces(name, g, d, stride, first_step=True)
    comp = ip.getComplementarity()
    err = abs(comp - g["kat/comp"][0]) / abs(g["kat/comp"][0])
    record(name, "comp", err, TOL_RES)
    assert err <= 1e-13, "complementarity: %g" % err


@pytest.mark.parametrize("name", KAT_CASES)
def test_kat_refined_step_through_the_fused_sequence(ctx, name):
    """The kernel sequence optimize() runs in a plain quasi-Newton iteration (dinv_d1, fused Gram over unformed
    L-SR1 columns, first solve pass with the refinement's products, refinement pass) against the reference's
    computeKKTStep + one refinement (:4971-4991), from the reference's own state."""
    ip, g, stride = inject(ctx, name)
    d = ip.debugKKT(g["kat/mu"][0], 1)
    check_pieces(name, g, d, stride, first_step=False)


@pytest.mark.parametrize("name", ["kat_convex_n2000_c32_sr1", "kat_quadratic_n2000_c8_bfgs20"])
@pytest.mark.parametrize("mode", [0, 1])
def test_kat_detects_a_perturbed_gram(ctx, name, mode):
    """A relative perturbation of 1e-9 in ONE entry of the Gram kernel's output (debug switch SW_PERTURB_W) must fail
    the comparison -- the sensitivity the flat trajectory tolerances never had."""
    import paropt_amd.lib as L

    SW_PERTURB_W = 11
    ip, g, stride = inject(ctx, name)
    L.lib.po_debug_set_switch(SW_PERTURB_W, 1)
    try:
        d = ip.debugKKT(g["kat/mu"][0], mode)
    finally:
        L.lib.po_debug_set_switch(SW_PERTURB_W, -1)
    with pytest.raises(AssertionError):
        check_pieces(name + "_perturbed", g, d, stride, first_step=(mode == 0))
    MEASURED.pop(name + "_perturbed", None)
    # ... and the unperturbed run of the same solver object passes
    d = ip.debugKKT(g["kat/mu"][0], mode)
    check_pieces(name, g, d, stride, first_step=(mode == 0))


# ---- the predictor-corrector step (round 6) -------------------------------------------------------------------------
# tests/golden/kat_mpc_*.npz: from the reference's own state at one iteration, its private methods in the order of
# optimize() :4956-5045 -- affine residual (mu = 0), computeKKTStep + one refinement, computeMaxStep(tau = 1),
# computeCompStep there, the Mehrotra rule, computeKKTRes at the new barrier parameter + addMehrotraCorrectorResidual
# (:1729-1789), ONE computeKKTStep.  The device runs mehrotraStep() of optimize() from the injected state
# (po_ip_debug_kkt mode 2): with the round's kernels (corrector right-hand side in one pass, corrector solve with the
# merit sums, affine complementarity from the polynomial) and with the plain sequence they replace.
MPC_CASES = ["kat_mpc_convex_n2000_c4_seqlin", "kat_mpc_quadratic_n2000_c8_bfgs3", "kat_mpc_quadratic_n30011_c3_bfgs4"]


@pytest.mark.parametrize("name", MPC_CASES)
def test_kat_predictor_corrector_step_from_reference_state(ctx, name):
    launches = {}
    for fused in (1, 0):
        launches[fused] = _mpc_kat(ctx, name, fused)
    assert launches[1] <= launches[0] - 4, launches  # (comp_step, corrector, d1, mdot, comp_merit against two passes)


def _mpc_kat(ctx, name, fused):
    import paropt_amd.lib as L

    SW_MPC_FUSE, SW_MPC_POLY = 14, 15
    ip, g, stride = inject(ctx, name)
    c = len(g["kat/z"])
    L.lib.po_debug_set_switch(SW_MPC_FUSE, fused)
    L.lib.po_debug_set_switch(SW_MPC_POLY, fused)
    try:
        n0 = ctx.counters()[1]
        d = ip.debugKKT(g["kat/mu"][0], 2)
        nlaunch = ctx.counters()[1] - n0
    finally:
        L.lib.po_debug_set_switch(SW_MPC_FUSE, -1)
        L.lib.po_debug_set_switch(SW_MPC_POLY, -1)
    tag = name + ("" if fused else "_plain")
    # the diagonal system (no quasi-Newton diagonal under the sequential linear method) and its Schur complement
    ulp = np.abs(d["Dinv"][::stride] - g["kat/Dinv"]) / np.spacing(np.abs(g["kat/Dinv"]))
    record(tag, "Dinv_ulp", ulp.max(), TOL_DINV_ULP)
    assert ulp.max() <= TOL_DINV_ULP
    G = colmajor(g["kat/Gmat"])
    err = np.abs(d["G"] - G).max() / np.abs(G).max()
    record(tag, "G", err, TOL_MAT)
    assert err <= TOL_MAT
    np.testing.assert_array_equal(d["gpiv"], g["kat/gpiv"])
    # the Mehrotra rule: mu = max(0.01, (comp_affine / comp)^3) comp -- three times the relative error of the affine
    # complementarity, which the fused form takes from the polynomial S00 + ax S10 + az S01 + ax az S11
    mu_new = ip.getBarrierParameter()
    err = abs(mu_new - g["kat/mpc_mu"][0]) / g["kat/mpc_mu"][0]
    record(tag, "mpc_mu", err, TOL_STEP)
    assert err <= TOL_STEP, "barrier parameter of the Mehrotra rule: %g vs %g" % (mu_new, g["kat/mpc_mu"][0])
    # the corrector step
    for key in ("x", "zl", "zu", "z", "s", "t", "zs", "zt"):
        ref = g["kat/mpc_step_" + key]
        mine = d["step_" + key]
        if key in ("x", "zl", "zu"):
            mine = mine[::stride]
        err = np.abs(mine - ref).max() / max(np.abs(ref).max(), 1e-300)
        record(tag, "mpc_step_" + key, err, TOL_STEP)
        assert err <= TOL_STEP, "corrector step %s: %g of its largest entry" % (key, err)
    # fraction to the boundary of the corrector step: the device took it with tau = max(min fraction, 1 - mu_new), the
    # record with 0.95 -- both are tau x (the smallest ratio), capped at 1
    tau_dev = max(0.95, 1.0 - mu_new)
    smin = np.minimum(1.0, d["step_mins"])
    for blk, var, i in (("s", "s", 0), ("t", "t", 0), ("zs", "zs", 1), ("zt", "zt", 1)):
        p, v = d["step_" + blk], g["kat/" + var]
        neg = p < 0.0
        if neg.any():
            smin[i] = min(smin[i], (-tau_dev * v[neg] / p[neg]).min())
    ref = g["kat/mpc_max_step_tau095"]
    for i in range(2):
        if smin[i] < 1.0 and ref[i] < 1.0:
            err = abs(smin[i] / tau_dev * 0.95 - ref[i]) / ref[i]
            record(tag, "mpc_max_step_%d" % i, err, TOL_MAXSTEP)
            assert err <= TOL_MAXSTEP, ("max step", i, smin, ref)
    assert c == d["c"]
    return nlaunch
