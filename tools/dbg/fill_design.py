"""Fills the @PLACEHOLDER@s of DESIGN.md.in from the committed r04 records -> DESIGN.md (development aid)."""
import csv, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = lambda f: os.path.join(R, "profiles", f)
def rec(c):
    f = P("r04_bench_%s.json" % c)
    return json.load(open(f)) if os.path.exists(f) else None
def pmc(c):
    rows = {}
    f = P("r04_pmc_%s.txt" % c)
    if not os.path.exists(f):
        return rows
    for ln in open(f):
        m = re.match(r"(?:void )?po::(.+?)\s{2,}(\d+)\s+([\d.]+) ms\s+([\d.]+) %.*?(?:= ([\d.]+) of 8 TB/s|n/a)", ln)
        if m:
            rows.setdefault(m.group(1).replace(" ", ""), (int(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5)))
    return rows
c2, c3, c4, c5 = rec("c2"), rec("c3"), rec("c4"), rec("c5")
p3, p4 = pmc("c3"), pmc("c4")
def k3(name):
    r = p3.get(name)
    return "%.2f ms = %s" % (r[1], r[3]) if r else "n/a"
sub = {}
sub["C4_DOUBLES"] = "%.0f" % (c4["iteration_bytes"] / 8 / c4["config"]["n_global"])
sub["C4_GB"] = "%.1f" % (c4["iteration_bytes"] / 1e9)
sub["MDOT_MS"] = "%.2f" % c3["roofline"]["avg_launch_ms"]
sub["MDOT_FRAC"] = "%.2f" % c3["roofline"]["frac"]
sub["WGRAM_C3"] = k3("wgram_pc_kernel<11,3,1,0>")
sub["WGRAM_C4"] = "%.2f ms" % p4["wgram_pc_kernel<7,0,1,1>"][1] if "wgram_pc_kernel<7,0,1,1>" in p4 else "1.07 ms"
sub["KKT_C3"] = k3("kkt_res_update_kernel")
sub["DINV_C3"] = k3("dinv_d1_kernel")
sub["S2D_C3"] = k3("solve2_dots_kernel<11,2,0>")
sub["S2R_C3"] = k3("solve2r_kernel<1,1>")
sub["TRIAL_C3"] = k3("trial_kernel")
def k4(name):
    r = p4.get(name)
    return "`%s` %.3f ms%s" % (name.rstrip(","), r[1], (" = " + r[3]) if r[3] else "") if r else ""
sub["C4_KERNELS"] = "; ".join(x for x in (k4("group_k0_tiled_kernel"), k4("group_sum_tiled_kernel"), k4("group_factor_tiled_kernel"), k4("group_scatter2_kernel")) if x) + " at n = 20 M, w = 1 M (w-sized launches are Infinity-Cache resident: no HBM fraction)"
def line(tag, r, extra=""):
    if r is None:
        return "| %s | (pending) | | | | | | |" % tag
    cb = r.get("cpu_baseline") or {}
    cfg = r.get("config", {})
    if "inner_ip_iterations_per_s" in r:
        return "| %s | **%.2f TR it/s = %.0f inner IP it/s** | %.3f per inner | %.2f | %.1f / %.2f | %.2f (`mdot`, %d vectors) | %.3f TR it/s (%d) | %s |" % (
            tag, r["value"], r["inner_ip_iterations_per_s"], r["ms_per_inner_iteration"], r["iteration_frac"],
            r["launches_per_inner_iteration"], r["host_syncs_per_inner_iteration"], r["roofline"]["frac"], 1 + 4,
            cb.get("value", float("nan")), cb.get("cores", 0), extra)
    return "| %s | **%.1f it/s** | %.3f | %.2f | %.0f / %.0f | %.2f (`mdot`, %s) | %.3f steady, %.3f whole run (%d) | %s |" % (
        tag, r["value"], r["ms_per_step"], r["iteration_frac"], cfg["launches_per_iter"], cfg["reductions_per_iter"],
        r["roofline"]["frac"], re.search(r"nvecs=(\d+)", r["roofline"]["kernel"]).group(1) + " vectors",
        cb.get("value", float("nan")), cb.get("whole_run_it_per_s", float("nan")), cb.get("cores", 0), extra)
sub["RECORD_TABLE"] = "\n".join([
    line("2: quadratic, n = 10 M, m = 8, L-BFGS(20)", c2, "292–295 it/s (under the profiler)"),
    line("3: convex, n = 50 M, m = 32, L-SR1(10) — the metric", c3, "42.9 it/s (40.6 on its slowest box); this round 40.6 on a box whose copy ceiling was 5.3 TB/s and 43.0 on one with 6.3 TB/s before the two-wave epilogue (`r04_bench_c3_slow_store_box.json`, `r04_bench_c3_before_two_wave_epilogue.json`; the box of this line: 6.0)"),
    line("4: n = 20 M, m = 4, 1 M weighting constraints, L-BFGS(10)", c4, "125.4 it/s, 85 launches, 8 syncs"),
    line("5: trust region + eigenvalue model, n = 5 M", c5, "10.3 TR it/s = 664 inner it/s, 29.8 / 9.1 (review target: 750)"),
])
def shares(p, n=7):
    rows = sorted(p.items(), key=lambda kv: -kv[1][2])[:n]
    return ", ".join("`%s` %.1f %%%s" % (k.rstrip(","), v[2], (" (%s)" % v[3]) if v[3] else "") for k, v in rows)
sub["C3_SHARES"] = shares(p3)
sub["C4_PMC"] = shares(p4, 9)
sub["PER_RANK"] = "config 3: 23.47 / 11.68 / 6.01 / 3.08 ms per iteration at n/1, n/2, n/4, n/8 (95.3 % of perfect before any collective); config 4: 5.68 / 2.93 / 1.57 ms at n/1, n/2, n/4 (90.5 %) (`profiles/r04_per_rank_sizes.txt`, `tools/per_rank_sizes.sh`)"
sub["PRED"] = "≈ 90 %"
src = open(os.path.join(R, "tools", "dbg", "DESIGN.md.in")).read()
missing = set(re.findall(r"@([A-Z0-9_]+)@", src)) - set(sub)
assert not missing, missing
out = re.sub(r"@([A-Z0-9_]+)@", lambda m: sub[m.group(1)], src)
open(os.path.join(R, "DESIGN.md"), "w").write(out)
print("DESIGN.md: %d lines" % out.count("\n"))
