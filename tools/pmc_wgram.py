#!/usr/bin/env python3
"""The weighted-Gram and solve-pass rows of a tools/pmc_summary.py file, with the derived ratios, next to round 1's
weighted-Gram kernel (profiles/r01_pmc_sq_wgram_mdot.json).
usage: pmc_wgram.py gpurun_out/pmc_r02.json profiles/r02_pmc_sq_wgram.json
Units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = kernel duration in shader cycles);
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles."""
import json
import os
import sys


def derived(c):
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out = {"kernel_cycles": g}
    if g > 0:
        out["mfma_busy_frac_of_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024.0)
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w > 0:
        out["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / w
        out["wait_inst_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / w
    a = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if a > 0:
        out["lds_bank_conflict_frac_of_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / a
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = json.load(open(src))
    res = {"source": src, "note": __doc__.split("usage")[0].strip(), "round2": {}, "round1": {}}
    for k, v in d.items():
        if not any(t in k for t in ("wgram", "solve2", "mdot_kernel<32>")):
            continue
        c = {n: x["mean_per_dispatch"] for n, x in v.items() if isinstance(x, dict)}
        e = {"dispatches": max(x["dispatches"] for x in v.values() if isinstance(x, dict)), "counters": c,
             "derived": derived(c)}
        for n in ("hbm_read_bytes_corrected", "hbm_write_bytes"):
            if n in v:
                e[n] = v[n]
        res["round2"][k] = e
    r1 = os.path.join(os.path.dirname(os.path.abspath(dst)), "r01_pmc_sq_wgram_mdot.json")
    if os.path.exists(r1):
        for k, c in json.load(open(r1))["counters"].items():
            res["round1"][k] = {"counters": c, "derived": derived(c)}
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    for rnd in ("round1", "round2"):
        for k, e in sorted(res[rnd].items()):
            if "wgram" in k:
                print(rnd, k, {a: round(b, 3) for a, b in e["derived"].items()})


if __name__ == "__main__":
    main()
