#!/usr/bin/env python3
"""Per-kernel HBM table of one configuration: PMC bytes per launch (tools/pmc_summary.py JSON: FETCH_SIZE with the gfx950
x2 correction of MI355X_MICROARCH.md, WRITE_SIZE) x average launch time (rocprofv3 --kernel-trace --stats CSV).
    python tools/pmc_table.py <pmc.json> <kernel_stats.csv> [iteration_ms] > profiles/rNN_pmc_cX.txt
A launch whose PMC traffic is below the 256 MB Infinity Cache is not an HBM stream (its operands were just written or
read by the launch before it): no fraction is stated for it -- a number above the HBM peak is not evidence."""
import csv
import json
import sys

PEAK = 8.0e12
MALL = 256.0e6


def main():
    pmc = json.load(open(sys.argv[1]))
    rows = list(csv.DictReader(open(sys.argv[2])))
    it_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# kernel | calls | avg ms | share of kernel time | PMC read GB + write GB per launch | TB/s | fraction of 8 TB/s")
    for r in rows:
        name = r["Name"].split("(")[0]
        key = next((k for k in pmc if k == name or k.startswith(name)), None)
        share = float(r["TotalDurationNs"]) / tot
        if share < 0.004:
            continue
        avg = float(r["AverageNs"]) * 1e-9
        if key is None or "hbm_read_bytes_corrected" not in pmc[key]:
            print("%-58s %6s  %8.3f ms  %5.1f %%   (no PMC row)" % (name[:58], r["Calls"], avg * 1e3, 100 * share))
            continue
        rd = pmc[key]["hbm_read_bytes_corrected"]
        wr = pmc[key].get("hbm_write_bytes", 0.0)
        if rd + wr < MALL:
            note = "n/a (%.0f MB per launch: resident in the 256 MB Infinity Cache, not an HBM stream)" % ((rd + wr) * 1e-6)
            print("%-58s %6s  %8.3f ms  %5.1f %%   read %.3f GB  write %.3f GB  -> %s" % (
                name[:58], r["Calls"], avg * 1e3, 100 * share, rd * 1e-9, wr * 1e-9, note))
            continue
        rate = (rd + wr) / avg
        print("%-58s %6s  %8.3f ms  %5.1f %%   read %.3f GB  write %.3f GB  -> %.2f TB/s = %.2f of 8 TB/s" % (
            name[:58], r["Calls"], avg * 1e3, 100 * share, rd * 1e-9, wr * 1e-9, rate * 1e-12, rate / PEAK))
    if it_ms:
        print("# iteration: %.3f ms" % it_ms)


if __name__ == "__main__":
    main()
