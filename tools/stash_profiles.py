#!/usr/bin/env python3
"""Copy the summaries tools/collect_profiles.sh / collect_pmc.sh / microbench.py left under gpurun_out/ into
profiles/ under this round's names (gpurun_out/ is scratch, profiles/ is tracked).  usage: stash_profiles.py r02"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("  %s -> profiles/%s" % (src, dst))
    else:
        print("  (missing: %s)" % src)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    cp("bench_%s_final.json" % tag, "%s_bench_final.json" % tag)
    cp("bench_%s_2rank_shared.json" % tag, "%s_bench_2rank_shared_gpu.json" % tag)
    for c in ("c2", "c3", "c4", "c5", "csr"):
        cp("prof_%s/%s_kernel_stats.csv" % (c, c), "%s_rocprofv3_kernel_stats_%s.csv" % (tag, c))
        cp("bench_under_rocprof_%s.json" % c, "%s_bench_under_rocprof_%s.json" % (tag, c))
    cp("pmc_%s.json" % tag, "%s_pmc_counters.json" % tag)
    cp("%s_microbench.jsonl" % tag, "%s_microbench.jsonl" % tag)
    # HBM traffic of the kernels the bench line quotes (bench.py reads the mdot<32> entry)
    src = os.path.join(G, "pmc_%s.json" % tag)
    if os.path.exists(src):
        d = json.load(open(src))
        raw = {}
        for k, v in d.items():
            if any(t in k for t in ("mdot_kernel<32>", "solve2_dots_kernel<11", "solve2_kernel<1, 0>", "solve2r_kernel",
                                    "kkt_res_update_kernel", "wgram_pc_kernel<11, 3>", "wgram_kernel<11, 3")):
                e = {n: v[n] for n in ("hbm_read_bytes_corrected", "hbm_write_bytes") if n in v}
                if "FETCH_SIZE" in v:
                    e["FETCH_SIZE_KB_mean"] = v["FETCH_SIZE"]["mean_per_dispatch"]
                    e["calls"] = v["FETCH_SIZE"]["dispatches"]
                if "WRITE_SIZE" in v:
                    e["WRITE_SIZE_KB_mean"] = v["WRITE_SIZE"]["mean_per_dispatch"]
                raw[k] = e
        out = {"command": "bash tools/collect_pmc.sh %s  (rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 6 "
                          "--warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin; one pass per group: FETCH_SIZE | "
                          "WRITE_SIZE | SQ_* | GRBM/SQ_WAVES)" % tag,
               "n": 50000000, "mdot32_algorithmic_bytes": 13200000000, "raw": raw,
               "units": "mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB; hbm_read_bytes_corrected = 2 * 1024 * FETCH_SIZE "
                        "(gfx950: FETCH_SIZE counts 64 B per 128 B request for wide streaming reads, MI355X_MICROARCH.md HBM)"}
        json.dump(out, open(os.path.join(P, "%s_pmc_hbm_traffic.json" % tag), "w"), indent=1, sort_keys=True)
        print("  wrote profiles/%s_pmc_hbm_traffic.json (%d kernels)" % (tag, len(raw)))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_wgram.py"), src,
                               os.path.join(P, "%s_pmc_sq_wgram.json" % tag)])


if __name__ == "__main__":
    main()
