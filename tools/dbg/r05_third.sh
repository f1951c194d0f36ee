#!/bin/bash
# round 5, third GPU call: tests of the exported config-4 route again, rocprof rows of the vector API, forced-RCCL
# per-rank sizes, config 3 with L-BFGS(20) (line of record + convergent comparison), one full-size CPU point
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_user_problem.py tests/test_gpu_csr.py tests/test_gpu_compat.py tests/test_gpu_ip.py tests/test_gpu_kat.py tests/test_gpu_multirank.py -q --no-header 2>&1 | tail -40 > gpurun_out/r05_tests3.log
tail -6 gpurun_out/r05_tests3.log
# --- vector API: kernel stats + HBM bytes (separate passes) ---
VEC="tools/microbench.py --vec-api --n 50000000 --reps 5"
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_vec_trace -o vec --output-format csv -- python3 $VEC > gpurun_out/r05_vec_trace.out 2> gpurun_out/r05_vec_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_r05vec_fetch -o fetch --output-format csv -- python3 $VEC > /dev/null 2> gpurun_out/pmc_r05vec_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_r05vec_write -o write --output-format csv -- python3 $VEC > /dev/null 2> gpurun_out/pmc_r05vec_write.err
python3 tools/pmc_summary.py gpurun_out/pmc_r05vec.json gpurun_out/pmc_r05vec_fetch gpurun_out/pmc_r05vec_write 2> gpurun_out/pmc_r05vec.err
find gpurun_out/pmc_r05vec_* gpurun_out/r05_vec_trace -name '*.csv' -size +2M -delete 2>/dev/null
ls gpurun_out/r05_vec_trace/*/ 2>/dev/null | head
# --- per-rank sizes of config 3, self communicator vs the forced single-rank RCCL communicator ---
for n in 50000000 25000000 12500000 6250000; do
  python3 bench.py --nglobal $n --steps 20 --warmup 12 --repeats 3 --boundary builtin --skip-extension-variant --no-cpu-baseline > gpurun_out/r05_rank_self_$n.json 2> gpurun_out/r05_rank_self_$n.err
  PAROPT_AMD_FORCE_RCCL=1 python3 bench.py --nglobal $n --steps 20 --warmup 12 --repeats 3 --boundary builtin --skip-extension-variant --no-cpu-baseline > gpurun_out/r05_rank_rccl_$n.json 2> gpurun_out/r05_rank_rccl_$n.err
  python3 - <<PY
import json
for k in ("self", "rccl"):
    try:
        r = json.load(open("gpurun_out/r05_rank_%s_$n.json" % k))
        print($n, k, "%.3f ms/it" % r["ms_per_step"], r["config"]["collective"], r.get("sync_counters"), (r.get("collective_us") or {}).get("allreduce_2628_doubles"))
    except Exception as e:
        print($n, k, "failed", e)
PY
done
# --- config 3 with L-BFGS(20): the fixed-K line of record (wide panel: 73 columns) and the convergent comparison ---
python3 bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 --boundary builtin --cpu-budget 60 > gpurun_out/r05_bench_c3_lbfgs20.json 2> gpurun_out/r05_bench_c3_lbfgs20.err
tail -2 gpurun_out/r05_bench_c3_lbfgs20.err
python3 tools/bench_convergent.py --n 50000000 --qn-size 20 --tol 1e-6 --cpu-n 2500000 --cpu-budget 900 > gpurun_out/r05_convergent_c3_lbfgs20.json 2> gpurun_out/r05_convergent_c3_lbfgs20.err
tail -2 gpurun_out/r05_convergent_c3_lbfgs20.err; head -c 1500 gpurun_out/r05_convergent_c3_lbfgs20.json
# --- config 3 as the driver runs it, plus ONE full-size run of the CPU reference ---
python3 bench.py --cpu-budget 240 --cpu-full-size > gpurun_out/r05_bench_c3_fullcpu.json 2> gpurun_out/r05_bench_c3_fullcpu.err
tail -3 gpurun_out/r05_bench_c3_fullcpu.err
