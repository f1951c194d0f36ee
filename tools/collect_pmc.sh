#!/bin/bash
# PMC passes over the bench run (config 3 by default), one rocprofv3 run per counter group, counters only
# with --kernel-trace (gpurun refuses --pmc together with the other trace domains).  Summaries are written
# by tools/pmc_summary.py to gpurun_out/pmc_<tag>.json: per kernel, the mean counter value per dispatch.
#   usage (repo root on the GPU box):  bash tools/collect_pmc.sh <tag> [bench args...]
set -u
tag=${1:-r02}
shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args=("$@")
if [ ${#args[@]} -eq 0 ]; then args=(--steps 6 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin); fi
pass() {  # name, counters...
  local name=$1
  shift
  rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmc_${tag}_$name -o $name --output-format csv -- \
      python3 bench.py "${args[@]}" > gpurun_out/pmc_${tag}_$name.out 2> gpurun_out/pmc_${tag}_$name.err
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
pass grbm GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}.json gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write \
    gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_grbm
# the per-dispatch CSVs are large: keep only the summary
find gpurun_out/pmc_${tag}_* -name '*.csv' -size +2M -delete 2>/dev/null
