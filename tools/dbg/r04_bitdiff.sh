# 17-digit trajectories of a dozen sweep draws with the working library and with paropt_amd/libbase.so: any difference?
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PAROPT_SWEEP_CASES=400 PAROPT_SWEEP_SEED=909
for i in 0 1 2 3 5 8 13 21 34 55 89 144 233 377; do
  python3 tools/dbg/sweep_case_detail.py $i 2>/dev/null | grep "device" > gpurun_out/bd_new_$i.txt
  PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so python3 tools/dbg/sweep_case_detail.py $i 2>/dev/null | grep "device" > gpurun_out/bd_base_$i.txt
  if cmp -s gpurun_out/bd_new_$i.txt gpurun_out/bd_base_$i.txt; then echo "draw $i: identical ($(wc -l < gpurun_out/bd_new_$i.txt) iterations)"; else echo "draw $i: DIFFERS"; diff gpurun_out/bd_new_$i.txt gpurun_out/bd_base_$i.txt | head -4; fi
  rm -f gpurun_out/bd_new_$i.txt gpurun_out/bd_base_$i.txt
done
