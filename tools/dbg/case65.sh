python tools/dbg/case65.py 2>&1
for e in PAROPT_AMD_NO_RECOMPUTE_DT PAROPT_AMD_NO_FUSED_DOTS PAROPT_AMD_NO_FUSED_TDOTS PAROPT_AMD_NO_BATCH PAROPT_AMD_EXPLICIT_DOTS; do echo $e; env $e=1 python tools/dbg/case65.py 2>&1 | grep "^all"; done
