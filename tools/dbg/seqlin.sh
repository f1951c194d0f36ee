TAG=default python tools/dbg/seqlin.py | head -1
for e in PAROPT_AMD_NO_LEAN_STEP PAROPT_AMD_NO_FUSED_MERIT PAROPT_AMD_NO_RECOMPUTE_DT PAROPT_AMD_NO_RECOMPUTE PAROPT_AMD_NO_RECOMPUTE_RHS PAROPT_AMD_NO_FUSED_UPDATE PAROPT_AMD_NO_FUSED_DOTS PAROPT_AMD_NO_FUSED_TDOTS PAROPT_AMD_EXPLICIT_DOTS PAROPT_AMD_NO_BATCH; do
  env $e=1 TAG=$e python tools/dbg/seqlin.py | head -1
done
TAG=noqnupdate python tools/dbg/seqlin.py "{'use_quasi_newton_update': False}" | head -1
python tools/dbg/seqlin.py | tail -8
