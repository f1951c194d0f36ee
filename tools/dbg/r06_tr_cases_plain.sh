# the six differing draws of the second trust-region campaign (seed 727) with the round-5 corrector sequence and with
# round 6's, against the oracle: is the difference this round's?
cd $GRAFT_REPO_ROOT
for sw in "0 0 0" "1 1 1"; do
  set -- $sw
  echo "== PAROPT_AMD_MPC_FUSE=$1 PAROPT_AMD_MPC_POLY=$2 PAROPT_AMD_SPEC_DT=$3"
  PAROPT_AMD_MPC_FUSE=$1 PAROPT_AMD_MPC_POLY=$2 PAROPT_AMD_SPEC_DT=$3 PAROPT_TR_SWEEP_SEED=727 PAROPT_TR_SWEEP_CASES=200 PAROPT_TR_SWEEP_ORACLE=1 python -m pytest tests/test_gpu_tr_sweep.py -q -m gpu -k "against_oracle and (54 or 97 or 146 or 165 or 183 or 184)" 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200
done
