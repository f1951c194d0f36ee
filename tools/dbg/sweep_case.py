"""Runs single cases of the random sweep (tests/test_gpu_random_sweep.py) and prints whether they agree with the oracle.
usage: PAROPT_SWEEP_CASES=N PAROPT_SWEEP_SEED=S python tools/dbg/sweep_case.py i j ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import test_gpu_random_sweep as S
c = S._make_ctx()
for a in sys.argv[1:]:
    i = int(a)
    try:
        S.test_random_case_against_oracle(c, i)
        print("case", i, "agrees", flush=True)
    except AssertionError as e:
        print("case", i, "DIFFERS:", " | ".join(str(e).strip().splitlines()[:4])[:300], flush=True)
