#!/usr/bin/env bash
# tuning aid: wgram with parts switched off (PAROPT_AMD_WGRAM_ABLATE: 1 no matrix work, 2 no staging either,
# 3 no loads) at both compiled occupancies
set -u
for o in 3 2; do
  for a in 0 1 3; do
    PAROPT_AMD_WGRAM_OCC=$o PAROPT_AMD_WGRAM_ABLATE=$a python tools/microbench.py --tag occ${o}_ab$a --reps 3 | grep wgram
  done
done
