set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3"
python3 bench.py $c4 > /dev/null 2>&1
for round in 1 2 3; do
  for occ in 2 0; do
    PAROPT_AMD_S2D_OCC=$occ python3 bench.py $c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('S2D_OCC=$occ round $round: %.2f it/s %.3f ms kkt_step %.3f' % (d['value'], d['ms_per_step'], d['phase_ms_per_iter']['kkt_step']))"
  done
done
