set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for a in 0 6 1; do PAROPT_AMD_WGRAM_ABLATE=$a python3 tools/dbg/gram_groups_ablate.py 25 2>&1 | tail -3; done
python -m pytest tests/test_gpu_vec.py -m gpu -q -k "structured_panel or declines" 2>&1 | tail -3
