"""CPU: register and scratch use of every kernel instantiation, read from the metadata of the gfx950 device assembly
(hipcc --cuda-device-only -S through `make resources`).

  * no kernel uses scratch memory (.private_segment_fixed_size == 0): nothing spills to memory, no local array lives
    there -- the register cliff of round 4 (solve2_dots_kernel<24,2,0> 123 spilled registers, <12,3,0> 26, ...;
    VERDICT r4 weak #4) is gone because the occupancy is now chosen per slot count so that every instantiation fits;
  * no kernel spills vector registers at all, except the widest one-workgroup-per-CU instantiations of the first
    solve pass, which park a handful of values in ACCUMULATOR registers of the unified file (copies inside the register
    file: the scratch size stays 0) -- listed here by name with their bound.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "paropt_amd", "csrc")

# vector-register copies into the accumulator half of the unified register file (no memory traffic)
AGPR_PARKED = {
    r"solve2_dots_kernelILi24ELi1ELi[01]E": 16,
    r"solve2_dots_kernelILi20ELi1ELi1E": 16,
}


def kernel_metadata():
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call(["make", "-C", CSRC, "-j8", "resources"], env=env, stdout=subprocess.DEVNULL)
    out = {}
    build = os.path.join(CSRC, "_build")
    for f in sorted(os.listdir(build)):
        if not f.endswith(".hip.s"):
            continue
        text = open(os.path.join(build, f)).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size:", text, re.S):
            body = m.group(2)
            vals = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|sgpr_spill_count|vgpr_count|"
                                                     r"vgpr_spill_count):\s+(\d+)", body)}
            out[m.group(1)] = vals
    return out


@pytest.fixture(scope="module")
def meta():
    return kernel_metadata()


def test_no_kernel_uses_scratch_memory(meta):
    assert len(meta) > 250, len(meta)
    bad = {k: v for k, v in meta.items() if v["private_segment_fixed_size"] != 0}
    assert not bad, "kernels with scratch memory: %s" % bad


def test_no_vector_register_spills(meta):
    for name, v in meta.items():
        allowed = 0
        for pat, bound in AGPR_PARKED.items():
            if re.search(pat, name):
                allowed = bound
        assert v["vgpr_spill_count"] <= allowed, "%s spills %d vector registers" % (name, v["vgpr_spill_count"])
    # the allow-list is not stale: each entry matches a kernel that exists
    for pat in AGPR_PARKED:
        assert any(re.search(pat, n) for n in meta), pat


def test_hot_kernels_of_the_metric_fit_their_occupancy(meta):
    """The instantiations config 3 runs: register counts within the budget of the occupancy they
    are compiled for (512 / OCC per lane, unified file)."""
    # (round 5: the tiled group kernels are launched eight workgroups per CU -- 64 registers per lane; left to itself
    # the compiler took 88-106 and five or four of the eight were resident, HISTORY R5.15)
    want = {r"mdot_kernelILi32E": 128, r"wgram_pc_kernelILi11ELi3ELi1ELi0E": 256, r"solve2_dots_kernelILi11ELi2ELi1E": 256,
            r"solve2r_kernelILi1ELi1E": 128, r"group_sum_tiled_kernelILi[01]E": 64, r"group_k0_tiled_kernel": 64,
            r"wgram_pc64_kernelILi(17|18|19|20)E": 256}
    for pat, budget in want.items():
        hits = [(n, v) for n, v in meta.items() if re.search(pat, n)]
        assert hits, pat
        for n, v in hits:
            assert v["vgpr_count"] <= budget and v["vgpr_spill_count"] == 0, (n, v)
