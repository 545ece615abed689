"""Evidence guard: no kernel instantiation may spill vector registers or use
scratch -- the round-3 BVH kernels got their zero-scratch property by hand and a
careless edit loses it silently: scratch shows up as HBM traffic, not as a failure.  The compiler's
own report (-Rpass-analysis=kernel-resource-usage, scripts/kernel_resources.py) needs no GPU.  The
committed table profiles/kernel_resources_latest.txt must be the one HEAD compiles to."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))

SOURCES = ["madrona_renderer_amd/csrc/raster.hip", "madrona_renderer_amd/csrc/bvh.hip"]


@pytest.fixture(scope="module")
def tables():
    import kernel_resources
    with ThreadPoolExecutor(2) as ex:
        return dict(zip(SOURCES, ex.map(lambda s: kernel_resources.resources(os.path.join(ROOT, s)), SOURCES)))


def test_no_instantiation_spills_vector_registers_or_uses_scratch(tables):
    seen = 0
    for src, kernels in tables.items():
        assert kernels, src
        for k in kernels:
            seen += 1
            assert int(k["ScratchSize [bytes/lane]"]) == 0, (src, k["name"], k)
            assert int(k["VGPRs Spill"]) == 0, (src, k["name"], k)
    assert seen >= 90          # 54 raster + 36 BVH instantiations in round 3, the flat kernels since


def test_the_bvh_kernels_keep_two_workgroups_per_cu_and_the_group_kernels_eight_waves_per_simd(tables):
    for k in tables[SOURCES[1]]:
        assert int(k["VGPRs"]) <= 128, k                     # 8-wave workgroups, 2 per CU: 4 waves per SIMD
    fast = [k for k in tables[SOURCES[0]] if k["name"].startswith("rasterGroupKernelFast<false, false")]
    assert fast and all(int(k["VGPRs"]) <= 64 for k in fast)  # the headline kernel: 8 waves per SIMD


def test_committed_resource_table_matches_head(tables):
    import kernel_resources
    want = []
    for src in SOURCES:
        want.append("# " + src)
        want += [kernel_resources.line(k) for k in tables[src]]
    path = os.path.join(ROOT, "profiles", "kernel_resources_latest.txt")
    have = [l.rstrip("\n") for l in open(path) if not l.startswith("##")]
    assert have == want, "stale: regenerate with `python scripts/kernel_resources.py --write`"
