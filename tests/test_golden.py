"""Golden fixtures (tests/golden/, produced by make_golden.py from the CPU
oracle): the oracle must keep reproducing them on CPU, the HIP path must match
them bit for bit on the MI355X."""
import json
import os

import numpy as np
import pytest

from tests.golden.make_golden import cases
from tests.util import digest, fetch, make_product, render_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "golden.json")))
VIEWS = np.load(os.path.join(HERE, "golden", "golden_views.npz"))
CASES = cases()


def _check(name, out, exact_depth):
    g = META[name]
    assert list(out["rgb"].shape) == g["shape"]
    assert digest(out["rgb"]) == g["rgb_sha256"]
    assert digest(out["tri_id"]) == g["tri_id_sha256"]
    assert np.array_equal(out["rgb"][0], VIEWS[name + "/rgb"])
    assert np.array_equal(out["tri_id"][0], VIEWS[name + "/tri_id"].astype(np.int32))
    if exact_depth:
        assert digest(out["depth"]) == g["depth_sha256"]
    np.testing.assert_allclose(out["depth"][0], VIEWS[name + "/depth"], rtol=1e-4, atol=0)
    for p in g["probes"]:
        at = tuple(p["at"])
        assert out["rgb"][at].tolist() == p["rgb"]
        assert int(out["tri_id"][at]) == p["tri_id"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle_mod, name):
    out = render_oracle(CASES[name])
    _check(name, out, exact_depth=True)
    assert digest(out["segmask"]) == META[name]["segmask_sha256"]
    assert int((out["tri_id"] >= 0).sum()) == META[name]["covered"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_golden(native, name):
    r = make_product(CASES[name], visibility=True)
    _check(name, fetch(r), exact_depth=False)
