import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_fuzz_gpu import _scene
from tests.util import assert_parity, fetch, make_product, render_oracle
bad = 0
for seed in range(100, 260):
    rng = np.random.default_rng(seed)
    w, h = [(64, 64), (128, 64), (96, 130), (33, 64), (256, 192), (64, 64)][int(rng.integers(0, 6))]
    mode = "Raytracer" if rng.integers(0, 3) == 0 else "Rasterizer"
    if mode == "Raytracer":
        h = w
    d = _scene(seed, num_worlds=int(rng.integers(3, 70)), width=w, height=h, mode=mode)
    for k in ("MRX_DEBUG_SLOTS", "MRX_GROUP_VIEWS", "MRX_XCD_SKEW", "MRX_XCD_ROTATE", "MRX_GROUP_TILES"):
        os.environ.pop(k, None)
    if rng.integers(0, 2):
        os.environ["MRX_DEBUG_SLOTS"] = str([16, 32, 64, 128, 256][int(rng.integers(0, 5))])
    if rng.integers(0, 2):
        os.environ["MRX_GROUP_VIEWS"] = str([1, 2, 4][int(rng.integers(0, 3))])
        os.environ["MRX_XCD_SKEW"] = str(int(rng.integers(0, 8)))
        os.environ["MRX_XCD_ROTATE"] = str(int(rng.integers(0, 2)))
    elif rng.integers(0, 2):
        os.environ["MRX_GROUP_TILES"] = str(int(rng.integers(1, 17)))
    os.environ["MRX_PLACEMENT_TRIES"] = str(int(rng.integers(1, 4)))
    ids = bool(rng.integers(0, 2))
    try:
        r = make_product(d, visibility=ids)
        got = fetch(r, visibility=ids, raytracer=(mode == "Raytracer"))
        assert_parity(got, render_oracle(d))
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, w, h, mode, dict((k, os.environ.get(k)) for k in ("MRX_DEBUG_SLOTS", "MRX_GROUP_VIEWS", "MRX_XCD_SKEW", "MRX_GROUP_TILES")), str(e)[:100], flush=True)
    del r
print("soak done, mismatching scenes:", bad, flush=True)
