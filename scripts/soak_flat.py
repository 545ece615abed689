"""Soak of the BVH path's flat kernel (worlds of <= 64 triangles): random triangle soups of
tests/test_fuzz_gpu.py under many seeds, view sizes and both modes, product (kernel_variant 2) against the oracle,
bit for bit.  python scripts/soak_flat.py [first seed = 1000] [count = 150]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

from tests.test_fuzz_gpu import _scene  # noqa: E402
from tests.util import assert_parity, fetch, make_product, render_oracle  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sizes = [(64, 64), (128, 128), (256, 256), (200, 136), (96, 48), (37, 53), (320, 64), (64, 192)]
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    mode = "Raytracer" if seed % 3 == 0 else "Rasterizer"
    w, h = sizes[int(rng.integers(0, len(sizes)))]
    if mode == "Raytracer":
        h = w
    os.environ["MRX_BVH_SMALL_AREA"] = str([256, 0, 8, 4096][seed % 4])
    d = _scene(seed, num_worlds=int(rng.integers(3, 40)), width=w, height=h, mode=mode)
    r = make_product(d, visibility=True, variant=2)
    try:
        assert r.render_path() == "bvh"
        assert_parity(fetch(r), render_oracle(d))
    except AssertionError as e:
        bad += 1
        print("seed %d %s %dx%d: %s" % (seed, mode, w, h, str(e)[:200]), flush=True)
    del r
print("flat-kernel soak: %d scenes, %d mismatches" % (count, bad), flush=True)
sys.exit(1 if bad else 0)
