"""Longer randomized parity run of the BVH path (GPU box): (1) the fuzz generator's raw-geometry
scenes forced through it with random tile shape / small-area threshold / TLAS pass size;
(2) mesh worlds (terrain, torus, spheres: three-level BLASes) seen by random cameras -- on the
terrain, inside boxes, looking away -- the cases the behind-the-eye culling and the
tile-corner classification decide.   python scripts/soak_bvh.py [first_seed [count]]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from madrona_renderer_amd import scenes
from tests import meshes
from tests.test_fuzz_gpu import _scene
from tests.util import assert_parity, fetch, make_product, render_oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 300
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
KNOBS = ("MRX_BVH_TILE", "MRX_BVH_SMALL_AREA", "MRX_BVH_PASS_INST")


def check(d, seed, what, ids, rt):
    global bad
    r = None
    try:
        r = make_product(d, visibility=ids, variant=2)
        assert r.render_path() == "bvh"
        got = fetch(r, visibility=ids, raytracer=rt)
        assert_parity(got, render_oracle(d))
    except AssertionError as e:
        bad += 1
        print("MISMATCH", what, "seed", seed, dict((k, os.environ.get(k)) for k in KNOBS), str(e)[:120], flush=True)
    del r


for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ["MRX_BVH_TILE"] = str(int(rng.integers(0, 3)))
    if rng.integers(0, 2):
        os.environ["MRX_BVH_SMALL_AREA"] = str([0, 8, 32, 200, 4096][int(rng.integers(0, 5))])
    if rng.integers(0, 3) == 0:
        os.environ["MRX_BVH_PASS_INST"] = str([64, 128, 256][int(rng.integers(0, 3))])
    os.environ["MRX_PLACEMENT_TRIES"] = "1"
    w, h = [(64, 64), (128, 64), (96, 130), (33, 64), (200, 136), (64, 64)][int(rng.integers(0, 6))]
    mode = "Raytracer" if rng.integers(0, 3) == 0 else "Rasterizer"
    if mode == "Raytracer":
        h = w
    ids = bool(rng.integers(0, 2))
    if seed % 3 == 0:
        check(meshes.mesh_scene_random_cameras(seed, min(w, 128), min(h, 128), mode), seed, "meshes", ids, mode == "Raytracer")
    else:
        check(_scene(seed, num_worlds=int(rng.integers(3, 40)), width=w, height=h, mode=mode), seed, "fuzz", ids,
              mode == "Raytracer")
    if (seed - first) % 20 == 19:
        print("... through seed", seed, "mismatches so far:", bad, flush=True)
print("soak done, mismatching scenes:", bad, flush=True)

# (3) bench-sized scenes, BVH path against the tiled raster kernels byte for byte (no oracle: it would take minutes)
import gc
for name, d in (("1024 x 482", meshes.cube_field(1024, 40)), ("1024 x 1202 RT", meshes.cube_field(1024, 100, mode="Raytracer")),
                ("256 x 4994 textured", meshes.cube_field(256, 416, textured=True)),
                ("64 x 14152 meshes 128^2", meshes.mesh_worlds(64, 128, 128)),
                ("256 x 14152 meshes RT", meshes.mesh_worlds(256, 64, 64, "Raytracer"))):
    for k in KNOBS:
        os.environ.pop(k, None)
    rt = d.render_mode == "Raytracer"
    outs = []
    for variant in (2, 3):
        r = make_product(d, visibility=not rt, variant=variant)
        outs.append(fetch(r, visibility=not rt, raytracer=rt))
        del r
        gc.collect()
    same = all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
    bad += 0 if same else 1
    print("bvh == raster on", name, ":", same, flush=True)
print("soak done incl. differential, mismatching scenes:", bad, flush=True)
