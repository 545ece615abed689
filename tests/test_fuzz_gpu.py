"""Randomised parity: raw-geometry scenes with triangles at every scale and
orientation -- crossing the eye plane, edge-on, far outside the frustum,
sub-pixel, coincident -- rendered by the HIP path and by the oracle must agree
bit for bit (visibility, colour) and to 1e-4 in depth."""
import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.util import assert_parity, fetch, make_product, render_oracle

pytestmark = pytest.mark.gpu


def _random_quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return tuple(float(np.float32(x)) for x in q)


def _scene(seed, num_worlds, width, height, mode):
    rng = np.random.default_rng(seed)
    verts, uvs, idx, voff, ioff, mats = [], [], [], [], [], []
    n_mesh = int(rng.integers(2, 6))
    for m in range(n_mesh):
        voff.append(len(verts))
        ioff.append(len(idx))
        scale = 10.0 ** rng.uniform(-2, 2)
        nv = int(rng.integers(3, 12))
        v = rng.normal(size=(nv, 3)) * scale
        if m == 0:                                   # a few exact duplicates / degenerate tris
            v[1] = v[0]
        verts += v.tolist()
        uvs += rng.uniform(-2, 3, size=(nv, 2)).tolist()
        nt = int(rng.integers(1, 9))
        idx += rng.integers(0, nv, size=3 * nt).tolist()
        mats.append(int(rng.integers(-1, 3)))
    instances, cameras, worlds = [], [], []
    for w in range(num_worlds):
        ni = int(rng.integers(0, 5))
        for _ in range(ni):
            pos = tuple(float(np.float32(x)) for x in rng.normal(size=3) * 10.0 ** rng.uniform(-1, 1.5))
            s = tuple(float(np.float32(x)) for x in rng.uniform(-2, 2, size=3))
            instances.append((pos, _random_quat(rng), s, int(rng.integers(0, n_mesh + 1))))
        nc = int(rng.integers(1, 3))
        for c in range(nc):
            cpos = tuple(float(np.float32(x)) for x in rng.normal(size=3) * 5.0)
            if ni and c == 0:
                # first camera looks at one of the world's instances (often from
                # very close: triangles cross the eye plane); the second is random
                tgt = np.asarray(instances[-1 - int(rng.integers(0, ni))][0]) + rng.normal(size=3) * 0.3
                cameras.append((cpos, scenes.look_at(cpos, tgt)))
            else:
                cameras.append((cpos, _random_quat(rng)))
        worlds.append((ni, len(instances) - ni, nc, len(cameras) - nc))
    import os
    return scenes.SceneDesc(
        num_worlds=num_worlds, render_mode=mode, width=width, height=height,
        mesh_vertices=np.asarray(verts, np.float32), mesh_uvs=np.asarray(uvs, np.float32),
        mesh_indices=np.asarray(idx, np.uint32), mesh_vertex_offsets=np.asarray(voff, np.uint32),
        mesh_indices_offsets=np.asarray(ioff, np.uint32), mesh_materials=np.asarray(mats, np.int32),
        materials=[((0.9, 0.5, 0.2, 1.0), -1, 0.5, 0.5), ((0.4, 0.7, 1.0, 1.0), 0, 0.5, 0.5),
                   ((1.0, 1.0, 1.0, 1.0), 5, 0.5, 0.5)],        # texture 5 does not exist
        texture_paths=[os.path.join(scenes.DATA_DIR, "cube.png")],
        instances=instances, cameras=cameras, worlds=worlds)


@pytest.mark.parametrize("seed,size,mode", [
    (1, (64, 64), "Rasterizer"), (2, (64, 64), "Rasterizer"), (3, (96, 48), "Rasterizer"),
    (4, (37, 53), "Rasterizer"), (5, (64, 64), "Raytracer"), (6, (100, 100), "Raytracer"),
    (7, (128, 64), "Rasterizer"), (8, (64, 64), "Rasterizer"),
])
def test_random_raw_geometry_scenes(native, seed, size, mode):
    d = _scene(seed, num_worlds=24, width=size[0], height=size[1], mode=mode)
    r = make_product(d, visibility=True)
    got = fetch(r)
    ref = render_oracle(d)
    assert_parity(got, ref)
    assert (ref["tri_id"] >= 0).mean() > 0.01          # the scenes do draw something


@pytest.mark.parametrize("seed", range(20, 36))
def test_random_scenes_under_random_kernel_shapes(native, monkeypatch, seed):
    # the same kind of scene under a random choice of the knobs that select the
    # kernel instantiation and the workgroup shape (none may change a byte):
    # triangle slots, whole views / tiles per workgroup, XCD split, store policy,
    # with or without the id tensor
    rng = np.random.default_rng(1000 + seed)
    w, h = [(64, 64), (64, 64), (128, 64), (96, 130), (33, 64), (256, 192)][int(rng.integers(0, 6))]
    mode = "Raytracer" if rng.integers(0, 3) == 0 else "Rasterizer"
    if mode == "Raytracer":
        h = w
    d = _scene(seed, num_worlds=int(rng.integers(5, 41)), width=w, height=h, mode=mode)
    env = {"MRX_DEBUG_SLOTS": str([16, 32, 64, 128, 256][int(rng.integers(0, 5))])}
    if rng.integers(0, 2):
        env["MRX_GROUP_VIEWS"] = str([1, 2, 4][int(rng.integers(0, 3))])
        env["MRX_XCD_SKEW"] = str(int(rng.integers(0, 8)))
        env["MRX_XCD_ROTATE"] = str(int(rng.integers(0, 2)))
    else:
        env["MRX_GROUP_TILES"] = str(int(rng.integers(1, 17)))
    if rng.integers(0, 4) == 0:
        env["MRX_WRITE_THROUGH"] = "0"
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ids = bool(rng.integers(0, 2))
    r = make_product(d, visibility=ids)
    got = fetch(r, visibility=ids, raytracer=(mode == "Raytracer"))
    ref = render_oracle(d)
    assert_parity(got, ref)


def test_random_scene_on_brute_variant_too(native):
    d = _scene(11, num_worlds=16, width=64, height=64, mode="Rasterizer")
    r = make_product(d, visibility=True, variant=1)
    assert_parity(fetch(r), render_oracle(d))


@pytest.mark.parametrize("seed,size,mode,tile", [
    (41, (64, 64), "Rasterizer", 0), (50, (96, 48), "Rasterizer", 0), (51, (37, 53), "Rasterizer", 1),
    (44, (64, 64), "Raytracer", 0), (45, (100, 100), "Raytracer", 2), (52, (128, 64), "Rasterizer", 0),
    (47, (64, 64), "Rasterizer", 2), (56, (200, 136), "Rasterizer", 0), (58, (64, 64), "Rasterizer", 1),
    (63, (160, 160), "Raytracer", 0),
])
def test_random_raw_geometry_scenes_through_the_bvh_path(native, monkeypatch, seed, size, mode, tile):
    # the same soup of triangles at every scale -- crossing the eye plane (unbounded screen
    # rectangles), edge-on, sub-pixel, coincident (ties) -- forced through the BVH path: TLAS
    # rectangles, the (triangle, row) walk of small boxes, the shared list of large ones and
    # the LDS depth buffer must agree with the oracle bit for bit, for every tile shape
    monkeypatch.setenv("MRX_BVH_TILE", str(tile))
    if seed % 2:
        monkeypatch.setenv("MRX_BVH_SMALL_AREA", str([0, 8, 200, 4096][seed % 4]))
    d = _scene(seed, num_worlds=24, width=size[0], height=size[1], mode=mode)
    r = make_product(d, visibility=True, variant=2)
    got = fetch(r)
    ref = render_oracle(d)
    assert_parity(got, ref)
    assert (ref["tri_id"] >= 0).mean() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size,mode,tile", [(300, (64, 64), "Rasterizer", 0), (303, (128, 128), "Raytracer", 0),
                                                 (306, (96, 128), "Rasterizer", 1), (309, (64, 64), "Raytracer", 2),
                                                 (312, (33, 64), "Rasterizer", 0)])
def test_mesh_worlds_seen_by_random_cameras(native, monkeypatch, seed, size, mode, tile):
    # cameras on the terrain, inside object boxes, looking away: triangles behind and across the eye
    # plane, unbounded boxes (scripts/soak_bvh.py runs 150 such scenes)
    from tests import meshes
    monkeypatch.setenv("MRX_BVH_TILE", str(tile))
    d = meshes.mesh_scene_random_cameras(seed, size[0], size[1], mode)
    r = make_product(d, visibility=True)
    assert r.render_path() == "bvh"
    ref = render_oracle(d)
    assert_parity(fetch(r), ref)
    assert (ref["tri_id"] >= 0).mean() > 0.2


@pytest.mark.parametrize("seed,size,mode", [(41, (64, 64), "Rasterizer"), (44, (64, 64), "Raytracer"),
                                            (56, (200, 136), "Rasterizer"), (63, (160, 160), "Raytracer")])
@pytest.mark.parametrize("flat", ["0", "1"], ids=["general-kernel", "flat-kernel"])
def test_small_worlds_through_both_bvh_kernels(native, monkeypatch, seed, size, mode, flat):
    # worlds of at most 64 triangles take bvhFlatKernel on the BVH path since round 4 (one set-up per view,
    # DESIGN.md 4.2b); MRX_BVH_FLAT=0 keeps them on bvhTileKernel.  Both must equal the oracle -- and each other --
    # on the triangle soups above (multi-camera worlds, empty worlds, eye-plane crossings, ties), and on the
    # BASELINE textured Raytracer shape at a size the oracle renders in a second.
    monkeypatch.setenv("MRX_BVH_FLAT", flat)
    d = _scene(seed, num_worlds=24, width=size[0], height=size[1], mode=mode)
    r = make_product(d, visibility=True, variant=2)
    assert r.render_path() == "bvh"
    assert_parity(fetch(r), render_oracle(d))
    d5 = scenes.synthetic_scene(40, width=256, height=256, textured=True, render_mode="Raytracer")
    r5 = make_product(d5, visibility=False, variant=2)
    assert_parity(fetch(r5, visibility=False, raytracer=True), render_oracle(d5))
