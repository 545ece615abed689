"""Parity of the BVH ray-trace path (madrona_renderer_amd/csrc/bvh.hip: per-step
TLAS over the world's instances, per-object BLAS, exact S6 leaf test) against
the CPU oracle, which knows no hierarchy at all: every triangle at every pixel.

Covers the counterpart of the reference's Raytracer render graph
(/root/reference/src/mgr.cpp:443-492): worlds of 500 - 10,000 triangles as many
instances of small objects (TLAS) and as few instances of large meshes (BLAS),
both render modes, ties between coincident triangles, eyes inside boxes."""
import ctypes
import math
import os

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests import meshes
from tests.meshes import CUBE, IDENT, PLANE
from tests.util import assert_parity, fetch, make_product, render_oracle

BVH = 2          # mrx_config.kernel_variant: the BVH path whatever the scene size


def _info(native, r):
    lib = native.load_capi()

    class Info(ctypes.Structure):
        _fields_ = [(n, ctypes.c_uint32) for n in
                    ("num_worlds", "num_views", "num_instances", "num_objects", "num_triangles",
                     "num_materials", "num_textures", "max_world_triangles", "storage_fast",
                     "storage_slow")] + \
                   [("device_id", ctypes.c_int32), ("kernel_variant", ctypes.c_int32),
                    ("bytes_per_step", ctypes.c_uint64), ("render_path", ctypes.c_int32),
                    ("bvh_nodes", ctypes.c_uint32), ("bvh_depth", ctypes.c_uint32),
                    ("max_world_instances", ctypes.c_uint32)]
    inf = Info()
    assert lib.mrx_info(ctypes.c_void_p(r.native_handle()), ctypes.byref(inf)) == 0
    return inf


def _parity(desc, variant=None, visibility=True):
    r = make_product(desc, visibility=visibility, variant=variant)
    rt = desc.render_mode == "Raytracer"
    got = fetch(r, visibility=visibility, raytracer=rt)
    ref = render_oracle(desc)
    assert_parity(got, ref)
    return r, got, ref


# ---------------------------------------------------------------------------
# host side (no GPU): the BLAS builder
# ---------------------------------------------------------------------------
def _blas_check(lib, tri_pos):
    tri_pos = np.ascontiguousarray(tri_pos, np.float32).reshape(-1, 9)
    nodes, depth, leaves = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    lib.mrx_last_error.restype = ctypes.c_char_p
    rc = lib.mrx_blas_check(tri_pos.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(tri_pos)),
                            ctypes.byref(nodes), ctypes.byref(depth), ctypes.byref(leaves))
    assert rc == 0, lib.mrx_last_error()
    return nodes.value, depth.value, leaves.value


def _soup(mesh):
    v, _, i = mesh
    return v[i.astype(np.int64)].reshape(-1, 3, 3)


def test_blas_builder_invariants(native):
    lib = native.load_capi()
    assert _blas_check(lib, np.zeros((0, 9), np.float32)) == (0, 0, 0)
    cube = _soup(meshes.sphere(4, 2))                    # 16 triangles: flat, no hierarchy
    assert _blas_check(lib, cube) == (0, 0, 0)
    for mesh, tris in ((meshes.sphere(32, 16), 1024), (meshes.torus(48, 24), 2304),
                       (meshes.terrain(70), 9800)):
        soup = _soup(mesh)
        assert len(soup) == tris
        nodes, depth, leaves = _blas_check(lib, soup)
        assert leaves >= tris / 16 and nodes >= leaves / 8
        assert 1 <= depth <= 6                           # 8-wide: log8(9800 / 16) ~ 3
    # every centroid in one place (nothing to split on), and a long thin line of
    # triangles (as unbalanced as the surface-area heuristic gets)
    same = np.tile(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), (500, 1))
    _blas_check(lib, same)
    line = np.array([[[i, 0, 0], [i + 0.5, 1, 0], [i + 1.0, 0, 0]] for i in range(3000)], np.float32)
    line[:, :, 0] = line[:, :, 0] ** 2 * 1e-3            # sizes grow along the line
    nodes, depth, leaves = _blas_check(lib, line)
    assert 1 + 7 * depth <= 64


# ---------------------------------------------------------------------------
# GPU parity
# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(num_worlds=40),
    dict(num_worlds=33, with_wall=True, textured=True),
    dict(num_worlds=6, width=128, height=128, with_wall=True),
    dict(num_worlds=5, width=96, height=40),                       # ragged tiles
    dict(num_worlds=6, width=50, height=30),                       # scalar stores
    dict(num_worlds=7, width=80, height=80, with_wall=True, textured=True, render_mode="Raytracer"),
    dict(num_worlds=4, width=256, height=256, textured=True, render_mode="Raytracer"),
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_bvh_path_forced_on_the_small_baseline_scenes(native, kw):
    # cube + plane (+ wall): flat objects, the TLAS alone selects the triangles
    r, _, _ = _parity(scenes.synthetic_scene(**kw), variant=BVH)
    assert _info(native, r).render_path == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Rasterizer", "Raytracer"])
def test_bvh_path_on_the_reference_demo_scene(native, mode):
    # scripts/test.py's scene: a triangle that crosses the eye plane (unbounded
    # screen rectangle) and worlds aliasing the same rows
    _parity(scenes.demo_scene(num_worlds=4, render_mode=mode), variant=BVH)


@pytest.mark.gpu
@pytest.mark.parametrize("cubes,kw", [
    (40, dict(num_worlds=48)),                                      # 482 triangles, 41 instances
    (40, dict(num_worlds=9, mode="Raytracer", textured=True)),
    (100, dict(num_worlds=12, width=96, height=72)),                # 1202 triangles, 101 instances
    (130, dict(num_worlds=5, width=128, height=128, mode="Raytracer")),   # two TLAS passes of 128
    (416, dict(num_worlds=3, textured=True)),                       # 4994 triangles, four passes
    (416, dict(num_worlds=2, width=160, height=160, mode="Raytracer")),
], ids=lambda v: str(v) if isinstance(v, int) else "-".join(f"{a}{b}" for a, b in v.items()))
def test_many_instance_worlds_take_the_bvh_path(native, cubes, kw):
    d = meshes.cube_field(cubes=cubes, **kw)
    r, got, ref = _parity(d)                      # default dispatch: > 256 triangles per world
    inf = _info(native, r)
    assert inf.render_path == 1 and inf.max_world_triangles == 12 * cubes + 2
    assert inf.max_world_instances == cubes + 1
    assert (ref["tri_id"] >= 0).mean() > 0.5
    # ... and the tiled raster kernels (forced) give the same bytes
    r3 = make_product(d, visibility=True, variant=3)
    assert _info(native, r3).render_path == 0
    got3 = fetch(r3)
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(got[k], got3[k])


def _mesh_world(mode, width, height, textured=True):
    sph, tor, ter = meshes.sphere(32, 16), meshes.torus(48, 24), meshes.terrain(70)
    geo = meshes.pack_meshes([(sph[0], sph[1], sph[2], 0), (tor[0], tor[1], tor[2], 1),
                              (ter[0], ter[1], ter[2], 2)])
    rng = np.random.default_rng(17)
    inst = [((0.0, 0.0, -1.0), IDENT, (1.0, 1.0, 1.0), 3),                       # terrain (object 3)
            ((0.0, 0.0, 2.5), meshes.random_quat(rng), (1.5, 1.0, 0.7), 2),      # torus, non-uniform scale
            ((4.0, -3.0, 2.0), meshes.random_quat(rng), (1.2, 1.2, 1.2), 1),     # sphere
            ((-5.0, 2.0, 1.5), meshes.random_quat(rng), (-1.0, 1.4, 1.0), 1),    # mirrored sphere
            ((2.0, 5.0, 1.0), IDENT, (2.0, 2.0, 2.0), 0),                        # cube asset
            ((-3.0, -6.0, 3.0), meshes.random_quat(rng), (0.8, 0.8, 0.8), 2)]
    cams = []
    for az, r, h in ((0.3, 14.0, 6.0), (2.1, 9.0, 3.0), (4.0, 20.0, 12.0), (5.5, 3.0, 1.0)):
        eye = (r * math.cos(az), r * math.sin(az), h)
        cams.append((eye, scenes.look_at(eye, (0.0, 0.0, 1.0))))
    cams.append(((0.5, 0.4, 2.5), scenes.look_at((0.5, 0.4, 2.5), (3.0, 0.0, 2.0))))   # inside the torus' box
    cams.append(((0.0, 0.0, 40.0), (0.7071068, -0.7071068, 0.0, 0.0)))                   # straight down
    return scenes.SceneDesc(
        num_worlds=2, render_mode=mode, width=width, height=height, asset_paths=[(CUBE, 0)],
        materials=[((0.9, 0.8, 0.7, 1.0), 0 if textured else -1, 0.5, 0.5),
                   ((0.3, 0.5, 0.9, 1.0), -1, 0.5, 0.5), ((0.4, 0.7, 0.3, 1.0), -1, 0.5, 0.5)],
        texture_paths=[os.path.join(scenes.DATA_DIR, "cube.png")],
        instances=inst, cameras=cams, worlds=[(6, 0, 4, 0), (4, 0, 2, 4)], **geo)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,size", [("Rasterizer", (64, 64)), ("Raytracer", (128, 128)),
                                       ("Rasterizer", (200, 136))])
def test_large_meshes_through_the_blas(native, mode, size):
    # sphere 1024, torus 2304, terrain 9800 triangles: three-level BLASes; closed
    # meshes (S6b culls their back faces), a mirrored instance, an eye inside a box
    d = _mesh_world(mode, *size)
    r, got, ref = _parity(d)
    inf = _info(native, r)
    assert inf.render_path == 1 and inf.bvh_nodes > 100 and inf.bvh_depth >= 2
    assert inf.max_world_triangles == 9800 + 2304 * 2 + 1024 * 2 + 12
    assert len(np.unique(ref["tri_id"])) > 500


@pytest.mark.gpu
def test_ties_between_coincident_triangles_go_to_the_lower_index(native):
    # the same sphere twice in one place, in one object and as two instances of
    # two objects: every covered pixel is an exact tie in 1/depth; the oracle's
    # in-order scan keeps the first triangle, whatever order the hierarchy is
    # walked in
    sph = meshes.sphere(24, 12)
    twice = (np.concatenate([sph[0], sph[0]]), np.concatenate([sph[1], sph[1]]),
             np.concatenate([sph[2], sph[2] + len(sph[0])]))
    geo = meshes.pack_meshes([(twice[0], twice[1], twice[2], 0), (sph[0], sph[1], sph[2], 1),
                              (sph[0], sph[1], sph[2], 0)])
    q = (0.9238795, 0.0, 0.3826834, 0.0)
    d = scenes.SceneDesc(
        num_worlds=2, width=64, height=64,
        materials=[((0.9, 0.3, 0.2, 1.0), -1, 0.5, 0.5), ((0.2, 0.3, 0.9, 1.0), -1, 0.5, 0.5)],
        instances=[((0.0, 4.0, 0.0), q, (1.5, 1.5, 1.5), 0),
                   ((0.0, 4.0, 0.0), q, (1.5, 1.5, 1.5), 2), ((0.0, 4.0, 0.0), q, (1.5, 1.5, 1.5), 1)],
        cameras=[((0.0, 0.0, 0.0), IDENT)], worlds=[(1, 0, 1, 0), (2, 1, 1, 0)], **geo)
    _, got, ref = _parity(d, variant=BVH)
    n = len(sph[2]) // 3
    hit = ref["tri_id"][0] >= 0
    assert hit.sum() > 500 and (ref["tri_id"][0][hit] < n).all()      # first copy wins
    hit1 = ref["tri_id"][1] >= 0
    assert hit1.sum() > 500 and (ref["tri_id"][1][hit1] < n).all()    # first instance wins
    px = got["rgb"][1][hit1].astype(int)
    assert (px[:, 0] > px[:, 2]).all()                                # ... the red one


@pytest.mark.gpu
def test_bvh_edge_cases_empty_worlds_bad_ids_no_cameras(native):
    sph = meshes.sphere(16, 8)
    geo = meshes.pack_meshes([(sph[0], sph[1], sph[2], -1)])
    d = scenes.SceneDesc(
        num_worlds=4, width=64, height=64, asset_paths=[(CUBE, -1)],
        instances=[((0.0, 5.0, 0.0), IDENT, (1.0, 1.0, 1.0), 1),
                   ((0.0, 4.0, 0.0), IDENT, (0.0, 0.0, 0.0), 1),       # zero scale
                   ((0.0, 2.0, 0.0), IDENT, (1.0, 1.0, 1.0), 9),       # no such object
                   ((0.0, 2.0, 0.0), IDENT, (1.0, 1.0, 1.0), -1),
                   ((1.0, 6.0, 0.5), IDENT, (1.0, 1.0, 1.0), 0)],
        cameras=[((0.0, 0.0, 0.0), IDENT), ((0.0, 10.0, 0.0), (0.0, 0.0, 0.0, 1.0))],
        worlds=[(5, 0, 1, 0), (0, 0, 1, 1), (2, 2, 2, 0), (1, 0, 0, 0)], **geo)
    r, got, ref = _parity(d, variant=BVH)
    assert got["rgb"].shape[0] == 4
    assert (got["tri_id"][1] == -1).all() and (got["depth"][2] == 0).all()
    assert (got["tri_id"][0] >= 0).any()


@pytest.mark.gpu
def test_bvh_path_sees_pose_writes_and_is_deterministic(native):
    import torch
    d = meshes.cube_field(num_worlds=6, cubes=60)
    r = make_product(d)
    a = fetch(r)
    r.step()
    b = fetch(r)
    for k in a:
        assert np.array_equal(a[k], b[k])
    pos = r.instance_position_tensor().to_torch()
    cam = r.camera_position_tensor().to_torch()
    pos[5, 2] += 0.75
    pos[61 * 3 + 7, 0] -= 1.25
    cam[2, 2] += 2.0
    r.step()
    got = fetch(r)
    inst, cams = list(d.instances), list(d.cameras)

    def bump(t, axis, dv):
        v = list(t)
        v[axis] = float(np.float32(np.float32(v[axis]) + np.float32(dv)))
        return tuple(v)
    inst[5] = (bump(inst[5][0], 2, 0.75),) + inst[5][1:]
    inst[61 * 3 + 7] = (bump(inst[61 * 3 + 7][0], 0, -1.25),) + inst[61 * 3 + 7][1:]
    cams[2] = (bump(cams[2][0], 2, 2.0), cams[2][1])
    d.instances, d.cameras = inst, cams
    assert_parity(got, render_oracle(d))


@pytest.mark.gpu
@pytest.mark.parametrize("pass_inst", [64, 128, 256])
def test_tlas_passes_of_every_size_give_the_same_bytes(native, monkeypatch, pass_inst):
    monkeypatch.setenv("MRX_BVH_PASS_INST", str(pass_inst))
    _parity(meshes.cube_field(num_worlds=4, cubes=150, width=96, height=64, textured=True))


@pytest.mark.gpu
@pytest.mark.parametrize("views_per_group", [1, 2, 4, 8])
@pytest.mark.parametrize("case", ["cubes-7-worlds", "cubes-101-instances", "textured-rt", "small-views", "ragged-two-cameras", "meshes", "hidden"])
def test_groups_of_one_tile_views_give_the_same_bytes(native, monkeypatch, views_per_group, case):
    # one-tile views whose worlds fit one TLAS pass: a workgroup renders `views_per_group` consecutive views, their
    # TLASes built side by side by different waves (bvh.hip, MULTI; the host picks 2 from 1024 views on).  View
    # counts that do not divide by the group, worlds of different sizes in one group, views that share a world.
    monkeypatch.setenv("MRX_BVH_GROUP_VIEWS", str(views_per_group))
    if case == "cubes-7-worlds":
        d = meshes.cube_field(num_worlds=7, cubes=40)
    elif case == "cubes-101-instances":
        # (TLAS blocks of 104 records: what the host gives two views per workgroup at 1024 views and more)
        d = meshes.cube_field(num_worlds=5, cubes=100)
    elif case == "textured-rt":
        d = meshes.cube_field(num_worlds=5, cubes=30, width=64, height=64, mode="Raytracer", textured=True)
    elif case == "small-views":
        d = scenes.synthetic_scene(9, width=50, height=44, with_wall=True, textured=True)
    elif case == "ragged-two-cameras":
        # worlds of 21, 17, 0 and 21 instances; the first and the last seen by two cameras each
        d = meshes.cube_field(num_worlds=6, cubes=20)
        d.num_worlds = 4
        d.worlds = [(21, 0, 2, 0), (17, 21, 1, 2), (0, 42, 1, 3), (21, 63, 2, 4)]
    elif case == "meshes":
        d = _mesh_world("Rasterizer", 64, 64)
    else:
        d = meshes.cube_field(num_worlds=6, cubes=20)
        d.instances = [(p, q, s, -1 if i % 5 == 0 else o) for i, (p, q, s, o) in enumerate(d.instances)]
    _parity(d, variant=BVH)


@pytest.mark.gpu
@pytest.mark.parametrize("prio", [0, 1, 2, 3])
def test_wave_priority_of_the_younger_workgroups_changes_no_pixel(native, monkeypatch, prio):
    # launches whose groups of views all run at once give the workgroups dispatched second to a CU wave priority 1
    # for their first view (bvh.hip; the host picks mode 2 from 1024 views on): every mode, on more groups than
    # the device has CUs, against the oracle (sampled) and against the raster kernels (all views)
    monkeypatch.setenv("MRX_BVH_GROUP_VIEWS", "2")
    monkeypatch.setenv("MRX_BVH_PRIO", str(prio))
    d = meshes.cube_field(num_worlds=640, cubes=12, textured=prio == 3)
    r, got, ref = _parity(d, variant=BVH)
    got3 = fetch(make_product(d, visibility=True, variant=3))
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(got[k], got3[k])


@pytest.mark.gpu
def test_small_batches_of_untextured_views_cross_over_to_the_bvh_path_earlier(native):
    # the default dispatch: from 129 triangles per world on, but from 65 for up to 640 64x64 views and from 91 for
    # up to 1024 untextured ones (profiles/r03_bvh_threshold.txt) -- same pixels either way
    d = meshes.cube_field(num_worlds=96, cubes=8)                     # 98 triangles
    r, got, ref = _parity(d)
    assert _info(native, r).render_path == 1
    got3 = fetch(make_product(d, visibility=True, variant=3))
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(got[k], got3[k])
    assert _info(native, make_product(meshes.cube_field(num_worlds=96, cubes=5))).render_path == 0        # 62 triangles
    assert _info(native, make_product(meshes.cube_field(num_worlds=96, cubes=8, textured=True))).render_path == 1
    assert _info(native, make_product(meshes.cube_field(num_worlds=700, cubes=8, textured=True))).render_path == 0
    assert _info(native, make_product(meshes.cube_field(num_worlds=96, cubes=8, width=128, height=64))).render_path == 0
    assert _info(native, make_product(meshes.cube_field(num_worlds=700, cubes=6))).render_path == 0       # 74 triangles
    assert _info(native, make_product(meshes.cube_field(num_worlds=700, cubes=8))).render_path == 1
    assert _info(native, make_product(meshes.cube_field(num_worlds=1100, cubes=8))).render_path == 0


@pytest.mark.gpu
def test_bench_shape_1024_worlds_482_triangles(native):
    # the shape VERDICT r1 quotes for the chunked raster kernel (94 us): 1024
    # worlds x 64x64, 40 cubes + plane; sampled views against the oracle, all
    # views against the raster kernels
    d = meshes.cube_field(num_worlds=1024, cubes=40)
    r = make_product(d, visibility=False)
    got = fetch(r, visibility=False)
    ref = render_oracle(d, want_ids=False)
    assert_parity(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [1, 2], ids=["64x32", "32x32"])
@pytest.mark.parametrize("case", ["cubes", "meshes-rt", "ragged-textured", "baseline-c3"])
def test_tile_shapes_of_the_lds_depth_buffer_give_the_same_bytes(native, monkeypatch, tile, case):
    # BASELINE configs[2] asks for an LDS tile-size sweep (64x64 / 64x32 / 32x32): the
    # tile of the BVH kernel is a template parameter; every shape must render the
    # oracle's image
    monkeypatch.setenv("MRX_BVH_TILE", str(tile))
    if case == "cubes":
        d = meshes.cube_field(num_worlds=6, cubes=130, width=96, height=64, textured=True)
    elif case == "meshes-rt":
        d = _mesh_world("Raytracer", 128, 128)
    elif case == "ragged-textured":
        d = scenes.synthetic_scene(5, width=50, height=94, with_wall=True, textured=True)
    else:
        d = scenes.synthetic_scene(12, width=128, height=128, with_wall=True)
    _parity(d, variant=BVH)


@pytest.mark.gpu
@pytest.mark.parametrize("classify", [0, 1])
@pytest.mark.parametrize("case", ["cubes", "meshes", "close-up"])
def test_strip_classification_of_large_triangles_is_exact(native, monkeypatch, case, classify):
    # the CLS instantiation (chosen by the host for scenes with BLAS meshes) and the plain
    # one give the oracle's bytes on scenes of either kind, and on a close-up where every
    # triangle is large and the list overflows (the round ends, the batch is taken again)
    monkeypatch.setenv("MRX_BVH_CLASSIFY", str(classify))
    if case == "cubes":
        d = meshes.cube_field(24, 40)
    elif case == "meshes":
        d = meshes.mesh_scene_random_cameras(321, 64, 64, "Rasterizer")
    else:
        # a camera inside a cube field scaled up: hundreds of large triangles per tile
        base = meshes.cube_field(6, 60, spread=3.0)
        d = scenes.SceneDesc(
            num_worlds=base.num_worlds, render_mode="Raytracer", width=64, height=64, asset_paths=base.asset_paths,
            materials=base.materials, texture_paths=base.texture_paths,
            instances=[(p_, q_, (s_[0] * 3.0, s_[1] * 3.0, s_[2] * 3.0), o_) for p_, q_, s_, o_ in base.instances],
            cameras=[((c[0][0] * 0.2, c[0][1] * 0.2, 1.0), c[1]) for c in base.cameras], worlds=base.worlds)
    r = make_product(d, visibility=True, variant=2)
    ref = render_oracle(d)
    assert_parity(fetch(r), ref)
    assert (ref["tri_id"] >= 0).mean() > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("classify", [0, 1])
def test_quaternions_that_are_not_rotations(native, monkeypatch, classify):
    # S1 uses quaternions as given: instances and cameras whose quaternion is not of unit length
    # shear and scale -- box tests, S6b and the leaf test still have to agree with the oracle
    # (both instantiations of the kernel)
    monkeypatch.setenv("MRX_BVH_CLASSIFY", str(classify))
    base = meshes.cube_field(12, 40)
    inst = []
    for i, (p_, q_, s_, o_) in enumerate(base.instances):
        f = (1.0, 1.3, 0.8, 1.00005, 0.6)[i % 5]
        inst.append((p_, tuple(float(np.float32(c * f)) for c in q_), s_, o_))
    cams = []
    for i, (e_, q_) in enumerate(base.cameras):
        f = (1.0, 0.9, 1.2)[i % 3]
        cams.append((e_, tuple(float(np.float32(c * f)) for c in q_)))
    d = scenes.SceneDesc(num_worlds=base.num_worlds, render_mode=base.render_mode, width=64, height=64,
                         asset_paths=base.asset_paths, materials=base.materials, texture_paths=base.texture_paths,
                         instances=inst, cameras=cams, worlds=base.worlds)
    r = make_product(d, visibility=True, variant=2)
    ref = render_oracle(d)
    assert_parity(fetch(r), ref)
    assert (ref["tri_id"] >= 0).mean() > 0.2
