"""Parity of the HIP path against the CPU oracle on the same seeded inputs
(-m gpu; calls go through the compiled module -> Manager -> C-ABI -> kernels).

Bar (BASELINE.json north_star): coverage / visibility bit-exact, colour and
depth within 1e-4.  Colour is in fact compared for equality of the RGBA8 bytes
and depth to rtol 1e-4 (it is bit-exact today; the tolerance is the contract).
"""
import ctypes
import os

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.util import assert_parity, fetch, make_product, render_oracle

pytestmark = pytest.mark.gpu

CUBE = os.path.join(scenes.DATA_DIR, "cube.obj")
PLANE = os.path.join(scenes.DATA_DIR, "plane.obj")
IDENT = (1.0, 0.0, 0.0, 0.0)


def _parity(desc, **kw):
    r = make_product(desc, visibility=True)
    got = fetch(r)
    ref = render_oracle(desc)
    assert_parity(got, ref, **kw)
    return r, got, ref


@pytest.mark.parametrize("mode", ["Rasterizer", "Raytracer"])
def test_reference_demo_scene(native, mode):
    # /root/reference/scripts/test.py: 4 worlds aliasing rows, cube + raw triangle
    _, got, _ = _parity(scenes.demo_scene(num_worlds=4, render_mode=mode))
    assert got["rgb"].shape == (4, 64, 64, 4)


@pytest.mark.parametrize("kw", [
    dict(num_worlds=64),
    dict(num_worlds=64, with_wall=True, textured=True),
    dict(num_worlds=16, width=128, height=128, with_wall=True),
    dict(num_worlds=8, width=96, height=40),            # ragged tiles
    dict(num_worlds=8, width=32, height=32),            # smaller than a tile
    dict(num_worlds=4, width=200, height=72, textured=True),
    dict(num_worlds=6, width=50, height=30),            # width % 4 != 0: scalar stores
    dict(num_worlds=3, width=33, height=33, textured=True, render_mode="Raytracer"),
    dict(num_worlds=130, width=64, height=64, with_wall=True),   # ragged last tile group
    dict(num_worlds=8, width=256, height=256, textured=True, render_mode="Raytracer"),
    dict(num_worlds=5, width=80, height=80, with_wall=True, render_mode="Raytracer"),
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_synthetic_scenes(native, kw):
    _parity(scenes.synthetic_scene(**kw))


@pytest.mark.parametrize("views,strips", [(4, 0), (4, 1), (4, 3), (4, 7), (2, 0), (2, 1), (2, 5)])
@pytest.mark.parametrize("kw", [
    dict(num_worlds=64),
    dict(num_worlds=150),                                # 38 groups of 4: four full rounds and a partial one
    dict(num_worlds=7),                                  # odd workgroup count: unpaired last group
    dict(num_worlds=9, textured=True),                   # four-wave variant, second classify pass
    dict(num_worlds=13, render_mode="Raytracer"),        # transposed storage, segmask
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_xcd_aware_split_forced_on_small_batches(native, monkeypatch, kw, views, strips):
    # the split that moves strips between the workgroups of a pair (odd -> even
    # XCD) normally starts at 1024 groups of 4 (or 2) one-tile views; force the
    # group size and the split so that small, ragged batches exercise the
    # workgroups that carry an extra view
    monkeypatch.setenv("MRX_GROUP_VIEWS", str(views))
    monkeypatch.setenv("MRX_XCD_SKEW", str(strips))
    monkeypatch.setenv("MRX_XCD_ROTATE", str(strips & 1))     # ... with and without the round rotation
    _parity(scenes.synthetic_scene(**kw))


@pytest.mark.parametrize("phase", [0, 1])
@pytest.mark.parametrize("views,strips", [(4, 3), (4, 7), (2, 1), (2, 5)])
@pytest.mark.parametrize("kw", [
    dict(num_worlds=150), dict(num_worlds=7), dict(num_worlds=13, render_mode="Raytracer"),
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_xcd_split_with_the_workgroups_of_a_pair_trading_places(native, monkeypatch, kw, views, strips, phase):
    # which XCD workgroup 0 lands on depends on the hardware queue; on an odd start the two
    # workgroups of every pair trade places (a kernel argument fed back from an earlier
    # launch).  Forced both ways here -- including the unpaired last workgroup of an odd
    # count -- and left to the feedback over a few launches: always the same image
    monkeypatch.setenv("MRX_GROUP_VIEWS", str(views))
    monkeypatch.setenv("MRX_XCD_SKEW", str(strips))
    monkeypatch.setenv("MRX_XCD_PHASE", str(phase))
    d = scenes.synthetic_scene(**kw)
    r, got, ref = _parity(d)
    monkeypatch.delenv("MRX_XCD_PHASE")
    r2 = make_product(d, visibility=True)
    for _ in range(3):
        r2.step()
    assert_parity(fetch(r2), ref)


def test_2048_worlds_use_the_split_with_two_views_per_group(native):
    desc = scenes.synthetic_scene(2048)
    r = make_product(desc, visibility=False)
    got = fetch(r, visibility=False)
    ref = render_oracle(desc, want_ids=False)
    assert_parity(got, ref)


@pytest.mark.parametrize("env", [
    {"MRX_GROUP_TILES": "1"}, {"MRX_GROUP_TILES": "3"}, {"MRX_GROUP_TILES": "16"},
    {"MRX_GROUP_VIEWS": "1"}, {"MRX_GROUP_VIEWS": "2"}, {"MRX_GROUP_VIEWS": "4"},
], ids=lambda e: "-".join(f"{k[10:]}{v}" for k, v in e.items()))
@pytest.mark.parametrize("kw", [
    dict(num_worlds=9, width=128, height=128, with_wall=True),          # 4 tiles / view, 32 slots
    dict(num_worlds=5, width=200, height=72, textured=True),            # 8 ragged tiles / view
    dict(num_worlds=3, width=256, height=256, textured=True, render_mode="Raytracer"),
    dict(num_worlds=2, width=320, height=192),                          # 15 tiles / view
    dict(num_worlds=1, width=384, height=320),                          # 30 tiles: chunks of a view
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_workgroup_shapes(native, monkeypatch, kw, env):
    # whole views per workgroup (setup shared by the view's tiles) and chunks of
    # one large view, including shapes the automatic choice would not pick
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _parity(scenes.synthetic_scene(**kw))


@pytest.mark.parametrize("slots", [32, 64, 128, 256])
@pytest.mark.parametrize("kw", [
    dict(num_worlds=6, with_wall=True, textured=True),
    dict(num_worlds=3, width=200, height=136, with_wall=True),
    dict(num_worlds=2, width=320, height=256, textured=True, render_mode="Raytracer"),
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_more_triangle_slots_than_needed(native, monkeypatch, kw, slots):
    # the 128- and 256-slot instantiations (several setup waves, several 64-slot
    # sub-chunks per region) on scenes whose answer is known from the small ones
    monkeypatch.setenv("MRX_DEBUG_SLOTS", str(slots))
    _parity(scenes.synthetic_scene(**kw))


def test_write_back_store_policy_gives_the_same_bytes(native, monkeypatch):
    desc = scenes.synthetic_scene(num_worlds=40, with_wall=True, textured=True)
    _, a, _ = _parity(desc)
    monkeypatch.setenv("MRX_WRITE_THROUGH", "0")
    _, b, _ = _parity(desc)
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(a[k], b[k])


def test_output_placement_search_changes_no_byte_and_pointers_stay_put(native, monkeypatch):
    # mrx_create times candidate allocations of the output tensors and keeps the
    # fastest; whatever it picks, the tensors hold the same bytes, do not overlap
    # and do not move afterwards
    desc = scenes.synthetic_scene(num_worlds=300, textured=True, render_mode="Raytracer")
    monkeypatch.setenv("MRX_PLACEMENT_TRIES", "1")
    _, a, _ = _parity(desc)
    monkeypatch.setenv("MRX_PLACEMENT_TRIES", "5")
    monkeypatch.setenv("MRX_OUT_SKEW_DEPTH_KB", "0")
    r, b, _ = _parity(desc)
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(a[k], b[k])
    def ptrs():
        return [r.rgb_cuda_ptr(), r.depth_cuda_ptr(), r.visibility_tensor().to_torch().data_ptr()]
    before = ptrs()
    nbytes = 300 * 64 * 64 * 4
    spans = sorted((p, p + nbytes) for p in before)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(2))
    for _ in range(3):
        r.step()
    r.sync()
    assert before == ptrs()
    assert r.rgb_tensor().to_torch().data_ptr() == before[0]


def test_outputs_backed_by_the_virtual_memory_api(native, monkeypatch):
    # MRX_OUT_ALLOC=vmm (a placement diagnostic): one physical handle mapped at a reserved
    # address; same bytes, and the mapping is torn down with the renderer
    monkeypatch.setenv("MRX_OUT_ALLOC", "vmm")
    desc = scenes.synthetic_scene(300, with_wall=True)
    ref = render_oracle(desc)
    for _ in range(3):
        r = make_product(desc, visibility=True)
        assert_parity(fetch(r), ref)
        del r


def test_headline_config_full_size(native):
    # BASELINE north star: 4096 worlds x 64x64 -- every pixel of every view
    desc = scenes.synthetic_scene(4096)
    r = make_product(desc, visibility=True)
    got = fetch(r)
    ref = render_oracle(desc)
    assert_parity(got, ref)
    assert (ref["tri_id"] >= 0).mean() > 0.6
    # without the visibility buffer (the benchmarked configuration) the
    # RGB / depth bytes are the same
    r2 = make_product(desc, visibility=False)
    got2 = fetch(r2, visibility=False)
    assert np.array_equal(got2["rgb"], got["rgb"])
    assert np.array_equal(got2["depth"], got["depth"])


def test_config_c3_4096_worlds_128(native):
    desc = scenes.synthetic_scene(4096, width=128, height=128, with_wall=True)
    r = make_product(desc, visibility=False)
    got = fetch(r, visibility=False)
    ref = render_oracle(desc, want_ids=False)
    assert_parity(got, ref)


def test_config_c2_exactly_1024_worlds(native):
    # BASELINE configs[1]: its own launch shape (two views per workgroup, 512
    # workgroups, no XCD split)
    desc = scenes.synthetic_scene(1024)
    r = make_product(desc, visibility=True)
    assert_parity(fetch(r), render_oracle(desc))


def test_config_c4_last_shard_of_16384_worlds(native):
    # BASELINE configs[3]: the rows rank 7 of 8 owns (worlds 14336 .. 16383)
    desc = scenes.synthetic_scene(2048, first_world=7 * 2048)
    whole_first = scenes.synthetic_scene(4, first_world=0)
    assert desc.cameras[:4] != whole_first.cameras          # really other worlds
    r = make_product(desc, visibility=False)
    assert_parity(fetch(r, visibility=False), render_oracle(desc, want_ids=False))


@pytest.mark.parametrize("variant", [0, 3], ids=["default-dispatch-bvh-path", "raster-kernels"])
def test_config_c5_4096_worlds_256_textured_raytracer(native, monkeypatch, variant):
    # BASELINE configs[4] at its full size: every pixel of every view, colour,
    # depth and segmask (3 GiB of output on the card) -- through the BVH ray-trace path
    # the config names (mgr.cpp:443-492), which the default dispatch picks for this
    # batch since round 4 (the flat kernel, raster.hpp bvhDispatchFlat), and through
    # the tiled raster kernel (kernel_variant 3)
    if variant:
        monkeypatch.setenv("MADRONA_MI355_KERNEL", str(variant))
    desc = scenes.synthetic_scene(4096, width=256, height=256, textured=True,
                                  render_mode="Raytracer")
    r = make_product(desc, visibility=False)
    assert r.render_path() == ("raster" if variant else "bvh")
    got = fetch(r, visibility=False, raytracer=True)
    assert got["rgb"].shape == (4096, 256, 256, 4) and got["segmask"].dtype == np.int32
    del r
    ref = render_oracle(desc)
    assert_parity(got, ref)
    assert (ref["segmask"] >= 0).mean() > 0.6


@pytest.mark.parametrize("path", ["raster", "bvh"])
def test_step_can_be_captured_in_a_hip_graph_with_the_pose_update(native, path):
    # an RL loop that is launch-bound captures "update the poses, render" once and replays it: step() is one
    # kernel launch on the renderer's stream with nothing that cannot be captured (the first frame, with its
    # one-time function attributes, was rendered by the constructor).  Three replays of (z += 0.5; step).
    import torch
    from tests import meshes as tmeshes
    d = scenes.synthetic_scene(48) if path == "raster" else tmeshes.cube_field(num_worlds=20, cubes=40)
    r = make_product(d)
    assert r.render_path() == path
    side = torch.cuda.Stream()
    r.set_stream(side.cuda_stream)
    pos = r.instance_position_tensor().to_torch()
    row = 1                                          # (the first cube of world 0 in both scenes)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        r.step()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            pos[row, 2] += 0.5
            r.step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    got = fetch(r)
    inst = list(d.instances)
    z = np.float32(inst[row][0][2])
    for _ in range(3):
        z = np.float32(z + np.float32(0.5))
    inst[row] = ((inst[row][0][0], inst[row][0][1], float(z)),) + tuple(inst[row][1:])
    d.instances = inst
    assert_parity(got, render_oracle(d))


@pytest.mark.parametrize("own_stream", [False, True], ids=["null-stream", "torch-side-stream"])
def test_pose_tensors_are_live_and_stepping_rerenders(native, own_stream):
    # scripts/test.py:137-151: mutate instance_position_tensor in place, step --
    # with NO host synchronisation between the write and step(): both are
    # ordered on one stream (the null stream, or a torch side stream handed to
    # the renderer with set_stream)
    import torch
    desc = scenes.demo_scene(num_worlds=4, render_mode="Rasterizer")
    r = make_product(desc)
    if own_stream:
        side = torch.cuda.Stream()
        r.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            _pose_steps_and_check(r, busy=True)
        return
    _pose_steps_and_check(r)


def _pose_steps_and_check(r, busy=False):
    import torch
    if busy:
        # ~10 ms of work queued on the side stream ahead of the pose writes: a
        # renderer still launching on the null stream would render stale poses
        x = torch.randn(4096, 4096, device="cuda")
        for _ in range(40):
            x = (x @ x) * 1e-4
    pos = r.instance_position_tensor().to_torch()
    rot = r.instance_rotation_tensor().to_torch()
    cpos = r.camera_position_tensor().to_torch()
    crot = r.camera_rotation_tensor().to_torch()
    assert pos.is_cuda and tuple(pos.shape) == (8, 3) and tuple(rot.shape) == (8, 4)
    assert tuple(cpos.shape) == (4, 3) and tuple(crot.shape) == (4, 4)
    before = fetch(r)
    for s in range(3):
        pos[0][2] += 1.0
        pos[2][2] += 2.0
        pos[4][2] += 1.5
        pos[6][2] += 0.5
        cpos[1][0] -= 0.75
        r.step()
    got = fetch(r)
    assert not np.array_equal(got["rgb"], before["rgb"])
    # oracle on the same mutated per-world rows (worlds own copies of the rows)
    d2 = scenes.demo_scene(num_worlds=4, render_mode="Rasterizer")
    inst = []
    for w, dz in enumerate((1.0, 2.0, 1.5, 0.5)):
        (p, q, s_, o), tri = d2.instances[0], d2.instances[1]
        z = np.float32(p[2])
        for _ in range(3):
            z = np.float32(z + np.float32(dz))
        inst += [((p[0], p[1], float(z)), q, s_, o), tri]
    d2.instances = inst
    cams = [d2.cameras[0]] * 4
    x = np.float32(cams[1][0][0])
    for _ in range(3):
        x = np.float32(x - np.float32(0.75))
    cams[1] = ((float(x),) + tuple(cams[1][0][1:]), cams[1][1])
    d2.cameras = cams
    d2.worlds = [(2, 2 * w, 1, w) for w in range(4)]
    assert_parity(got, render_oracle(d2))


def test_rendering_is_deterministic(native):
    desc = scenes.synthetic_scene(256, with_wall=True, textured=True)
    r = make_product(desc)
    a = fetch(r)
    r.step()
    r.render()
    b = fetch(r)
    for k in a:
        assert np.array_equal(a[k], b[k])


def _world(instances, assets, cams, **kw):
    return scenes.SceneDesc(num_worlds=1, asset_paths=assets, instances=instances,
                            cameras=cams, worlds=[(len(instances), 0, len(cams), 0)], **kw)


def test_more_than_64_triangles_per_world_uses_chunks(native):
    # 9 cubes + plane = 110 triangles: two setup chunks per band
    rng = np.random.default_rng(3)
    inst = [((0.0, 0.0, 0.0), IDENT, (1.0, 1.0, 1.0), 1)]
    for _ in range(9):
        p = rng.uniform(-5, 5, 3)
        inst.append(((float(p[0]), float(p[1]), float(abs(p[2]))), IDENT,
                     (1.5, 1.0, 2.0), 0))
    cams = [((12.0, -9.0, 7.0), scenes.look_at((12.0, -9.0, 7.0), (0, 0, 1))),
            ((-3.0, 14.0, 3.0), scenes.look_at((-3.0, 14.0, 3.0), (0, 0, 1)))]
    d = _world(inst, [(CUBE, 0), (PLANE, 1)], cams,
               materials=[((0.9, 0.4, 0.2, 1.0), 0, 0.5, 0.5), ((0.3, 0.6, 0.3, 1.0), -1, 0.5, 0.5)],
               texture_paths=[os.path.join(scenes.DATA_DIR, "cube.png")], width=128, height=64)
    _, got, _ = _parity(d)
    assert got["rgb"].shape == (2, 64, 128, 4)
    assert got["tri_id"].max() >= 64


@pytest.mark.parametrize("cubes", [21, 25, 44, 70])
def test_several_passes_of_256_triangles(native, cubes):
    # 254 / 302 / 530 / 842 triangles: a pass that is exactly short of full, a
    # second pass with one sub-chunk, three and four passes; textured cubes so
    # that winners of an early pass are shaded before their records are replaced
    rng = np.random.default_rng(cubes)
    inst = [((0.0, 0.0, 0.0), IDENT, (1.0, 1.0, 1.0), 1)]
    for _ in range(cubes):
        p = rng.uniform(-7, 7, 3)
        s = float(rng.uniform(0.4, 1.3))
        inst.append(((float(p[0]), float(p[1]), float(abs(p[2]) * 0.3)), IDENT, (s, s, s), 0))
    cams = [((13.0, -10.0, 8.0), scenes.look_at((13.0, -10.0, 8.0), (0, 0, 0.5))),
            ((-2.0, 15.0, 2.5), scenes.look_at((-2.0, 15.0, 2.5), (0, 0, 0.5)))]
    d = _world(inst, [(CUBE, 0), (PLANE, 1)], cams,
               materials=[((0.9, 0.7, 0.5, 1.0), 0, 0.5, 0.5), ((0.3, 0.6, 0.3, 1.0), -1, 0.5, 0.5)],
               texture_paths=[os.path.join(scenes.DATA_DIR, "cube.png")], width=96, height=72)
    _, got, _ = _parity(d)                                    # default dispatch: the BVH path
    assert got["tri_id"].max() >= min(256, 10 * cubes)        # later passes own pixels too
    # the tiled raster kernels, forced (group kernel up to 256 triangles, chunked above)
    r3 = make_product(d, visibility=True, variant=3)
    got3 = fetch(r3)
    for k in ("rgb", "depth", "tri_id"):
        assert np.array_equal(got[k], got3[k])


def test_edge_cases_empty_worlds_bad_ids_degenerate_triangles(native):
    verts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0],        # collinear: zero area
                      [0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    d = scenes.SceneDesc(
        num_worlds=4, width=64, height=64,
        asset_paths=[(CUBE, -1)],
        mesh_vertices=verts, mesh_uvs=np.zeros((6, 2), np.float32),
        mesh_indices=np.array([0, 1, 2, 0, 1, 2], np.uint32),
        mesh_vertex_offsets=np.array([0, 3], np.uint32),
        mesh_indices_offsets=np.array([0, 3], np.uint32),
        mesh_materials=np.array([-1, 7], np.int32),            # 7: no such material
        instances=[((0.0, 5.0, 0.0), IDENT, (1.0, 1.0, 1.0), 0),
                   ((0.0, 4.0, 0.0), IDENT, (1.0, 1.0, 1.0), 1),   # degenerate
                   ((0.0, 3.0, -0.5), IDENT, (1.0, 1.0, 1.0), 2),
                   ((0.0, 2.0, 0.0), IDENT, (1.0, 1.0, 1.0), 9),   # no such object
                   ((0.0, 2.0, 0.0), IDENT, (1.0, 1.0, 1.0), -1),
                   ((0.0, 6.0, 0.0), IDENT, (0.0, 0.0, 0.0), 0)],  # zero scale
        cameras=[((0.0, 0.0, 0.0), IDENT), ((0.0, 10.0, 0.0), (0.0, 0.0, 0.0, 1.0))],
        worlds=[(6, 0, 1, 0),      # everything
                (0, 0, 1, 1),      # empty world: background only
                (2, 3, 2, 0),      # only invalid ids, two cameras
                (1, 1, 0, 0)])     # no camera at all
    r, got, ref = _parity(d)
    assert got["rgb"].shape[0] == 4            # 1 + 1 + 2 + 0 views
    assert (got["tri_id"][1] == -1).all() and (got["depth"][1] == 0).all()
    assert (got["tri_id"][2] == -1).all() and (got["tri_id"][3] == -1).all()
    assert (got["tri_id"][0] >= 0).any()


def test_back_face_culling_cases(native):
    # S6b: eye outside / inside a closed mesh, mirrored instance, open mesh
    # with the same winding, inward-wound closed mesh given as raw geometry
    from oracle import oracle
    cube, cuv = oracle.parse_obj(CUBE)
    inward = np.ascontiguousarray(cube[:, ::-1]).reshape(-1, 3)
    d = scenes.SceneDesc(
        num_worlds=5, width=64, height=64, asset_paths=[(CUBE, -1), (PLANE, -1)],
        mesh_vertices=inward, mesh_uvs=np.zeros((len(inward), 2), np.float32),
        mesh_indices=np.arange(len(inward), dtype=np.uint32),
        mesh_vertex_offsets=np.array([0], np.uint32), mesh_indices_offsets=np.array([0], np.uint32),
        mesh_materials=np.array([-1], np.int32),
        instances=[((0.0, 6.0, 0.0), IDENT, (2.0, 2.0, 2.0), 0),
                   ((0.5, 6.0, 0.3), (0.9238795, 0.0, 0.3826834, 0.0), (-2.0, 1.5, 2.5), 0),
                   ((0.0, 6.0, -1.0), IDENT, (0.001, 0.001, 1.0), 1),
                   ((0.0, 6.0, 0.0), IDENT, (2.0, 2.0, 2.0), 2)],
        cameras=[((0.0, 0.0, 0.0), IDENT), ((0.0, 6.0, 0.0), IDENT),
                 ((0.3, 5.2, 0.1), (0.9659258, 0.0, 0.0, 0.2588190))],
        worlds=[(1, 0, 1, 0), (1, 0, 2, 1), (1, 1, 1, 0), (2, 2, 1, 0), (1, 3, 3, 0)])
    _, got, ref = _parity(d)
    assert got["rgb"].shape[0] == 8
    assert (got["tri_id"][1] >= 0).all()        # eye inside the cube: walls everywhere


def test_back_face_culling_is_per_shell_and_respects_the_near_plane(native):
    # an object of two shells wound oppositely (each judged on its
    # own), and an eye outside a cube's box but within the near plane's reach
    # of its front face (the far wall must show); the product must match the
    # oracle AND the oracle must match its own render with S6b switched off
    from oracle import oracle
    cube, _ = oracle.parse_obj(CUBE)
    shifted = cube[:, ::-1] + np.array([2.5, 0.0, 0.4], np.float32)
    verts = np.concatenate([cube, shifted]).reshape(-1, 3)
    for mode in ("Rasterizer", "Raytracer"):
        d = scenes.SceneDesc(
            num_worlds=2, width=64, height=64, render_mode=mode, asset_paths=[(CUBE, -1)],
            mesh_vertices=verts, mesh_uvs=np.zeros((len(verts), 2), np.float32),
            mesh_indices=np.arange(len(verts), dtype=np.uint32),
            mesh_vertex_offsets=np.array([0], np.uint32),
            mesh_indices_offsets=np.array([0], np.uint32),
            mesh_materials=np.array([-1], np.int32),
            instances=[((-1.0, 6.0, 0.0), (0.9238795, 0.0, 0.0, 0.3826834), (1.5, 1.5, 1.5), 1),
                       ((0.0, 1.05, 0.0), IDENT, (2.0, 2.0, 2.0), 0)],
            cameras=[((0.0, 0.0, 0.5), IDENT), ((0.0, 0.0, 0.0), IDENT)],
            worlds=[(1, 0, 1, 0), (1, 1, 1, 1)])
        _, got, ref = _parity(d)
        fs = oracle.FlatScene(d)
        fs.tri_orient[:] = 0.0
        plain = fs.render()
        assert np.array_equal(plain["rgb"], got["rgb"])
        assert (got["tri_id"][0] >= 12).any()          # the inward-wound shell is visible
        if mode == "Raytracer":
            assert abs(got["depth"][1, 32, 32] - 2.05) < 1e-4


def test_obj_materials_from_mtl(native, tmp_path):
    # mat_id -1: faces keep the materials their OBJ names (mtllib / usemtl):
    # Kd colour, map_Kd PNG texture; unknown names, a missing library and an
    # unreadable texture all fall back to the defaults
    import shutil
    shutil.copy(os.path.join(scenes.DATA_DIR, "cube.png"), tmp_path / "tex.png")
    (tmp_path / "two.mtl").write_text(
        "newmtl red\nKd 0.9 0.2 0.1\n"
        "newmtl skin\nKd 0.8 0.8 1.0\nmap_Kd tex.png\n"
        "newmtl broken\nKd 0.3 0.9 0.3\nmap_Kd missing.png\n")
    (tmp_path / "quad.obj").write_text(
        "mtllib two.mtl\nmtllib nowhere.mtl\n"
        "v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 3 0 -1\nv 3 0 1\nv 5 0 -1\nv 5 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "f 1/1 2/2 3/3\n"                       # no usemtl yet: default material
        "usemtl skin\nf 1/1 3/3 4/4\n"
        "usemtl red\nf 2/1 5/2 6/3\n"
        "usemtl ghost\nf 2/1 6/3 3/4\n"        # not in any library
        "usemtl broken\nf 5/1 7/2 8/3\n"
        "usemtl skin\nf 5/1 8/3 6/4\n")
    cams = [((2.0, -6.0, 0.0), IDENT), ((2.0, 6.0, 0.5), (0.0, 0.0, 0.0, 1.0))]
    d = scenes.SceneDesc(
        num_worlds=1, width=96, height=64,
        asset_paths=[(str(tmp_path / "quad.obj"), -1), (CUBE, -1), (str(tmp_path / "quad.obj"), 0)],
        materials=[((0.2, 0.4, 0.9, 1.0), -1, 0.5, 0.5)],
        instances=[((0.0, 0.0, 0.0), IDENT, (1.0, 1.0, 1.0), 0),
                   ((0.0, -2.0, 1.5), IDENT, (1.0, 1.0, 1.0), 1),
                   ((0.0, 0.0, -2.2), IDENT, (1.0, 1.0, 1.0), 2)],
        cameras=cams, worlds=[(3, 0, 2, 0)])
    r, got, ref = _parity(d)
    from oracle import oracle
    fs = oracle.FlatScene(d)
    # API material 0, then quad.obj's skin / red / broken in order of first use
    # ("ghost" is in no library), then cube.obj's cube
    assert fs.mat_tex.tolist() == [-1, 0, -1, -1, 1]
    assert fs.tri_mat[:6].tolist() == [-1, 1, 2, -1, 3, 1]
    assert set(fs.tri_mat[6:18].tolist()) == {4} and set(fs.tri_mat[18:].tolist()) == {0}
    colours = {tuple(c) for c in got["rgb"][0].reshape(-1, 4).tolist()}
    assert len(colours) > 10                      # textured faces show many texel colours


@pytest.mark.parametrize("split", [False, True])
def test_multi_block_obj_is_one_object_unless_split(native, tmp_path, monkeypatch, split):
    # An OBJ file is one object whatever `o` / `g` blocks it holds (importFromDisk(..., true),
    # objects[i] <-> asset i: mgr.cpp:301-303,340-345): the next asset is object 1.  With
    # MRX_OBJ_SPLIT_BLOCKS=1 the three blocks are objects 0..2 and the next asset is object 3.
    from tests.test_host_logic import MULTI_OBJ
    path = tmp_path / "multi.obj"
    path.write_text(MULTI_OBJ)
    if split:
        monkeypatch.setenv("MRX_OBJ_SPLIT_BLOCKS", "1")
    else:
        monkeypatch.delenv("MRX_OBJ_SPLIT_BLOCKS", raising=False)
    q = (0.7071068, 0.7071068, 0.0, 0.0)
    cube_id = 3 if split else 1
    for mode in ("Rasterizer", "Raytracer"):
        d = scenes.SceneDesc(
            num_worlds=2, width=64, height=64, render_mode=mode,
            asset_paths=[(str(path), 0), (CUBE, 1)],
            materials=[((0.9, 0.2, 0.2, 1.0), -1, 0.5, 0.5), ((0.2, 0.9, 0.2, 1.0), -1, 0.5, 0.5)],
            instances=[((-2.0, 6.0, -1.0), IDENT, (1.0, 1.0, 1.0), 0), ((-1.0, 6.0, -1.0), IDENT, (1.0, 1.0, 1.0), 1),
                       ((-1.5, 6.5, 0.5), IDENT, (1.0, 1.0, 1.0), 2), ((2.0, 7.0, 0.0), q, (1.5, 1.5, 1.5), 3),
                       ((0.0, 5.0, 1.0), IDENT, (1.0, 1.0, 1.0), 4)],       # 4: no such object
            cameras=[((0.0, 0.0, 0.0), IDENT)], worlds=[(5, 0, 1, 0), (2, 1, 1, 0)])
        r = make_product(d, visibility=False)
        got = fetch(r, visibility=False, raytracer=(mode == "Raytracer"))
        ref = render_oracle(d)
        assert_parity(got, ref)
        from oracle import oracle
        fs = oracle.FlatScene(d)
        assert fs.obj_num_tris.tolist() == ([2, 3, 2, 12] if split else [7, 12])
        assert int(fs.obj_num_tris[cube_id]) == 12
        if mode == "Raytracer":
            # (objects that do not exist draw nothing: ids 2.. unsplit, id 4 split)
            assert set(np.unique(ref["segmask"][0]).tolist()) == ({-1, 0, 1, 2, 3} if split else {-1, 0, 1})


@pytest.mark.parametrize("kind", ["uniform", "ragged", "bvh"])
def test_instances_hidden_and_shown_between_steps(native, kind):
    # f4 (/root/reference/src/sim.inl:5-16): a negative ObjectID hides the instance from
    # the next step on, the id written back shows it again; triangle slots (and
    # with them the visibility ids of everything else) stay where they are
    import torch
    from tests import meshes
    from oracle import oracle
    if kind == "uniform":
        d = scenes.synthetic_scene(70, with_wall=True, textured=True)       # the arithmetic draw list
    elif kind == "ragged":
        d = scenes.synthetic_scene(12, with_wall=True)
        d.worlds = [(3, 3 * w, 1, w) if w % 3 else (2, 3 * w, 1, w) for w in range(12)]
    else:
        d = meshes.cube_field(num_worlds=5, cubes=30, textured=True)        # 362 triangles: BVH path
    r = make_product(d, visibility=True)
    fs = oracle.FlatScene(d)
    obj = r.instance_object_tensor().to_torch()
    assert obj.dtype == torch.int32 and tuple(obj.shape) == (len(fs.inst_obj),) and obj.is_cuda
    assert np.array_equal(obj.cpu().numpy(), fs.inst_obj)
    rng = np.random.default_rng(5)
    first = None
    for step in range(4):
        hide = rng.random(len(fs.inst_obj)) < (0.5 if step < 3 else 0.0)    # last step: all back
        now = np.where(hide, -1 - fs.inst_obj0, fs.inst_obj0).astype(np.int32)
        obj.copy_(torch.from_numpy(now).to(obj.device))
        r.step()
        fs.inst_obj[:] = now
        got, ref = fetch(r), fs.render()
        assert_parity(got, ref)
        first = ref if first is None else first
        assert step == 3 or not np.array_equal(ref["tri_id"], full_ids(fs))
    assert np.array_equal(got["tri_id"], full_ids(fs))


@pytest.mark.parametrize("kind", ["raster", "bvh"])
def test_a_different_nonnegative_object_id_changes_neither_geometry_nor_segmask(native, kind):
    # the geometry of an instance is bound at creation; only the sign of the live
    # ObjectID is interpreted.  The segmask therefore shows the id of the BOUND object, so
    # that labels and geometry agree whatever non-negative value the column holds.
    import torch
    from tests import meshes
    if kind == "raster":
        d = scenes.synthetic_scene(6, with_wall=True, textured=True, render_mode="Raytracer")
    else:
        d = meshes.cube_field(num_worlds=3, cubes=30, mode="Raytracer")
    r = make_product(d, visibility=False)
    before = fetch(r, visibility=False, raytracer=True)
    ref = render_oracle(d)
    assert_parity(before, ref)
    obj = r.instance_object_tensor().to_torch()
    bound = obj.clone()
    obj.copy_((bound + 1) % 3)                    # other valid ids, all non-negative
    r.step()
    after = fetch(r, visibility=False, raytracer=True)
    for k in ("rgb", "depth", "segmask"):
        assert np.array_equal(before[k], after[k]), k
    assert set(np.unique(after["segmask"]).tolist()) <= set(bound.cpu().tolist()) | {-1}


@pytest.mark.parametrize("worlds,kw", [(40, {}), (1100, {}), (300, {"textured": True}), (90, {"render_mode": "Raytracer", "textured": True})])
def test_preloaded_header_entry_point_renders_the_same_bytes(native, monkeypatch, worlds, kw):
    # The group kernel's fast entry point (argument header preloaded into SGPRs, pose / geometry addresses derived
    # from two block pointers) against the plain one (MRX_GROUP_FAST=0, read per launch) on uniform 64x64 worlds:
    # small and chip-filling batches (the XCD-aware split and its trade instantiations), textured, Raytracer, with ids
    d = scenes.synthetic_scene(worlds, **kw)
    rt = kw.get("render_mode") == "Raytracer"
    r = make_product(d, visibility=not rt)
    ref = render_oracle(d)
    fast = fetch(r, visibility=not rt, raytracer=rt)
    assert_parity(fast, ref)
    monkeypatch.setenv("MRX_GROUP_FAST", "0")
    for _ in range(3):
        r.step()
    plain = fetch(r, visibility=not rt, raytracer=rt)
    for k in fast:
        assert np.array_equal(fast[k], plain[k]), k
    monkeypatch.delenv("MRX_GROUP_FAST")
    # a multi-camera uniform world (view / cameras-per-world division in the header path)
    d2 = scenes.synthetic_scene(12)
    cams = list(d2.cameras)
    d2.cameras = [c for pq in cams for c in (pq, ((pq[0][0], pq[0][1], pq[0][2] + 1.5), pq[1]))]
    d2.worlds = [(2, 2 * w, 2, 2 * w) for w in range(12)]
    r2 = make_product(d2)
    assert_parity(fetch(r2), render_oracle(d2))


def full_ids(fs):
    keep = fs.inst_obj.copy()
    fs.inst_obj[:] = fs.inst_obj0
    out = fs.render()["tri_id"]
    fs.inst_obj[:] = keep
    return out


def test_zero_worlds_and_zero_views(native):
    d = scenes.synthetic_scene(2)
    d.worlds = []
    d.num_worlds = 0
    r = make_product(d)
    r.step()
    r.sync()
    assert tuple(r.rgb_tensor().shape) == (0, 64, 64, 4)
    d = scenes.synthetic_scene(2)
    d.worlds = [(2, 0, 0, 0), (2, 2, 0, 0)]       # instances but no cameras
    r = make_product(d)
    r.step()
    r.sync()
    assert tuple(r.depth_tensor().shape) == (0, 64, 64, 1)
    assert tuple(r.instance_position_tensor().shape) == (4, 3)


def test_error_behaviour(native):
    m = native.load_module()
    r = make_product(scenes.demo_scene(num_worlds=1, render_mode="Rasterizer"), visibility=False)
    # /root/reference/src/mgr.cpp:594-596: FATAL("Segmask not implemented for rasterizer")
    with pytest.raises(RuntimeError, match="Segmask not implemented for rasterizer"):
        r.segmask_tensor()
    with pytest.raises(RuntimeError):
        r.visibility_tensor()
    bad = scenes.demo_scene(num_worlds=1)
    bad.asset_paths = [("/nonexistent/file.obj", 0)]
    with pytest.raises(RuntimeError, match="Failed to load render assets"):
        scenes.make_renderer(bad)
    bad = scenes.demo_scene(num_worlds=1)
    bad.worlds = [(5, 0, 1, 0)]
    with pytest.raises(RuntimeError, match="outside the tables"):
        scenes.make_renderer(bad)
    with pytest.raises(RuntimeError, match="gpu_id"):
        scenes.make_renderer(scenes.demo_scene(num_worlds=1), gpu_id=99)
    rt = make_product(scenes.demo_scene(num_worlds=1, render_mode="Raytracer"), visibility=False)
    seg = rt.segmask_tensor().to_torch()
    assert tuple(seg.shape) == (1, 64, 64) and str(seg.dtype) == "torch.int32"
    assert rt.rgb_cuda_ptr() == rt.rgb_tensor().device_ptr() != 0
    assert tuple(rt.depth_tensor().to_torch().shape) == (1, 64, 64)


def test_raw_c_abi_through_ctypes(native):
    """The boundary itself, without the C++/pybind layers."""
    lib = native.load_capi()

    class Geo(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in
                    ("vertices", "uvs", "indices", "mvo", "mio", "mm")] + \
                   [(n, ctypes.c_uint32) for n in ("nv", "ni", "nm")]

    class Cfg(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_uint32), ("gpu_id", ctypes.c_int32),
                    ("num_worlds", ctypes.c_uint32), ("render_mode", ctypes.c_int32),
                    ("view_width", ctypes.c_uint32), ("view_height", ctypes.c_uint32),
                    ("geo", Geo),
                    ("asset_paths", ctypes.POINTER(ctypes.c_char_p)), ("num_asset_paths", ctypes.c_uint32),
                    ("mat_assignments", ctypes.POINTER(ctypes.c_int32)), ("num_mat_assignments", ctypes.c_uint32),
                    ("materials", ctypes.c_void_p), ("num_materials", ctypes.c_uint32),
                    ("texture_paths", ctypes.POINTER(ctypes.c_char_p)), ("num_textures", ctypes.c_uint32),
                    ("instances", ctypes.c_void_p), ("num_instances", ctypes.c_uint32),
                    ("cameras", ctypes.c_void_p), ("num_cameras", ctypes.c_uint32),
                    ("worlds", ctypes.c_void_p),
                    ("stream", ctypes.c_void_p), ("flags", ctypes.c_uint32),
                    ("kernel_variant", ctypes.c_int32)]

    desc = scenes.synthetic_scene(3)
    inst = np.zeros((len(desc.instances), 11), np.float32)
    for i, (p, q, s, o) in enumerate(desc.instances):
        inst[i, :3], inst[i, 3:7], inst[i, 7:10] = p, q, s
        inst[i, 10:11].view(np.int32)[0] = o
    cams = np.array([list(p) + list(q) for p, q in desc.cameras], np.float32)
    worlds = np.array(desc.worlds, np.uint32)
    mats = np.zeros((len(desc.materials), 7), np.float32)
    for i, (c, t, ro, me) in enumerate(desc.materials):
        mats[i, :4] = c
        mats[i, 4:5].view(np.int32)[0] = t
        mats[i, 5], mats[i, 6] = ro, me
    paths = (ctypes.c_char_p * 2)(*[p.encode() for p, _ in desc.asset_paths])
    assign = (ctypes.c_int32 * 2)(*[i for _, i in desc.asset_paths])
    tex = (ctypes.c_char_p * 1)(desc.texture_paths[0].encode())
    cfg = Cfg()
    cfg.struct_size = ctypes.sizeof(Cfg)
    cfg.gpu_id, cfg.num_worlds, cfg.render_mode = 0, 3, 0
    cfg.view_width = cfg.view_height = 64
    cfg.asset_paths, cfg.num_asset_paths = paths, 2
    cfg.mat_assignments, cfg.num_mat_assignments = assign, 2
    cfg.materials, cfg.num_materials = mats.ctypes.data, len(mats)
    cfg.texture_paths, cfg.num_textures = tex, 1
    cfg.instances, cfg.num_instances = inst.ctypes.data, len(inst)
    cfg.cameras, cfg.num_cameras = cams.ctypes.data, len(cams)
    cfg.worlds = worlds.ctypes.data
    cfg.flags = 1
    h = ctypes.c_void_p()
    lib.mrx_last_error.restype = ctypes.c_char_p
    lib.mrx_buffer.restype = ctypes.c_void_p
    assert lib.mrx_create(ctypes.byref(cfg), ctypes.byref(h)) == 0, lib.mrx_last_error()
    bad = Cfg.from_buffer_copy(cfg)
    bad.struct_size = 8
    h2 = ctypes.c_void_p()
    assert lib.mrx_create(ctypes.byref(bad), ctypes.byref(h2)) == -1
    assert b"size mismatch" in lib.mrx_last_error()
    assert lib.mrx_step(h) == 0 and lib.mrx_render(h) == 0 and lib.mrx_sync(h) == 0
    dims = (ctypes.c_int64 * 4)()
    nd, dt, dev = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    p = lib.mrx_buffer(h, 0, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev))
    assert p and list(dims) == [3, 64, 64, 4] and nd.value == 4 and dt.value == 0
    assert lib.mrx_buffer(h, 2, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev)) is None
    assert lib.mrx_buffer(h, 99, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev)) is None
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    host = np.empty((3, 64, 64, 4), np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(p),
                         ctypes.c_size_t(host.nbytes), 2) == 0
    ref = render_oracle(desc)
    assert np.array_equal(host, ref["rgb"])
    # loader cross-check: what was uploaded equals the independent parse
    from oracle import oracle
    fs = oracle.FlatScene(desc)
    n = len(fs.tri_mat)
    tp = np.empty((n, 9), np.float32); tu = np.empty((n, 6), np.float32)
    tm = np.empty(n, np.int32); of = np.empty(2, np.int32); oc = np.empty(2, np.int32)
    assert lib.mrx_copy_triangles(h, tp.ctypes.data_as(ctypes.c_void_p), tu.ctypes.data_as(ctypes.c_void_p),
                                  tm.ctypes.data_as(ctypes.c_void_p), of.ctypes.data_as(ctypes.c_void_p),
                                  oc.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(tp.reshape(-1, 3, 3), fs.tri_pos) and np.array_equal(tu.reshape(-1, 3, 2), fs.tri_uv)
    assert np.array_equal(tm, fs.tri_mat) and np.array_equal(of, fs.obj_first_tri)
    # the same cross-check with file materials in play (cube.obj -> cube.mtl)
    d2 = scenes.synthetic_scene(2)
    d2.asset_paths = [(d2.asset_paths[0][0], -1), d2.asset_paths[1]]
    r2 = make_product(d2)
    fs2 = oracle.FlatScene(d2)
    tm2 = np.empty(len(fs2.tri_mat), np.int32)
    assert lib.mrx_copy_triangles(ctypes.c_void_p(r2.native_handle()), None, None,
                                  tm2.ctypes.data_as(ctypes.c_void_p), None, None) == 0
    assert np.array_equal(tm2, fs2.tri_mat) and tm2[0] == len(d2.materials)
    ms = ctypes.c_float()
    assert lib.mrx_time_renders(h, 5, ctypes.byref(ms)) == 0 and ms.value > 0
    lib.mrx_destroy(h)


@pytest.mark.parametrize("cus", [32, 128])
def test_launch_shapes_of_a_smaller_device_render_the_same_pixels(native, monkeypatch, cus):
    # "does the batch fill the chip" is decided from the device's CU count
    # (raster.hpp groupFill / bvhDispatchMinTris; mrx_create reads the attribute).  MRX_FAKE_CUS
    # makes the host choose the shapes a 32- or 128-CU device (a partitioned MI355X) would get:
    # other workgroup shapes, other dispatch thresholds, the same pixels.
    from tests import meshes as tmeshes
    monkeypatch.setenv("MRX_FAKE_CUS", str(cus))
    for d in (scenes.synthetic_scene(300), scenes.synthetic_scene(130, with_wall=True, textured=True),
              tmeshes.cube_field(num_worlds=100, cubes=6)):          # 74 triangles: BVH path up to 2.5 views per CU
        r = make_product(d, visibility=True)
        if d.instances and len(d.instances) // d.num_worlds == 7:
            assert r.render_path() == ("bvh" if 2 * 100 <= 5 * cus else "raster")
        assert_parity(fetch(r), render_oracle(d))
