"""Single-process multi-device (mrx_config.device_ids) and spare instance rows with re-binding
(max_instances_per_world + mrx_refresh_objects): VERDICT r2 items 5 and 8.

The GPU box has one device, so a renderer of several shards is rehearsed with the shards on the
same device (device_ids = [0, 0, ...]): every shard has its own tensors, its own world range and
its own launch -- what differs on a real node is only which device that launch runs on."""
import os

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests import meshes
from tests.util import assert_parity, fetch, make_product, render_oracle

pytestmark = pytest.mark.gpu


def _shard_outputs(r, n, raytracer=False, visibility=True):
    import torch
    out = {"rgb": [], "depth": [], "ids": []}
    r.sync()
    for i in range(n):
        out["rgb"].append(r.rgb_tensor(shard=i).to_torch().cpu().numpy())
        d = r.depth_tensor(shard=i).to_torch().cpu().numpy()
        out["depth"].append(d.reshape(d.shape[0], d.shape[1], d.shape[2]))
        t = r.visibility_tensor(shard=i) if visibility else r.segmask_tensor(shard=i)
        out["ids"].append(t.to_torch().cpu().numpy())
    return {k: np.concatenate(v) for k, v in out.items()}


@pytest.mark.parametrize("kind", ["raster-uniform", "raster-multicam", "bvh"])
@pytest.mark.parametrize("shards", [2, 3])
def test_one_renderer_of_several_shards_equals_the_single_shard_render(native, kind, shards):
    import torch
    if kind == "raster-uniform":
        d = scenes.synthetic_scene(11, with_wall=True, textured=True)
    elif kind == "raster-multicam":
        d = scenes.synthetic_scene(7)
        cams = list(d.cameras)
        d.cameras = cams + [((p[0], p[1], p[2] + 2.0), q) for p, q in cams]
        d.worlds = [(2, 2 * w, 2 if w % 2 else 1, w) for w in range(7)]       # 1 or 2 cameras: views != worlds
        d.worlds = [(ni, io, nc, co if nc == 1 else co) for ni, io, nc, co in d.worlds]
    else:
        d = meshes.cube_field(num_worlds=5, cubes=30, textured=True)          # 362 triangles: BVH path
    os.environ["MADRONA_MI355_VISIBILITY"] = "1"
    try:
        one = scenes.make_renderer(d)
        many = scenes.make_renderer(d, device_ids=[0] * shards)
    finally:
        os.environ.pop("MADRONA_MI355_VISIBILITY")
    assert many.num_shards == shards and one.num_shards == 1
    ref = _shard_outputs(one, 1)
    got = _shard_outputs(many, shards)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    # world ranges: contiguous, scenes.shard_range
    for i in range(shards):
        lo, hi = scenes.shard_range(d.num_worlds, i, shards)
        assert many.shard_first_world(i) == lo and many.shard_first_world(i + 1) == hi
        nv = sum(w[2] for w in d.worlds[lo:hi])
        assert tuple(many.rgb_tensor(shard=i).shape)[0] == nv
    # poses are per shard and live: move one instance of the last shard, step the whole renderer
    last = shards - 1
    pos = many.instance_position_tensor(shard=last).to_torch()
    pos1 = one.instance_position_tensor().to_torch()
    lo, _ = scenes.shard_range(d.num_worlds, last, shards)
    row0 = sum(w[0] for w in d.worlds[:lo])
    assert torch.equal(pos.cpu(), pos1[row0:].cpu())
    pos[1, 2] += 0.5
    pos1[row0 + 1, 2] += 0.5
    many.step()
    one.step()
    ref, got = _shard_outputs(one, 1), _shard_outputs(many, shards)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    # oracle parity of the whole
    ref_o = render_oracle(d)
    # (the oracle's scene has the unmoved pose: compare the views of the other shards only)
    nv0 = sum(w[2] for w in d.worlds[:lo])
    assert np.array_equal(got["ids"][:nv0], ref_o["tri_id"][:nv0])
    assert np.array_equal(got["rgb"][:nv0], ref_o["rgb"][:nv0])
    # a tensor of a multi-shard renderer needs its shard named
    with pytest.raises(ValueError, match="spans"):
        many.rgb_tensor()
    with pytest.raises(IndexError):
        many.rgb_tensor(shard=shards)
    assert many.bytes_per_step() == one.bytes_per_step()
    assert many.time_renders(3) > 0


def test_multi_shard_through_the_raw_c_abi_and_v2_config_still_accepted(native):
    import ctypes
    lib = native.load_capi()
    lib.mrx_num_shards.restype = ctypes.c_int
    lib.mrx_shard.restype = ctypes.c_void_p
    lib.mrx_shard.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mrx_buffer.restype = ctypes.c_void_p
    lib.mrx_buffer_shard.restype = ctypes.c_void_p
    lib.mrx_last_error.restype = ctypes.c_char_p
    d = scenes.synthetic_scene(5)
    r = scenes.make_renderer(d, device_ids=[0, 0])
    h = ctypes.c_void_p(r.native_handle())
    assert lib.mrx_num_shards(h) == 2
    dims = (ctypes.c_int64 * 4)()
    nd, dt, dev = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.mrx_buffer(h, 0, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev)) is None
    assert b"several devices" in lib.mrx_last_error()
    p0 = lib.mrx_buffer_shard(h, 0, 0, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev))
    assert p0 and list(dims) == [3, 64, 64, 4]
    sh1 = ctypes.c_void_p(lib.mrx_shard(h, 1))
    p1 = lib.mrx_buffer(sh1, 0, dims, ctypes.byref(nd), ctypes.byref(dt), ctypes.byref(dev))
    assert p1 and p1 != p0 and list(dims) == [2, 64, 64, 4]
    assert lib.mrx_shard(h, 2) is None


@pytest.mark.parametrize("kind", ["uniform", "ragged", "bvh"])
def test_cubes_spawned_into_a_running_renderer(native, kind):
    # f4 / VERDICT r2 item 8 (/root/reference/src/mgr.cpp:378-388 capacity, src/sim.inl:5-8
    # makeEntityRenderable): worlds own spare rows; writing an object id and a pose into one and
    # calling refresh_objects() makes it draw -- on the uniform raster path (which turns into the
    # draw-list path), on ragged worlds, and on the BVH path.
    import torch
    from oracle import oracle
    if kind == "uniform":
        d = scenes.synthetic_scene(20, with_wall=True, textured=True)
        d.max_instances_per_world = 5
    elif kind == "ragged":
        d = scenes.synthetic_scene(9, with_wall=True)
        d.worlds = [(3, 3 * w, 1, w) if w % 3 else (2, 3 * w, 1, w) for w in range(9)]
        d.max_instances_per_world = 6
    else:
        d = meshes.cube_field(num_worlds=4, cubes=30, textured=True)          # 362 triangles: BVH path
        d.max_instances_per_world = 34
    r = make_product(d, visibility=True)
    fs = oracle.FlatScene(d)
    assert_parity(fetch(r), fs.render())                                      # spare rows draw nothing
    obj = r.instance_object_tensor().to_torch()
    pos = r.instance_position_tensor().to_torch()
    scl = r.instance_scale_tensor().to_torch()
    assert tuple(obj.shape) == (len(fs.inst_obj),) and np.array_equal(obj.cpu().numpy(), fs.inst_obj)
    rng = np.random.default_rng(11)
    spare = np.flatnonzero(fs.inst_obj0 < 0)
    assert len(spare) >= 2 * d.num_worlds - 2
    # two cubes per world into the first two spare rows of each world
    rows = []
    for w in range(d.num_worlds):
        lo, hi = fs.world_inst_start[w], fs.world_inst_start[w + 1]
        rows += [int(x) for x in spare[(spare >= lo) & (spare < hi)][:2]]
    for row in rows:
        fs.inst_obj[row] = 0
        fs.inst_pos[row] = np.float32(rng.uniform(-3, 3, 3) + np.array([0, 0, 2.0]))
        fs.inst_scale[row] = np.float32(rng.uniform(0.5, 1.5))
    obj.copy_(torch.from_numpy(fs.inst_obj).to(obj.device))
    pos.copy_(torch.from_numpy(fs.inst_pos).to(pos.device))
    scl.copy_(torch.from_numpy(fs.inst_scale).to(scl.device))
    r.step()
    assert_parity(fetch(r), fs.render())          # ids written but not bound: still nothing new
    r.refresh_objects()
    fs.refresh_objects()
    r.step()
    got, ref = fetch(r), fs.render()
    assert_parity(got, ref)
    assert ref["tri_id"].max() >= (14 if kind != "bvh" else 362)              # the new cubes' triangles are visible
    # ... and the same picture as a renderer-independent restatement: a STATIC scene that holds the spawned
    # cubes as ordinary instances from the start (no spare rows, no refresh on either side)
    static = scenes.SceneDesc(**{k: getattr(d, k) for k in d.__dataclass_fields__})
    static.max_instances_per_world = 0
    inst, worlds = [], []
    for w in range(d.num_worlds):
        lo, hi = int(fs.world_inst_start[w]), int(fs.world_inst_start[w + 1])
        rows_w = [i for i in range(lo, hi) if fs.inst_obj0[i] >= 0]
        worlds.append((len(rows_w), len(inst), d.worlds[w][2], d.worlds[w][3]))
        for i in rows_w:
            inst.append((tuple(float(x) for x in fs.inst_pos[i]), tuple(float(x) for x in fs.inst_rot[i]),
                         tuple(float(x) for x in fs.inst_scale[i]), int(fs.inst_obj0[i])))
    static.instances, static.worlds = inst, worlds
    ref_static = render_oracle(static)
    for k in ("rgb", "tri_id", "segmask"):
        assert np.array_equal(ref[k], ref_static[k]), k
    assert np.array_equal(ref["depth"], ref_static["depth"])
    # hide one again by sign, swap the geometry of another (cube -> plane object 1), refresh
    fs.inst_obj[rows[0]] = -1
    fs.inst_obj[rows[1]] = 1
    obj.copy_(torch.from_numpy(fs.inst_obj).to(obj.device))
    r.refresh_objects()
    fs.refresh_objects()
    r.step()
    assert_parity(fetch(r), fs.render())
    # refresh with nothing changed is a no-op
    r.refresh_objects()
    r.step()
    assert_parity(fetch(r), fs.render())


def test_spawn_crosses_the_kernel_threshold(native):
    # binding enough triangles moves a renderer from the raster group kernel to the BVH path
    import torch
    from oracle import oracle
    d = meshes.cube_field(num_worlds=3, cubes=4)          # 50 triangles: group kernel (small batches cross at 65)
    d.max_instances_per_world = 20
    r = make_product(d, visibility=True)
    fs = oracle.FlatScene(d)
    assert r.render_path() == "raster"
    obj = r.instance_object_tensor().to_torch()
    pos = r.instance_position_tensor().to_torch()
    rng = np.random.default_rng(3)
    for row in np.flatnonzero(fs.inst_obj0 < 0):
        fs.inst_obj[row] = 0
        fs.inst_pos[row] = np.float32(rng.uniform(-6, 6, 3) * np.array([1, 1, 0]) + np.array([0, 0, 0.5]))
    obj.copy_(torch.from_numpy(fs.inst_obj).to(obj.device))
    pos.copy_(torch.from_numpy(fs.inst_pos).to(pos.device))
    r.refresh_objects()
    fs.refresh_objects()
    r.step()
    assert r.render_path() == "bvh"                       # 19 cubes + plane = 230 triangles
    assert_parity(fetch(r), fs.render())


def test_spawned_rows_on_a_renderer_of_three_shards_agree_with_the_float64_ray_caster(native, monkeypatch):
    # spawned rows and multi-shard renders were checked against the oracle only.
    # Here the PRODUCT's pixels -- cubes bound at run time into spare rows, on a renderer of three
    # shards -- are compared with the independent float64 Moeller-Trumbore ray caster and shading
    # model of tests/test_independent_raycast.py (world-space rays, no edge functions, no S6b, NumPy):
    # wherever float64 is decisive the product names the same triangle, the same depth to 1e-4 and
    # the same colour byte.
    import torch
    from oracle import oracle
    from tests.test_independent_raycast import raycast_colour, raycast_view
    d = scenes.synthetic_scene(20, with_wall=True, textured=True)
    d.max_instances_per_world = 5
    monkeypatch.setenv("MADRONA_MI355_VISIBILITY", "1")
    monkeypatch.setenv("MRX_SHARD_THREADS", "2")
    r = scenes.make_renderer(d, device_ids=[0, 0, 0])
    fs = oracle.FlatScene(d)                      # (used as the scene's data container and kept in step by hand)
    rng = np.random.default_rng(5)
    spare = np.flatnonzero(fs.inst_obj0 < 0)
    for w in range(d.num_worlds):
        lo, hi = fs.world_inst_start[w], fs.world_inst_start[w + 1]
        for row in spare[(spare >= lo) & (spare < hi)][:2]:
            fs.inst_obj[row] = 0
            fs.inst_pos[row] = np.float32(rng.uniform(-3, 3, 3) + np.array([0, 0, 2.0]))
            fs.inst_scale[row] = np.float32(rng.uniform(0.5, 1.5))
    for i in range(3):
        lo, hi = scenes.shard_range(d.num_worlds, i, 3)
        a, b = int(fs.world_inst_start[lo]), int(fs.world_inst_start[hi])
        for t, src in ((r.instance_object_tensor(shard=i), fs.inst_obj), (r.instance_position_tensor(shard=i), fs.inst_pos),
                       (r.instance_scale_tensor(shard=i), fs.inst_scale)):
            tt = t.to_torch()
            tt.copy_(torch.from_numpy(np.ascontiguousarray(src[a:b])).to(tt.device))
    r.refresh_objects()
    fs.refresh_objects()
    r.step()
    got = _shard_outputs(r, 3)
    checked = spawned_seen = 0
    for v in (0, 6, 7, 13, 19):                   # views of all three shards (7 + 7 + 6 worlds)
        tri, depth, margin = raycast_view(fs, v)
        sure = margin > 1e-5
        assert np.array_equal(got["ids"][v][sure], tri[sure]), \
            f"view {v}: {(got['ids'][v][sure] != tri[sure]).sum()} decisive pixels name another triangle"
        # depth: 1e-4 on everything but the ground quad, whose float32 1/depth plane over +-10000 units is that
        # coarse by itself (k = 0, 1: plane.obj is the world's first instance; the oracle shows the same against
        # float64 -- tests/test_independent_raycast.py -- and the product equals the oracle bit for bit)
        ground = tri < 2
        np.testing.assert_allclose(got["depth"][v][sure & ~ground], depth[sure & ~ground], rtol=1e-4)
        np.testing.assert_allclose(got["depth"][v][sure & ground], depth[sure & ground], rtol=3e-4)
        rgb, sure_tex = raycast_colour(fs, v)
        ok = sure & sure_tex & (tri >= 0) & (margin > 1e-4)
        diff = np.abs(got["rgb"][v][..., :3].astype(np.float64) - np.floor(rgb + 0.5))
        near_half = np.abs(rgb - np.floor(rgb) - 0.5) < 0.02
        assert not (ok[..., None] & (diff > np.where(near_half, 1.0, 0.0))).any()
        checked += int(sure.sum())
        spawned_seen += int((tri[sure] >= 26).sum())      # cube + plane + wall = 26 triangles: beyond = spawned cubes
    assert checked > 10000 and spawned_seen > 50
