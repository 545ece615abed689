"""CPU-only checks of the host side: the C-ABI library loads and exports what
include/mrx.h declares, the asset readers agree with independent decoders, the
API fails loudly without a GPU, and the scene helpers are deterministic."""
import ctypes
import os
import re

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.conftest import ROOT, has_gpu


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mrx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrx_[a-z_0-9]+)\s*\(", text)))


def test_capi_exports_every_declared_symbol(native):
    lib = native.load_capi()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libmrx_hip.so does not export {n}"
    lib.mrx_abi_version.restype = ctypes.c_int
    assert lib.mrx_abi_version() == 4


def test_module_surface_matches_reference_bindings(native):
    # /root/reference/src/bindings.cpp:21-233
    m = native.load_module()
    for cls in ("RenderMode", "ImportedAsset", "AdditionalMaterial", "ImportedInstance",
                "ImportedCamera", "WorldInit", "MadronaRenderer", "inspect"):
        assert hasattr(m, cls), cls
    assert {"Rasterizer", "Raytracer"} <= set(m.RenderMode.__members__)
    for meth in ("step", "rgb_tensor", "depth_tensor", "segmask_tensor", "rgb_cuda_ptr",
                 "depth_cuda_ptr", "segmask_cuda_ptr", "instance_position_tensor",
                 "instance_rotation_tensor", "camera_position_tensor",
                 "camera_rotation_tensor"):
        assert hasattr(m.MadronaRenderer, meth), meth
    m.ImportedInstance(position=[0, 0, 0], rotation=[1, 0, 0, 0], scale=[1, 1, 1], object_id=0)
    m.WorldInit(num_instances=1, instance_offset=0, num_cameras=1, camera_offset=0)
    with pytest.raises(TypeError):
        m.ImportedInstance(position=[0, 0], rotation=[1, 0, 0, 0], scale=[1, 1, 1], object_id=0)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_a_gpu(native):
    with pytest.raises(RuntimeError, match="no HIP device"):
        scenes.make_renderer(scenes.demo_scene())
    lib = native.load_capi()
    assert lib.mrx_device_count() == 0


def _load_obj(lib, path):
    pp = ctypes.POINTER(ctypes.c_float)()
    uu = ctypes.POINTER(ctypes.c_float)()
    n = ctypes.c_uint32()
    rc = lib.mrx_load_obj(path.encode(), ctypes.byref(pp), ctypes.byref(uu), ctypes.byref(n))
    if rc != 0:
        return rc, None, None
    pos = np.ctypeslib.as_array(pp, shape=(n.value, 3, 3)).copy() if n.value else np.zeros((0, 3, 3))
    uv = np.ctypeslib.as_array(uu, shape=(n.value, 3, 2)).copy() if n.value else np.zeros((0, 3, 2))
    lib.mrx_free(pp)
    lib.mrx_free(uu)
    return 0, pos, uv


def _decode_png(lib, path):
    img = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.mrx_decode_png(path.encode(), ctypes.byref(img), ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        return rc, None
    a = np.ctypeslib.as_array(img, shape=(h.value, w.value, 4)).copy()
    lib.mrx_free(img)
    return 0, a


@pytest.mark.parametrize("name,ntris", [("cube.obj", 12), ("plane.obj", 2),
                                        ("wall_render.obj", 12)])
def test_obj_reader_matches_independent_parser(native, oracle_mod, name, ntris):
    lib = native.load_capi()
    path = os.path.join(scenes.DATA_DIR, name)
    rc, pos, uv = _load_obj(lib, path)
    assert rc == 0 and pos.shape == (ntris, 3, 3)
    rp, ru = oracle_mod.parse_obj(path)
    assert np.array_equal(pos, rp) and np.array_equal(uv, ru)


def test_obj_reader_polygons_negative_indices_missing_uv(native, oracle_mod, tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# quad + pentagon, relative indices, no vt on the second face\n"
                 "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0.25\n"
                 "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                 "f 1/1 2/2 3/3 4/4\n"
                 "f -5//1 -4//1 -3//1 -1//1 -2//1\n")
    lib = native.load_capi()
    rc, pos, uv = _load_obj(lib, str(p))
    assert rc == 0 and pos.shape == (2 + 3, 3, 3)
    rp, ru = oracle_mod.parse_obj(str(p))
    assert np.array_equal(pos, rp) and np.array_equal(uv, ru)
    assert np.all(uv[2:] == 0)


def test_obj_material_statements_match_independent_parser(native, oracle_mod, tmp_path):
    import json
    lib = native.load_capi()
    lib.mrx_describe_obj_materials.restype = ctypes.c_int64
    (tmp_path / "a.mtl").write_text("# comment\nnewmtl one\nKd 0.25 0.5 0.75\nmap_Kd -s 1 1 1 tex/one.png\n"
                                    "newmtl two words\nKd 1 0 0\n")
    (tmp_path / "m.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\n"
                                    "f 1 2 3\nusemtl two words\nf 2 4 3\nusemtl one\nf 1 2 4 3\n"
                                    "usemtl two words\nf 1 4 3\n")
    for path in (str(tmp_path / "m.obj"), os.path.join(scenes.DATA_DIR, "cube.obj"),
                 os.path.join(scenes.DATA_DIR, "plane.obj")):
        buf = ctypes.create_string_buffer(1 << 16)
        n = lib.mrx_describe_obj_materials(path.encode(), buf, ctypes.c_uint64(len(buf)))
        assert n > 0
        got = json.loads(buf.value.decode())
        pos, uv, tri_mtl, names, libs = oracle_mod.parse_obj(path, with_materials=True)
        assert got["num_tris"] == len(pos)
        assert got["tri_mtl"] == tri_mtl.tolist() and got["names"] == names and got["libs"] == libs
        ref = [m for ml in libs for m in oracle_mod.parse_mtl(ml)]
        assert [m["name"] for m in got["materials"]] == [m[0] for m in ref]
        for g, r in zip(got["materials"], ref):
            assert np.array_equal(np.float32(g["kd"]), np.float32(r[1]))
            assert g["map_kd"] == (r[2] or "")
    # cube.obj names cube.mtl -> Kd 0.588, map_Kd cube.png (data/cube.mtl:7,13)
    assert got is not None


def test_obj_reader_errors(native, tmp_path):
    lib = native.load_capi()
    lib.mrx_last_error.restype = ctypes.c_char_p
    rc, _, _ = _load_obj(lib, str(tmp_path / "missing.obj"))
    assert rc == -4 and b"cannot open" in lib.mrx_last_error()
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    rc, _, _ = _load_obj(lib, str(bad))
    assert rc == -4 and b"out of range" in lib.mrx_last_error()
    empty = tmp_path / "empty.obj"
    empty.write_text("")
    rc, pos, _ = _load_obj(lib, str(empty))
    assert rc == 0 and pos.shape[0] == 0


def test_png_decoder_reference_texture(native, oracle_mod):
    # data/cube.png: 256x256, 8-bit palette, one IDAT
    lib = native.load_capi()
    path = os.path.join(scenes.DATA_DIR, "cube.png")
    rc, a = _decode_png(lib, path)
    assert rc == 0 and a.shape == (256, 256, 4)
    assert np.array_equal(a, oracle_mod.decode_image(path))


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "P-trns", "1", "I;16"])
def test_png_decoder_variants_match_pillow(native, oracle_mod, tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(7)
    w, h = 37, 23                      # odd sizes exercise the filters' edges
    if mode in ("RGB", "RGBA", "L", "LA"):
        ch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2}[mode]
        # smooth + noise so every PNG filter type gets picked
        base = (np.add.outer(np.arange(h), np.arange(w)) * 3)[..., None]
        arr = ((base + rng.integers(0, 40, (h, w, ch))) % 256).astype(np.uint8)
        im = Image.fromarray(arr.squeeze() if ch == 1 else arr, mode)
    elif mode.startswith("P"):
        arr = rng.integers(0, 200, (h, w)).astype(np.uint8)
        im = Image.fromarray(arr, "P")
        im.putpalette(rng.integers(0, 256, 768).astype(np.uint8).tolist())
        if mode == "P-trns":
            im.info["transparency"] = bytes(rng.integers(0, 256, 200).astype(np.uint8))
    elif mode == "1":
        im = Image.fromarray(rng.integers(0, 2, (h, w)).astype(bool))
    else:
        im = Image.fromarray(rng.integers(0, 65536, (h, w)).astype(np.uint16))
    path = str(tmp_path / f"t_{mode.replace(';', '')}.png")
    kw = {"transparency": im.info["transparency"]} if mode == "P-trns" else {}
    im.save(path, **kw)
    lib = native.load_capi()
    rc, a = _decode_png(lib, path)
    assert rc == 0
    if mode == "I;16":
        # 16-bit samples are reduced to their high byte
        with Image.open(path) as r:
            hi = (np.asarray(r).astype(np.uint16) >> 8).astype(np.uint8)
        assert np.array_equal(a[..., 0], hi) and np.all(a[..., 3] == 255)
    else:
        assert np.array_equal(a, oracle_mod.decode_image(path))


def test_png_decoder_errors(native, tmp_path):
    lib = native.load_capi()
    lib.mrx_last_error.restype = ctypes.c_char_p
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png at all")
    rc, _ = _decode_png(lib, str(p))
    assert rc == -4 and b"not a PNG" in lib.mrx_last_error()
    good = open(os.path.join(scenes.DATA_DIR, "cube.png"), "rb").read()
    p.write_bytes(good[:2000])
    rc, _ = _decode_png(lib, str(p))
    assert rc == -4


def test_pod_layouts_match_the_reference_structs():
    # sizes fixed by /root/reference/src/sim.hpp:31-50,76-82 and bindings.cpp:44-49
    text = open(os.path.join(ROOT, "include", "madrona_mi355", "types.hpp")).read()
    for s, n in (("ImportedInstance", 44), ("ImportedCamera", 28),
                 ("Sim::WorldInit", 16), ("madrona::imp::SourceMaterial", 28)):
        assert f"sizeof({s}) == {n}" in text


def test_rng_and_scene_are_deterministic():
    # splitmix64 known answers (seed 0x4D52584D xor k), exact in float32
    u = scenes.uniform(np.arange(4))
    assert np.all((u >= 0) & (u < 1))
    assert np.array_equal(u.astype(np.float32).astype(np.float64), u)
    a = scenes.synthetic_scene(8, with_wall=True)
    b = scenes.synthetic_scene(8, with_wall=True)
    assert a.instances == b.instances and a.cameras == b.cameras
    # a shard generated on its own equals the slice of the whole job's scene
    whole = scenes.synthetic_scene(8)
    part = scenes.synthetic_scene(4, first_world=4)
    assert part.instances == whole.instances[8:] and part.cameras == whole.cameras[4:]
    # every camera looks at (0,0,1): its +Y axis points from eye to target
    for (eye, q) in whole.cameras:
        w, x, y, z = q
        fwd = np.array([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)])
        to = np.array([0, 0, 1.0]) - np.array(eye)
        assert np.dot(fwd, to / np.linalg.norm(to)) > 0.9999


def test_shard_ranges_partition_the_worlds():
    for n in (0, 1, 7, 8, 4096, 16384, 16385):
        for ws in (1, 2, 3, 8):
            spans = [scenes.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    d = scenes.synthetic_scene(10)
    s = d.shard(1, 4)
    assert s.num_worlds == 3 and s.worlds == d.worlds[3:6]


MULTI_OBJ = """# three objects: a quad, a triangle fan, and one more quad; blocks without faces open nothing
v 0 0 0
v 1 0 0
v 1 0 1
v 0 0 1
o first
f 1 2 3 4
g
o second
v 2 0 0
v 3 0 0
v 3 0 1
v 2 0 1
v 2.5 0 2
f 5 6 7 8 9
o empty_block
g third
f -5 -4 -1
f 1 2 3
"""


def test_obj_file_is_one_object_and_blocks_split_on_request(native, oracle_mod, tmp_path, monkeypatch):
    # /root/reference/src/mgr.cpp:301-303,340-345: importFromDisk(..., one_object_per_asset)
    # and objects[i] <-> asset path i -- a file is one object whatever blocks it holds
    # (ADVICE r2); MRX_OBJ_SPLIT_BLOCKS=1 makes one object per `o` / `g` block with faces
    path = tmp_path / "multi.obj"
    path.write_text(MULTI_OBJ)
    lib = native.load_capi()
    first = (ctypes.c_uint32 * 8)()
    n = lib.mrx_obj_objects(str(path).encode(), first, 8)
    assert n == 3 and list(first[:3]) == [0, 2, 5]
    pos, uv, starts = oracle_mod.parse_obj(str(path), with_objects=True)
    assert starts == [0, 2, 5] and len(pos) == 7
    rc, cpos, _ = _load_obj(lib, str(path))
    assert rc == 0 and np.array_equal(cpos, pos)
    # the files of data/ hold one block each
    assert lib.mrx_obj_objects(os.path.join(scenes.DATA_DIR, "cube.obj").encode(), first, 8) == 1
    # in a scene: the file is object 0, the next asset 1, the raw mesh 2 ...
    d = scenes.SceneDesc(
        num_worlds=1, asset_paths=[(str(path), -1), (os.path.join(scenes.DATA_DIR, "plane.obj"), -1)],
        mesh_vertices=np.zeros((3, 3), np.float32), mesh_uvs=np.zeros((3, 2), np.float32),
        mesh_indices=np.arange(3, dtype=np.uint32), mesh_vertex_offsets=np.zeros(1, np.uint32),
        mesh_indices_offsets=np.zeros(1, np.uint32), mesh_materials=np.array([-1], np.int32),
        instances=[((0.0, 3.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 2)],
        cameras=[((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))], worlds=[(1, 0, 1, 0)])
    monkeypatch.delenv("MRX_OBJ_SPLIT_BLOCKS", raising=False)
    fs = oracle_mod.FlatScene(d)
    assert fs.obj_first_tri.tolist() == [0, 7, 9] and fs.obj_num_tris.tolist() == [7, 2, 1]
    # ... on request the file's blocks take ids 0..2, the next asset id 3, the raw mesh id 4
    monkeypatch.setenv("MRX_OBJ_SPLIT_BLOCKS", "1")
    fs = oracle_mod.FlatScene(d)
    assert fs.obj_first_tri.tolist() == [0, 2, 5, 7, 9] and fs.obj_num_tris.tolist() == [2, 3, 2, 2, 1]


def test_library_shard_split_is_scenes_shard_range(native):
    # single-process multi-device (mrx_config.device_ids): the library splits the worlds exactly
    # as the one-process-per-GPU path does (scenes.shard_range), so shard i of a multi-device
    # renderer holds the worlds rank i of a torchrun job would
    lib = native.load_capi()
    lib.mrx_shard_split.restype = ctypes.c_int64
    lib.mrx_shard_split.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    for n in (0, 1, 7, 8, 9, 4096, 16384, 1000003):
        for k in (1, 2, 3, 4, 8, 64):
            for i in range(k):
                lo, hi = scenes.shard_range(n, i, k)
                assert lib.mrx_shard_split(n, i, k) == lo and lib.mrx_shard_split(n, i + 1, k) == hi
    assert lib.mrx_shard_split(10, 5, 4) < 0 and lib.mrx_shard_split(10, 0, 0) < 0


def test_oracle_spare_rows_and_rebinding(oracle_mod):
    # max_instances_per_world: spare rows start hidden and unbound; refresh_objects binds a row to
    # the id its ObjectID column holds, and the world-local triangle numbering follows row order
    d = scenes.synthetic_scene(3)
    d.max_instances_per_world = 4
    fs = oracle_mod.FlatScene(d)
    assert fs.world_inst_start.tolist() == [0, 4, 8, 12]
    assert fs.inst_obj.tolist() == [1, 0, -1, -1] * 3 and np.array_equal(fs.inst_obj, fs.inst_obj0)
    plain = oracle_mod.FlatScene(scenes.synthetic_scene(3)).render()
    before = fs.render()
    for k in ("rgb", "depth", "tri_id", "segmask"):
        assert np.array_equal(plain[k], before[k]), k             # unbound rows draw nothing
    # spawn a second cube in world 1 (row 6): id written, then bound
    fs.inst_obj[6] = 0
    fs.inst_pos[6] = (1.5, -2.0, 2.0)
    same = fs.render()
    assert np.array_equal(same["tri_id"], before["tri_id"])        # not bound yet: only the sign counts
    fs.refresh_objects()
    after = fs.render()
    assert not np.array_equal(after["tri_id"][1], before["tri_id"][1])
    assert np.array_equal(after["tri_id"][0], before["tri_id"][0]) and np.array_equal(after["tri_id"][2], before["tri_id"][2])
    assert after["tri_id"][1].max() >= 14                          # its triangles are numbered after the cube's and plane's
    # a different non-negative id without refresh changes neither geometry nor labels
    fs.inst_obj[0] = 0
    again = fs.render()
    assert np.array_equal(again["segmask"], after["segmask"]) and np.array_equal(again["rgb"], after["rgb"])


def test_dispatch_constants_follow_the_cu_count(native):
    # the small-batch BVH thresholds and the "chip is full" workgroup count were
    # MI355X literals (640 / 1024 views, 1024 workgroups); they are functions of the device's CU
    # count now -- pure functions, checked here with faked devices
    lib = native.load_capi()
    f = lib.mrx_dispatch_min_tris
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_uint32] * 2 + [ctypes.c_int] + [ctypes.c_uint32] * 3
    g = lib.mrx_group_fill
    g.restype = ctypes.c_uint32
    g.argtypes = [ctypes.c_uint32]
    assert [g(c) for c in (256, 128, 32)] == [1024, 512, 128]
    for cus in (256, 128, 32):
        small, mid = cus * 5 // 2, cus * 4
        # up to 2.5 views per CU: from 65 triangles, textured or not
        assert f(0, small, 0, 64, 64, cus) == 65 and f(0, small, 1, 64, 64, cus) == 65
        # up to 4 per CU: from 91, untextured worlds only
        assert f(0, small + 1, 0, 64, 64, cus) == 91 and f(0, small + 1, 1, 64, 64, cus) == 129
        assert f(0, mid, 0, 64, 64, cus) == 91 and f(0, mid + 1, 0, 64, 64, cus) == 129
        # views of several tiles: the general threshold
        assert f(0, 8, 0, 128, 64, cus) == 129
        # an explicit threshold is never raised
        assert f(40, 8, 0, 64, 64, cus) == 40 and f(200, small, 0, 64, 64, cus) == 65
    # Raytracer-mode batches of small worlds that take the BVH path's flat kernel by default: BASELINE configs[4] does,
    # an eighth of it does not, Rasterizer mode never, views of fewer than 16 tiles never, worlds beyond 64 triangles never
    h = lib.mrx_dispatch_flat
    h.restype = ctypes.c_int
    h.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 6
    assert h(1, 4096, 256, 256, 14, 2, 256) == 1 and h(1, 512, 256, 256, 14, 2, 256) == 0
    assert h(1, 3072, 256, 256, 14, 2, 256) == 1 and h(1, 3071, 256, 256, 14, 2, 256) == 0
    assert h(0, 4096, 256, 256, 14, 2, 256) == 0 and h(1, 16384, 128, 128, 14, 2, 256) == 0
    assert h(1, 4096, 256, 256, 65, 2, 256) == 0 and h(1, 4096, 256, 256, 14, 65, 256) == 0
    assert h(1, 512, 256, 256, 14, 2, 32) == 1 and h(1, 383, 256, 256, 14, 2, 32) == 0
    # the MI355X values the measurements were taken at (profiles/r03_bvh_threshold.txt)
    assert f(0, 640, 1, 64, 64, 256) == 65 and f(0, 641, 0, 64, 64, 256) == 91 and f(0, 1025, 0, 64, 64, 256) == 129


def test_new_entry_points_reject_bad_arguments_without_a_device(native):
    # ABI 4 additions: argument checks come before anything touches HIP
    lib = native.load_capi()
    lib.mrx_last_error.restype = ctypes.c_char_p
    buf = (ctypes.c_uint8 * 96)()
    lib.mrx_info_sized.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    assert lib.mrx_info_sized(None, ctypes.byref(buf), 80) == -1 and b"null" in lib.mrx_last_error()
    assert lib.mrx_info(None, ctypes.byref(buf)) == -1
    us = ctypes.c_double()
    lib.mrx_time_steps_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    assert lib.mrx_time_steps_host(None, 10, ctypes.byref(us)) == -1
    lib.mrx_shard_split.restype = ctypes.c_int64
    lib.mrx_shard_split.argtypes = [ctypes.c_uint32] * 3
    # the split every form of sharding uses (scenes.shard_range): contiguous, sizes differ by at most one
    for worlds, n in ((16384, 8), (1003, 8), (7, 3), (5, 5)):
        cuts = [lib.mrx_shard_split(worlds, i, n) for i in range(n + 1)]
        assert cuts[0] == 0 and cuts[-1] == worlds
        assert [(cuts[i], cuts[i + 1]) for i in range(n)] == [scenes.shard_range(worlds, i, n) for i in range(n)]
