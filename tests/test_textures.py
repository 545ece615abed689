"""KTX2 / BC7 textures (the reference's "ktx2" handler,
/root/reference/src/mgr.cpp:199-212,297-298): the C++ container reader and BC7
block decoder of madrona_renderer_amd/csrc/ktx2.cpp against an independent
decode (Python struct + Pillow's BC7) -- on random blocks of every mode, on the
committed fixtures of tests/golden/, and in a render."""
import ctypes
import os
import struct

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.util import assert_parity, fetch, make_product, render_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXPECTED = np.load(os.path.join(GOLDEN, "ktx2_expected.npz"))


def _decode_texture(lib, path):
    img = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.mrx_decode_texture(str(path).encode(), ctypes.byref(img), ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        lib.mrx_last_error.restype = ctypes.c_char_p
        return rc, lib.mrx_last_error().decode()
    a = np.ctypeslib.as_array(img, shape=(h.value, w.value, 4)).copy()
    lib.mrx_free(img)
    return 0, a


@pytest.mark.parametrize("mode", range(8))
def test_bc7_block_decoder_matches_pillow_on_random_blocks(native, oracle_mod, mode):
    lib = native.load_capi()
    rng = np.random.default_rng(100 + mode)
    n = 512                                           # a 128 x 64 image of random blocks
    raw = bytearray()
    for _ in range(n):
        v = int.from_bytes(rng.bytes(16), "little")
        v = (v & ~((1 << (mode + 1)) - 1)) | (1 << mode)
        raw += v.to_bytes(16, "little")
    raw = bytes(raw)
    out = np.zeros((n, 16, 4), np.uint8)
    assert lib.mrx_decode_bc7(raw, ctypes.c_uint32(n), out.ctypes.data_as(ctypes.c_void_p)) == 0
    ref = oracle_mod.decode_bc7(raw, 128, 64)         # 32 x 16 blocks
    mine = out.reshape(16, 32, 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(64, 128, 4)
    assert np.array_equal(mine, ref), f"mode {mode}: {(mine != ref).any(axis=-1).sum()} pixels differ"


def test_reserved_bc7_block_decodes_to_transparent_black(native):
    lib = native.load_capi()
    out = np.full((1, 16, 4), 7, np.uint8)
    assert lib.mrx_decode_bc7(bytes(16), ctypes.c_uint32(1), out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert not out.any()


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_ktx2_fixtures_decode_like_the_independent_reader(native, oracle_mod, name):
    lib = native.load_capi()
    path = os.path.join(GOLDEN, name)
    rc, img = _decode_texture(lib, path)
    assert rc == 0, img
    assert np.array_equal(img, EXPECTED[name])                    # the committed expectation
    assert np.array_equal(img, oracle_mod.decode_image(path))     # ... and today's independent decode


def test_mode6_fixture_is_a_faithful_picture_of_cube_png(native):
    from PIL import Image
    with Image.open(os.path.join(scenes.DATA_DIR, "cube.png")) as im:
        cube = np.asarray(im.convert("RGBA").resize((64, 64), Image.NEAREST)).astype(int)
    assert np.abs(EXPECTED["cube64_bc7.ktx2"].astype(int) - cube).mean() < 2.0


def test_ktx2_reader_refuses_what_it_cannot_read(native, tmp_path):
    lib = native.load_capi()
    good = open(os.path.join(GOLDEN, "bc7_modes.ktx2"), "rb").read()

    def attempt(data, name="t.ktx2"):
        p = tmp_path / name
        p.write_bytes(data)
        return _decode_texture(lib, p)
    assert attempt(good)[0] == 0
    rc, msg = attempt(b"not a texture at all" * 10)
    assert rc != 0 and "KTX2" in msg
    rc, msg = attempt(good[:200])                                  # level data cut off
    assert rc != 0 and "out of range" in msg
    basis = bytearray(good)
    struct.pack_into("<I", basis, 44, 1)                           # supercompression BasisLZ
    rc, msg = attempt(bytes(basis))
    assert rc != 0 and "transcoder" in msg
    etc = bytearray(good)
    struct.pack_into("<I", etc, 12, 147)                           # VK_FORMAT_ETC2_R8G8B8_UNORM_BLOCK
    rc, msg = attempt(bytes(etc))
    assert rc != 0 and "vkFormat 147" in msg
    zstd = bytearray(good)
    struct.pack_into("<I", zstd, 44, 2)                            # says Zstandard, holds raw blocks
    rc, msg = attempt(bytes(zstd))
    assert rc != 0 and "Zstandard payload does not decompress" in msg
    lzma = bytearray(good)
    struct.pack_into("<I", lzma, 44, 4)                            # no such scheme
    rc, msg = attempt(bytes(lzma))
    assert rc != 0 and "supercompression" in msg
    real = bytearray(open(os.path.join(GOLDEN, "bc7_modes_zstd.ktx2"), "rb").read())
    assert attempt(bytes(real))[0] == 0
    off = struct.unpack_from("<Q", real, 80)[0]
    real[off] ^= 0xFF                                              # not a Zstandard frame any more (magic number)
    assert attempt(bytes(real))[0] != 0
    cube = bytearray(good)
    struct.pack_into("<I", cube, 36, 6)                            # six faces
    assert attempt(bytes(cube))[0] != 0
    # the same bytes under a .png name go to the PNG reader, which refuses them too
    assert attempt(good, "t.png")[0] != 0


@pytest.mark.gpu
def test_scene_textured_from_ktx2_files(native, tmp_path):
    # API texture from a .ktx2 (BC7), and an OBJ whose MTL names a .ktx2 map_Kd
    import shutil
    shutil.copy(os.path.join(GOLDEN, "bc7_modes.ktx2"), tmp_path / "noise.ktx2")
    (tmp_path / "q.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd noise.ktx2\n")
    (tmp_path / "q.obj").write_text(
        "mtllib q.mtl\nv -2 0 -2\nv 2 0 -2\nv 2 0 2\nv -2 0 2\nvt 0 0\nvt 3 0\nvt 3 3\nvt 0 3\n"
        "usemtl m\nf 1/1 2/2 3/3 4/4\n")
    for mode in ("Rasterizer", "Raytracer"):
        d = scenes.synthetic_scene(24, with_wall=True, textured=True, render_mode=mode, width=96, height=96)
        d.texture_paths = [os.path.join(GOLDEN, "cube64_bc7.ktx2")]
        d.asset_paths = d.asset_paths + [(str(tmp_path / "q.obj"), -1)]
        inst, worlds = [], []
        for w in range(24):
            rows = d.instances[3 * w:3 * w + 3] + [((0.0, -3.0, 2.5), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 3)]
            worlds.append((4, len(inst), 1, w))
            inst += rows
        d.instances, d.worlds = inst, worlds
        r = make_product(d, visibility=True)
        got, ref = fetch(r), render_oracle(d)
        assert_parity(got, ref)
        colours = {tuple(c) for c in got["rgb"].reshape(-1, 4)[::7].tolist()}
        assert len(colours) > 200                     # the noise texture shows
