"""Writes the KTX2 fixtures of tests/golden/ (the reference ships no .ktx2 file;
its "ktx2" handler is /root/reference/src/mgr.cpp:199-212,297-298):

  cube64_bc7.ktx2      data/cube.png reduced to 64x64, BC7 (mode 6 blocks), vkFormat 145
  bc7_modes.ktx2       32x32, 64 blocks with random payloads, eight of every BC7 mode
  bc7_modes_zlib.ktx2  the same level, supercompression scheme 3 (ZLIB), vkFormat 146
  bc7_modes_zstd.ktx2  the same level, supercompression scheme 2 (Zstandard, written with
                       pyarrow's codec)
  rgba8_zstd_9x7.ktx2  a 9x7 R8G8B8A8_UNORM image, Zstandard
  rgba8_5x3.ktx2       a 5x3 R8G8B8A8_UNORM image (no block compression, ragged size)
  ktx2_expected.npz    what they decode to, by Pillow's BC7 decoder (independent of
                       madrona_renderer_amd/csrc/ktx2.cpp)

    python tests/golden/make_ktx2.py
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle                        # noqa: E402

IDENT = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])


def ktx2(vk_format, width, height, payload, scheme=0, uncompressed_len=None):
    """A minimal KTX 2.0 file: header, index, one level, a placeholder data
    format descriptor block (readers of the pixel data go by vkFormat), no
    key/value data."""
    dfd = struct.pack("<I", 4 + 40) + bytes(40)                 # size + a blank 40-byte descriptor block
    level_off = 80 + 24 + len(dfd)
    level_off = (level_off + 15) // 16 * 16
    head = IDENT + struct.pack("<9I", vk_format, 1, width, height, 0, 0, 1, 1, scheme)
    index = struct.pack("<4I2Q", 80 + 24, len(dfd), 0, 0, 0, 0)
    ulen = len(payload) if uncompressed_len is None else uncompressed_len
    levels = struct.pack("<3Q", level_off, len(payload), ulen)
    body = head + index + levels + dfd
    return body + bytes(level_off - len(body)) + payload


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, bits):
        self.v |= (int(value) & ((1 << bits) - 1)) << self.n
        self.n += bits

    def block(self):
        assert self.n == 128, self.n
        return self.v.to_bytes(16, "little")


def encode_mode6(px):
    """px [16,4] u8 -> one BC7 mode-6 block: endpoints = per-channel min / max
    (7 bits + a p-bit), 4-bit indices by projection on the endpoint axis."""
    px = px.astype(np.int32)
    lo, hi = px.min(axis=0), px.max(axis=0)
    p0, p1 = int(round(lo.mean())) & 1, int(round(hi.mean())) & 1
    e0 = np.clip((lo - p0 + 1) >> 1, 0, 127)
    e1 = np.clip((hi - p1 + 1) >> 1, 0, 127)
    f0 = ((e0 << 1) | p0).astype(np.float64)
    f1 = ((e1 << 1) | p1).astype(np.float64)
    axis = f1 - f0
    den = float(axis @ axis)
    t = ((px - f0) @ axis / den) if den > 0 else np.zeros(16)
    idx = np.clip(np.rint(t * 15), 0, 15).astype(int)
    if idx[0] >= 8:                                   # anchor: most significant bit of index 0 is implicit 0
        e0, e1, p0, p1 = e1, e0, p1, p0
        idx = 15 - idx
    b = Bits()
    b.put(1 << 6, 7)
    for ch in range(4):
        b.put(e0[ch], 7); b.put(e1[ch], 7)
    b.put(p0, 1); b.put(p1, 1)
    b.put(idx[0], 3)
    for i in range(1, 16):
        b.put(idx[i], 4)
    return b.block()


def encode_bc7_mode6(img):
    h, w = img.shape[:2]
    out = bytearray()
    for by in range(0, h, 4):
        for bx in range(0, w, 4):
            out += encode_mode6(img[by:by + 4, bx:bx + 4].reshape(16, 4))
    return bytes(out)


def random_blocks(rng, per_mode=8):
    out = bytearray()
    for mode in range(8):
        for _ in range(per_mode):
            v = int.from_bytes(rng.bytes(16), "little")
            v &= ~((1 << (mode + 1)) - 1)             # clear the mode field ...
            v |= 1 << mode                            # ... and set this mode's marker bit
            out += v.to_bytes(16, "little")
    return bytes(out)


def main():
    from PIL import Image
    with Image.open(os.path.join(ROOT, "data", "cube.png")) as im:
        cube = np.asarray(im.convert("RGBA").resize((64, 64), Image.NEAREST)).copy()
    files = {
        "cube64_bc7.ktx2": ktx2(145, 64, 64, encode_bc7_mode6(cube)),
    }
    rng = np.random.default_rng(7)
    blocks = random_blocks(rng)
    files["bc7_modes.ktx2"] = ktx2(145, 32, 32, blocks)
    files["bc7_modes_zlib.ktx2"] = ktx2(146, 32, 32, zlib.compress(blocks, 9), scheme=3,
                                        uncompressed_len=len(blocks))
    import pyarrow
    files["bc7_modes_zstd.ktx2"] = ktx2(145, 32, 32, pyarrow.compress(blocks, codec="zstd", asbytes=True), scheme=2,
                                        uncompressed_len=len(blocks))
    # (a compressible image, so the frame holds matches and not only literals)
    grad = np.zeros((7, 9, 4), np.uint8)
    grad[..., 0] = np.arange(9)[None, :] * 28
    grad[..., 1] = np.arange(7)[:, None] * 36
    grad[..., 2] = 77
    grad[..., 3] = 255
    files["rgba8_zstd_9x7.ktx2"] = ktx2(37, 9, 7, pyarrow.compress(grad.tobytes(), codec="zstd", asbytes=True),
                                        scheme=2, uncompressed_len=grad.size)
    small = rng.integers(0, 256, size=(3, 5, 4), dtype=np.uint8)
    files["rgba8_5x3.ktx2"] = ktx2(37, 5, 3, small.tobytes())
    expected = {}
    for name, data in files.items():
        path = os.path.join(HERE, name)
        with open(path, "wb") as f:
            f.write(data)
        expected[name] = oracle.decode_image(path)
    assert np.array_equal(expected["bc7_modes.ktx2"], expected["bc7_modes_zlib.ktx2"])
    assert np.array_equal(expected["bc7_modes.ktx2"], expected["bc7_modes_zstd.ktx2"])
    assert np.array_equal(expected["rgba8_zstd_9x7.ktx2"], grad)
    assert np.array_equal(expected["rgba8_5x3.ktx2"], small)
    err = np.abs(expected["cube64_bc7.ktx2"].astype(int) - cube.astype(int))
    print("mode-6 encoder: mean abs error %.2f, max %d" % (err.mean(), err.max()))
    np.savez_compressed(os.path.join(HERE, "ktx2_expected.npz"), **expected)
    print("wrote", sorted(files), "and ktx2_expected.npz")


if __name__ == "__main__":
    main()
