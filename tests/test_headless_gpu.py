"""The headless CLI (reference: src/headless.cpp, src/args.cpp, src/dump.cpp):
argument contract, the two output lines, and the tiled last-frame dump checked
against the oracle image of the same scene."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

from madrona_renderer_amd import build, scenes
from tests.util import render_oracle

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    exe = build.headless_path()
    assert os.path.exists(exe), "renderer_headless not built"
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True,
                          text=True, timeout=300)


def _tiles(png, n, res_x, res_y):
    from PIL import Image
    img = np.asarray(Image.open(png).convert("RGBA"))
    ty = math.ceil(math.sqrt(n))
    tx = math.ceil(n / ty)
    assert img.shape == (ty * res_y, tx * res_x, 4)
    return [img[(i // tx) * res_y:(i // tx + 1) * res_y, (i % tx) * res_x:(i % tx + 1) * res_x]
            for i in range(n)]


def test_usage_errors(native, tmp_path):
    assert _run([], tmp_path).returncode != 0
    r = _run([4, 2, "vulkan", 64, 64], tmp_path)
    assert r.returncode != 0 and "NUM_WORLDS" in r.stderr


@pytest.mark.parametrize("mode", ["rast", "rt"])
def test_demo_scene_dump_matches_oracle(native, tmp_path, mode):
    r = _run([5, 3, mode, 64, 64, "--dump-last-frame", "frame", "--scene", "demo"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert re.search(r"^FPS [0-9.]+$", r.stdout, re.M)
    assert re.search(r"^Average total step time: [0-9.]+ ms$", r.stdout, re.M)
    ref = render_oracle(scenes.demo_scene(num_worlds=5, render_mode="Rasterizer"))
    for i, tile in enumerate(_tiles(tmp_path / "frame.png", 5, 64, 64)):
        # the dump un-transposes Raytracer storage, so both modes show the same picture
        assert np.array_equal(tile, ref["rgb"][i])


def test_synthetic_scene_matches_python_generator(native, tmp_path):
    r = _run([9, 1, "rast", 64, 64, "--dump-last-frame", "syn"], tmp_path)
    assert r.returncode == 0, r.stderr
    ref = render_oracle(scenes.synthetic_scene(9))
    for i, tile in enumerate(_tiles(tmp_path / "syn.png", 9, 64, 64)):
        assert np.array_equal(tile, ref["rgb"][i]), f"world {i}"
    r = _run([9, 1, "rast", 64, 64, "--dump-last-frame", "dep", "--depth"], tmp_path)
    assert r.returncode == 0, r.stderr
    for i, tile in enumerate(_tiles(tmp_path / "dep.png", 9, 64, 64)):
        g = (255.0 * np.minimum(ref["depth"][i] / 255.0, 1.0)).astype(np.uint8)
        assert np.abs(tile[..., 0].astype(int) - g.astype(int)).max() <= 1


def test_sharded_over_devices_renders_the_same_worlds(native, tmp_path, monkeypatch):
    # --gpus N: ONE Manager over N devices (Config::deviceIDs), contiguous world shards.
    # On this one-GPU box the rehearsal switch puts every shard on device 0; the
    # per-shard dumps must tile exactly the worlds a single renderer draws.
    r = _run([10, 2, "rast", 64, 64, "--gpus", "3"], tmp_path)
    assert r.returncode != 0 and "HIP device" in r.stderr          # only one device here
    monkeypatch.setenv("MRX_HEADLESS_REHEARSAL", "1")
    r = _run([10, 2, "rast", 64, 64, "--gpus", "3", "--dump-last-frame", "part"], tmp_path)
    assert r.returncode == 0, r.stderr
    lines = re.findall(r"^GPU 0: worlds \[(\d+), (\d+)\) FPS [0-9.]+$", r.stdout, re.M)
    assert [(int(a), int(b)) for a, b in lines] == [scenes.shard_range(10, k, 3) for k in range(3)]
    assert re.search(r"^FPS [0-9.]+$", r.stdout, re.M)
    ref = render_oracle(scenes.synthetic_scene(10))
    for k in range(3):
        lo, hi = scenes.shard_range(10, k, 3)
        for i, tile in enumerate(_tiles(tmp_path / f"part.gpu{k}.png", hi - lo, 64, 64)):
            assert np.array_equal(tile, ref["rgb"][lo + i]), f"shard {k} world {lo + i}"
