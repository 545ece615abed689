"""LDS tile-size sweep (BASELINE configs[2]): the BVH kernel's depth-buffer tile
64x64 / 64x32 / 32x32, device us per render."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from madrona_renderer_amd import scenes  # noqa: E402
from tests import meshes  # noqa: E402

os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
CASES = [
    ("C3: 4096 x 128^2 cube+plane+wall (26 tris)", lambda: scenes.synthetic_scene(4096, width=128, height=128, with_wall=True), 20),
    ("1024 x 64^2, 40 cubes (482 tris)", lambda: meshes.cube_field(1024, 40), 200),
    ("1024 x 64^2, 100 cubes (1202 tris)", lambda: meshes.cube_field(1024, 100), 100),
    ("256 x 64^2, 416 cubes (4994 tris)", lambda: meshes.cube_field(256, 416), 50),
    ("256 x 128^2 RT, 100 cubes", lambda: meshes.cube_field(256, 100, width=128, height=128, mode="Raytracer"), 50),
    ("1024 x 64^2 textured, 40 cubes", lambda: meshes.cube_field(1024, 40, textured=True), 100),
]
print("%-48s %12s %12s %12s   (LDS KB / workgroup)" % ("", "64x64", "64x32", "32x32"))
for name, make, steps in CASES:
    desc = make()
    row = []
    for tile in (0, 1, 2):
        os.environ["MRX_BVH_TILE"] = str(tile)
        r = scenes.make_renderer(desc)
        t0 = time.time()
        while time.time() - t0 < 0.2:
            r.time_renders(10)
        row.append(min(r.time_renders(steps) for _ in range(3)) / steps * 1000.0)
        del r
    print("%-48s %9.1f us %9.1f us %9.1f us" % (name, *row), flush=True)
