"""Tiles of a view per workgroup in the BVH kernel (MRX_BVH_GROUP_TILES): device us per render of multi-tile
shapes under 1, 2, 4, 8, 16 tiles per group (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
cases = [("512 x 256^2 RT textured cube+plane (C5 / 8)", scenes.synthetic_scene(512, width=256, height=256, textured=True, render_mode="Raytracer"), 30),
         ("256 x 128^2 RT, 100 cubes", scenes.cube_field(256, 100, width=128, height=128, mode="Raytracer"), 50),
         ("64 x 256^2 RT, 100 cubes", scenes.cube_field(64, 100, width=256, height=256, mode="Raytracer"), 50),
         ("1024 x 128^2 cube+plane+wall", scenes.synthetic_scene(1024, width=128, height=128, with_wall=True), 50),
         ("256 x 128^2, 40 textured cubes", scenes.cube_field(256, 40, width=128, height=128, textured=True), 50)]
for name, desc, steps in cases:
    row = []
    for g in ("1", "2", "4", "8", "16"):
        os.environ["MRX_BVH_GROUP_TILES"] = g
        r = scenes.make_renderer(desc)
        t0 = time.time()
        while time.time() - t0 < 0.15:
            r.time_renders(10)
        row.append(min(r.time_renders(steps) for _ in range(3)) / steps * 1000.0)
        del r
    print("%-46s " % name + "  ".join("g=%s %8.1f" % (g, v) for g, v in zip((1, 2, 4, 8, 16), row)), flush=True)
