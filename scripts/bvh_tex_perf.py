import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from madrona_renderer_amd import scenes
from tests import meshes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
def timed(desc, variant, steps):
    os.environ["MADRONA_MI355_KERNEL"] = str(variant)
    r = scenes.make_renderer(desc)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        r.time_renders(20)
    best = min(r.time_renders(steps) for _ in range(3)) / steps * 1000.0
    del r
    return best
for name, desc, steps in [
    ("1024 x 64^2, 40 textured cubes", meshes.cube_field(1024, 40, textured=True), 200),
    ("1024 x 64^2, 100 textured cubes", meshes.cube_field(1024, 100, textured=True), 100),
    ("256 x 128^2 RT, 100 textured cubes", meshes.cube_field(256, 100, width=128, height=128, mode="Raytracer", textured=True), 50),
    ("512 x 256^2 RT textured cube+plane", scenes.synthetic_scene(512, width=256, height=256, textured=True, render_mode="Raytracer"), 30),
    ("1024 x 64^2, 40 cubes", meshes.cube_field(1024, 40), 200)]:
    print("%-40s bvh %9.2f us" % (name, timed(desc, 2, steps)), flush=True)
