"""One-tile views per workgroup in the BVH kernel (MRX_BVH_GROUP_VIEWS): device us per render under 1, 2, 4
views per group (two: with and without the wave priority of the younger workgroups, MRX_BVH_PRIO), and what the
host picks by itself (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
cases = [("1024 x 64^2, 40 cubes (482 tris)", scenes.cube_field(1024, 40), 200),
         ("1024 x 64^2, 100 cubes (1202 tris)", scenes.cube_field(1024, 100), 100),
         ("1024 x 64^2, 40 textured cubes", scenes.cube_field(1024, 40, textured=True), 100),
         ("1024 x 64^2 RT, 40 textured cubes", scenes.cube_field(1024, 40, textured=True, mode="Raytracer"), 100),
         ("4096 x 64^2, 40 cubes", scenes.cube_field(4096, 40), 50),
         ("4096 x 64^2, cube + plane", scenes.synthetic_scene(4096), 100),
         ("512 x 64^2, 40 cubes", scenes.cube_field(512, 40), 200),
         ("512 x 64^2, 100 cubes", scenes.cube_field(512, 100), 200),
         ("384 x 64^2, 40 textured cubes", scenes.cube_field(384, 40, textured=True), 200),
         ("1024 x 32^2, 40 cubes", scenes.cube_field(1024, 40, width=32, height=32), 200)]
for name, desc, steps in cases:
    row = []
    for g in ("1 prio0", "1", "2 prio0", "2", "4", ""):
        for k in ("MRX_BVH_GROUP_VIEWS", "MRX_BVH_PRIO"):
            os.environ.pop(k, None)
        if g:
            os.environ["MRX_BVH_GROUP_VIEWS"] = g.split()[0]
        if "prio0" in g:
            os.environ["MRX_BVH_PRIO"] = "0"
        r = scenes.make_renderer(desc)
        t0 = time.time()
        while time.time() - t0 < 0.15:
            r.time_renders(10)
        row.append(min(r.time_renders(steps) for _ in range(3)) / steps * 1000.0)
        del r
    print("%-40s " % name + "  ".join("%s %7.1f" % (g, v) for g, v in zip(("v=1 no priority", "v=1", "v=2 no priority", "v=2", "v=4", "host"), row)), flush=True)
