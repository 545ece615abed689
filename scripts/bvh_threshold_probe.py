"""Where should the default dispatch hand a world to the BVH path?  BVH kernel against the tiled raster kernels on worlds of
62 ... 134 triangles (5 ... 11 cubes + plane), by view count; TEXTURED=1 for textured cubes (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
for views in ((256, 512, 1024) if os.environ.get("TEXTURED") == "1" else (256, 512, 1024, 2048, 4096)):
    for cubes in (5, 6, 8, 10, 11):
        desc = scenes.cube_field(views, cubes, textured=os.environ.get("TEXTURED") == "1")
        row = []
        for variant in ("2", "3"):
            os.environ["MADRONA_MI355_KERNEL"] = variant
            r = scenes.make_renderer(desc)
            t0 = time.time()
            while time.time() - t0 < 0.1:
                r.time_renders(20)
            row.append(min(r.time_renders(200) for _ in range(3)) / 200 * 1000.0)
            del r
        print("%5d views x %3d triangles: bvh %6.1f us   raster %6.1f us   raster / bvh %.2f" % (views, 12 * cubes + 2, row[0], row[1], row[1] / row[0]), flush=True)
