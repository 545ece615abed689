"""From how many one-tile views on do two views per workgroup pay?  Device us per render of 40-cube worlds through the
BVH kernel under MRX_BVH_GROUP_VIEWS = 1 / 2 and as the host picks, by view count (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for views in (384, 512, 520, 576, 640, 768, 896, 1000, 1024, 1280, 1536, 2048, 3072):
    desc = scenes.cube_field(views, cubes)
    row = []
    for g in ("1", "2 pure", ""):
        os.environ.pop("MRX_BVH_GROUP_VIEWS", None)
        os.environ.pop("MRX_BVH_NO_MIXED", None)
        if g:
            os.environ["MRX_BVH_GROUP_VIEWS"] = g.split()[0]
        if "pure" in g:
            os.environ["MRX_BVH_NO_MIXED"] = "1"
        r = scenes.make_renderer(desc)
        t0 = time.time()
        while time.time() - t0 < 0.1:
            r.time_renders(10)
        row.append(min(r.time_renders(200) for _ in range(3)) / 200 * 1000.0)
        del r
    print("%5d views x %d cubes: one view per workgroup %6.1f   two (all pairs) %6.1f   host %6.1f" % (views, cubes, *row), flush=True)
