"""Textured cube fields through the BVH kernel (1024 x 64x64), device us per render, by records per round
(MRX_BVH_TEX_CAP; default = what the host derives from the LDS budget)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
for cap in ("256", "384", "512", None):
    if cap is None:
        os.environ.pop("MRX_BVH_TEX_CAP", None)
    else:
        os.environ["MRX_BVH_TEX_CAP"] = cap
    out = []
    for cubes in (40, 60, 80, 100, 140, 200):
        r = scenes.make_renderer(scenes.cube_field(1024, cubes, textured=True))
        t0 = time.time()
        while time.time() - t0 < 0.15:
            r.time_renders(20)
        out.append("%d: %.2f" % (12 * cubes + 2, min(r.time_renders(200) for _ in range(3)) / 200 * 1000.0))
        del r
    print("cap %-8s textured, us by triangles per world:  " % (cap or "default") + "   ".join(out), flush=True)
