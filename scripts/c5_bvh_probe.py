"""BASELINE configs[4] through the BVH path (variant 2) under the tile shapes of the sweep and the
TLAS pass sizes: device us per render of N views of 256x256 Raytracer textured cube+plane."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
views = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
desc = scenes.synthetic_scene(views, width=256, height=256, textured=True, render_mode="Raytracer")
for env in ({}, {"MRX_BVH_TILE": "1"}, {"MRX_BVH_TILE": "2"}, {"MRX_BVH_SMALL_AREA": "1024"}, {"MRX_BVH_SMALL_AREA": "4096"},
            {"MRX_BVH_SMALL_AREA": "64"}):
    for k, v in env.items():
        os.environ[k] = v
    r = scenes.make_renderer(desc)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        r.time_renders(10)
    us = min(r.time_renders(30) for _ in range(3)) / 30 * 1000.0
    print("%-32s %9.1f us   (x %d views / 4096 -> %.0f us at 4096 views)" % (env or "default", us, views, us * 4096 / views), flush=True)
    del r
    for k in env:
        del os.environ[k]
