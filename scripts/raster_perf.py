"""Device time per render of the tiled raster kernels on the BASELINE shapes (GPU box):
python scripts/raster_perf.py   (used with scripts/ab_run.sh to compare library variants)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from madrona_renderer_amd import scenes  # noqa: E402

os.environ["MRX_PLACEMENT_TRIES"] = "1"
cases = [
    ("headline 4096 x 64^2", scenes.synthetic_scene(4096), 400),
    ("C2 1024 x 64^2", scenes.synthetic_scene(1024), 800),
    ("C4 shard 2048 x 64^2", scenes.synthetic_scene(2048, first_world=14336), 600),
    ("C3 4096 x 128^2 + wall", scenes.synthetic_scene(4096, width=128, height=128, with_wall=True), 100),
    ("C5/8: 512 x 256^2 RT textured", scenes.synthetic_scene(512, width=256, height=256, textured=True,
                                                             render_mode="Raytracer"), 100),
    ("4096 x 64^2 textured + wall", scenes.synthetic_scene(4096, textured=True, with_wall=True), 200),
    ("4096 x 64^2 with visibility ids", None, 400),
]
for name, desc, steps in cases:
    if desc is None:
        os.environ["MADRONA_MI355_VISIBILITY"] = "1"
        desc = scenes.synthetic_scene(4096)
    r = scenes.make_renderer(desc)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        r.time_renders(50)
    best = min(r.time_renders(steps) for _ in range(5)) / steps * 1000.0
    print("%-36s %9.2f us" % (name, best), flush=True)
    del r
    os.environ.pop("MADRONA_MI355_VISIBILITY", None)
