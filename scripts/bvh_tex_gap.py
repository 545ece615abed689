"""Textured against untextured cube fields through the BVH kernel, by cubes per world: where the textured
instantiation's extra time comes from (rounds of its 255-record table, or per-pixel texturing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
for cubes in (5, 10, 20, 30, 40, 60, 100):
    row = []
    for tex in (False, True):
        r = scenes.make_renderer(scenes.cube_field(1024, cubes, textured=tex))
        t0 = time.time()
        while time.time() - t0 < 0.15:
            r.time_renders(20)
        row.append(min(r.time_renders(200) for _ in range(3)) / 200 * 1000.0)
        del r
    print("%3d cubes (%4d triangles): untextured %7.2f us   textured %7.2f us   ratio %.2f" % (cubes, 12 * cubes + 2, row[0], row[1], row[1] / row[0]), flush=True)
