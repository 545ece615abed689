"""Timing of worlds with more than 64 triangles (chunked kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madrona_renderer_amd import scenes

def many_cubes(num_worlds, cubes, width=64, height=64):
    d = scenes.synthetic_scene(num_worlds, width=width, height=height)
    rng = np.random.default_rng(5)
    inst, worlds = [], []
    base = d.instances
    for w in range(num_worlds):
        inst.append(base[2 * w])          # plane
        for c in range(cubes):
            p = rng.uniform(-6, 6, 2)
            s = float(rng.uniform(0.5, 1.5))
            inst.append(((float(p[0]), float(p[1]), 0.5 * s), (1.0, 0.0, 0.0, 0.0), (s, s, s), 0))
        worlds.append((1 + cubes, w * (1 + cubes), 1, w))
    d.instances, d.worlds = inst, worlds
    return d

for cubes in (4, 5, 10, 40):
    d = many_cubes(1024, cubes)
    r = scenes.make_renderer(d)
    r.time_renders(3)
    ms = min(r.time_renders(10) for _ in range(3)) / 10
    print(f"1024 worlds x (plane + {cubes:3d} cubes = {2 + 12 * cubes:4d} tris): {ms * 1000:9.1f} us/step "
          f"{1024 / ms * 1000:10.3e} views/s", flush=True)
    del r
