"""Diagnostic: which output allocations stream fast (MRX_PLACEMENT_TRACE=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
K = {"C2": dict(num_worlds=1024), "HL": dict(num_worlds=4096), "C4": dict(num_worlds=2048),
     "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
     "TW": dict(num_worlds=4096, textured=True, with_wall=True)}
keep = []
for name in sys.argv[1:]:
    hold = name.endswith("+")
    name = name.rstrip("+")
    d = scenes.synthetic_scene(**K[name])
    r = scenes.make_renderer(d)
    n = 60 if name == "C5" else 400
    r.time_renders(5 * n)
    us = sorted(r.time_renders(n) / n * 1000 for _ in range(3))
    print(name, " ".join(f"{u:.2f}" for u in us), flush=True)
    if hold:
        keep.append(r)
    del r
