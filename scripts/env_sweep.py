"""Tuning aid: time one config under several values of one MRX_* variable.
   python scripts/env_sweep.py C5 MRX_OUT_SKEW 0 4096 65536 ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madrona_renderer_amd import scenes

CONFIGS = {
    "C4": dict(num_worlds=2048),
    "C2": dict(num_worlds=1024),
    "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
    "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
    "TW": dict(num_worlds=4096, with_wall=True, textured=True),
    "HL": dict(num_worlds=4096),
}
name, var, values = sys.argv[1], sys.argv[2], sys.argv[3:]
desc = scenes.synthetic_scene(**CONFIGS[name])
for rep in range(2):
    for v in values:
        os.environ[var] = v
        r = scenes.make_renderer(desc)
        r.sync()
        n = 400 if name != "C5" else 60
        r.time_renders(5 * n)
        us = sorted(r.time_renders(n) / n * 1000 for _ in range(3))
        print(f"{name} {var}={v}: us/step " + " ".join(f"{u:.2f}" for u in us), flush=True)
        del r
