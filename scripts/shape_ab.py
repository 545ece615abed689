"""Tuning aid: A/B of two workgroup shapes over several renderer instances each
(the output placement differs from instance to instance)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
K = {
    "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
    "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
    "U256": dict(num_worlds=4096, width=256, height=256),
    "T128": dict(num_worlds=4096, width=128, height=128, textured=True),
    "U512": dict(num_worlds=1024, width=512, height=512),
    "T256s": dict(num_worlds=1024, width=256, height=256, textured=True),
    "C2": dict(num_worlds=1024),
    "C4": dict(num_worlds=2048),
    "W512": dict(num_worlds=512),
    "T64": dict(num_worlds=4096, textured=True),
    "TW64": dict(num_worlds=4096, textured=True, with_wall=True),
    "U128s": dict(num_worlds=1024, width=128, height=128),
}
name = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[2:]]
desc = scenes.synthetic_scene(**K[name])
n = 300 if desc.num_worlds * desc.width * desc.height < 2 ** 27 else 60
for rep in range(4):
    for env in variants:
        for k in ("MRX_GROUP_VIEWS", "MRX_GROUP_TILES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = scenes.make_renderer(desc)
        r.time_renders(3 * n)
        us = min(r.time_renders(n) / n * 1000 for _ in range(3))
        print(f"{name} {env or 'auto'}: {us:.1f}", flush=True)
        del r
