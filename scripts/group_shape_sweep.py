"""Tuning aid: workgroup shape of the group kernel (MRX_GROUP_VIEWS whole views
per workgroup, or MRX_GROUP_TILES tiles of one view) on the larger configs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madrona_renderer_amd import scenes

CONFIGS = {
    "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
    "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
    "TW": dict(num_worlds=4096, with_wall=True, textured=True),
    "HL": dict(num_worlds=4096),
    "U256": dict(num_worlds=4096, width=256, height=256),
    "T128": dict(num_worlds=4096, width=128, height=128, textured=True),
    "TW128": dict(num_worlds=4096, width=128, height=128, textured=True, with_wall=True),
    "U512": dict(num_worlds=1024, width=512, height=512),
    "T256s": dict(num_worlds=1024, width=256, height=256, textured=True),
}
which = sys.argv[1:] or ["C3", "C5"]
for name in which:
    desc = scenes.synthetic_scene(**CONFIGS[name])
    for env in [{}, {"MRX_GROUP_VIEWS": "1"}, {"MRX_GROUP_VIEWS": "2"}, {"MRX_GROUP_VIEWS": "4"},
                {"MRX_GROUP_TILES": "1"}, {"MRX_GROUP_TILES": "2"}, {"MRX_GROUP_TILES": "4"},
                {"MRX_GROUP_TILES": "8"}, {"MRX_GROUP_TILES": "16"}]:
        for k in ("MRX_GROUP_VIEWS", "MRX_GROUP_TILES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = scenes.make_renderer(desc)
        r.sync()
        n = 300 if desc.num_worlds * desc.width * desc.height < 2 ** 27 else 60
        r.time_renders(3 * n)
        us = sorted(r.time_renders(n) / n * 1000 for _ in range(3))
        print(f"{name} {env or 'auto'}: us/step " + " ".join(f"{u:.1f}" for u in us), flush=True)
        del r
