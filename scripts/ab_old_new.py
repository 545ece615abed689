"""A/B of two builds in one gpurun call: python scripts/ab_old_new.py <root>  (imports madrona_renderer_amd from <root>)"""
import os, sys, time
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
for name, kw, n in (("HL", dict(num_worlds=4096), 400), ("C2", dict(num_worlds=1024), 800), ("C4", dict(num_worlds=2048), 600),
                    ("TW", dict(num_worlds=4096, textured=True, with_wall=True), 300)):
    r = scenes.make_renderer(scenes.synthetic_scene(**kw))
    t0 = time.time()
    while time.time() - t0 < 0.3:
        r.time_renders(50)
    print(f"{os.path.basename(root) or 'new':8s} {name}: " + " ".join(f"{r.time_renders(n) / n * 1000:6.2f}" for _ in range(5)), flush=True)
    del r
