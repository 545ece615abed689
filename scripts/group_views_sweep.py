import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
for worlds in (512, 1024, 2048):
    d = scenes.synthetic_scene(worlds)
    for gv in ("default", "1", "2", "4"):
        os.environ.pop("MRX_GROUP_VIEWS", None)
        if gv != "default":
            os.environ["MRX_GROUP_VIEWS"] = gv
        r = scenes.make_renderer(d)
        t0 = time.time()
        while time.time() - t0 < 0.3:
            r.time_renders(50)
        us = min(r.time_renders(800) for _ in range(5)) / 800 * 1000
        print("%5d worlds, views per group %-8s %7.2f us" % (worlds, gv, us), flush=True)
        del r
