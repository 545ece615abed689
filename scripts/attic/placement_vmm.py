"""Does the physical backing decide the fast / slow mode of 256 MiB+ outputs?  Fresh renderers of C3
(4096 x 128^2 + wall) and C5/2 (2048 x 256^2 RT textured), one try each, alternating hipMalloc
and the virtual-memory API (MRX_OUT_ALLOC=vmm, one physical handle), granularities 2 MiB ... 1 GiB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
os.environ["MRX_OUT_KIND"] = "one"
cases = [("C3", scenes.synthetic_scene(4096, width=128, height=128, with_wall=True), 60),
         ("C5/2", scenes.synthetic_scene(2048, width=256, height=256, textured=True, render_mode="Raytracer"), 20)]
for name, d, steps in cases:
    for rep in range(4):
        for how in ("malloc", "vmm", "vmm:64", "vmm:1024"):
            os.environ.pop("MRX_OUT_ALLOC", None); os.environ.pop("MRX_OUT_VMM_GRAN_MB", None)
            if how.startswith("vmm"):
                os.environ["MRX_OUT_ALLOC"] = "vmm"
                if ":" in how:
                    os.environ["MRX_OUT_VMM_GRAN_MB"] = how.split(":")[1]
            r = scenes.make_renderer(d)
            r.time_renders(steps)
            us = min(r.time_renders(steps) for _ in range(3)) / steps * 1000
            print("%-5s rep %d %-9s %8.1f us" % (name, rep, how, us), flush=True)
            del r
