import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRACE"] = "1"
os.environ["MRX_PLACEMENT_ADDR"] = "1"
os.environ["MRX_PLACEMENT_TRIES"] = sys.argv[2] if len(sys.argv) > 2 else "8"
K = {"C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer")}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
d = scenes.synthetic_scene(**K[name])
for kind in ("split", "one"):
    os.environ["MRX_OUT_KIND"] = kind
    print("== first kind", kind, flush=True)
    r = scenes.make_renderer(d)
    del r
