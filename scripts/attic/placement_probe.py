"""Which ingredient of the placement search makes a candidate fast?  C3 by default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
K = {"HL": dict(num_worlds=4096), "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer")}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
d = scenes.synthetic_scene(**K[name])
n = 40 if name == "C5" else 300
os.environ["MRX_PLACEMENT_TRACE"] = "1"


def run(label, **env):
    for k in ("MRX_PLACEMENT_TRIES", "MRX_OUT_KIND", "MRX_OUT_PRE_MB", "MRX_OUT_PRE_HOLD"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = v
    r = scenes.make_renderer(d)
    r.time_renders(4 * n)
    us = min(r.time_renders(n) for _ in range(3)) / n * 1000
    print(f"{label:50s} {us:8.2f} us  rgb@{r.rgb_cuda_ptr():#x}", flush=True)
    return r


for rep in range(2):
    run("default search")
    run("tries=2")
    run("tries=2 first kind one", MRX_PLACEMENT_TRIES="2", MRX_OUT_KIND="one")
    run("tries=2 first kind split", MRX_PLACEMENT_TRIES="2", MRX_OUT_KIND="split")
    run("tries=1", MRX_PLACEMENT_TRIES="1")
# two renderers alive at once: is the second one's placement fast?
a = run("tries=1, first of two alive", MRX_PLACEMENT_TRIES="1")
b = run("tries=1, second of two alive", MRX_PLACEMENT_TRIES="1")
c = run("tries=1, third of three alive", MRX_PLACEMENT_TRIES="1")
us = min(a.time_renders(n) for _ in range(3)) / n * 1000
print("first again: %.2f" % us)
del a
us = min(b.time_renders(n) for _ in range(3)) / n * 1000
print("second after the first was destroyed: %.2f" % us)
