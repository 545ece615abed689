"""One block, MRX_PLACEMENT_TRIES=1: render time against the distance between
the rgb and the depth tensor at the MiB scale (depth phase = X MiB + 256 KiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
os.environ["MRX_OUT_KIND"] = "one"
K = {"HL": dict(num_worlds=4096), "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
     "C4": dict(num_worlds=2048), "C2": dict(num_worlds=1024)}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
xs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(0, 257, 16))
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 256
d = scenes.synthetic_scene(**K[name])
n = 40 if name == "C5" else 300
out = []
for x in xs:
    os.environ["MRX_OUT_SKEW_DEPTH_KB"] = str(x * 1024 + sub)
    r = scenes.make_renderer(d)
    r.time_renders(3 * n)
    us = min(r.time_renders(n) for _ in range(3)) / n * 1000
    out.append(f"{x}:{us:.2f}")
    del r
print(name, "sub-phase", sub, "KiB; extra MiB: us  ", " ".join(out), flush=True)
