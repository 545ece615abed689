"""With MRX_PLACEMENT_TRIES=1: how the layout of the output tensors (one block
vs one block per tensor, phase of depth, gap) decides the render time."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
K = {"HL": dict(num_worlds=4096), "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer")}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["HL", "C3", "C5"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
settings = [("one", None, None), ("one", "0", None), ("one", "64", None), ("one", "128", None), ("one", "384", None),
            ("split", None, None), ("split", "0", None), ("split", None, "2"), ("split", None, "34"),
            ("split", "0", "34"), ("split", "128", None)]
for name in names:
    d = scenes.synthetic_scene(**K[name])
    n = 40 if name == "C5" else 300
    for kind, skew, gap in settings:
        os.environ["MRX_OUT_KIND"] = kind
        for k, v in (("MRX_OUT_SKEW_DEPTH_KB", skew), ("MRX_OUT_GAP_MB", gap)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        res = []
        for _ in range(reps):
            r = scenes.make_renderer(d)
            r.time_renders(4 * n)
            res.append(min(r.time_renders(n) for _ in range(3)) / n * 1000)
            del r
        print(f"{name} kind={kind:5s} depth_phase_kb={str(skew):4s} gap_mb={str(gap):4s}  " +
              " ".join(f"{u:8.2f}" for u in res), flush=True)
