"""With MRX_PLACEMENT_TRIES=1: does a block allocated ahead of the output
tensors (and freed again) change where they land and how fast they stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
K = {"HL": dict(num_worlds=4096), "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer")}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C3", "C5"]
for name in names:
    d = scenes.synthetic_scene(**K[name])
    n = 40 if name == "C5" else 300
    for kind in ("one", "split"):
        os.environ["MRX_OUT_KIND"] = kind
        for pre in (None, "2", "64", "256", "600", "1024", "3200", "8192"):
            for hold in ("0", "1"):
                if pre is None and hold == "1":
                    continue
                os.environ.pop("MRX_OUT_PRE_MB", None)
                if pre is not None:
                    os.environ["MRX_OUT_PRE_MB"] = pre
                os.environ["MRX_OUT_PRE_HOLD"] = hold
                res = []
                for _ in range(2):
                    r = scenes.make_renderer(d)
                    r.time_renders(4 * n)
                    res.append((min(r.time_renders(n) for _ in range(3)) / n * 1000, r.rgb_cuda_ptr(), r.depth_cuda_ptr()))
                    del r
                print(f"{name} kind={kind:5s} pre_mb={str(pre):5s} hold={hold}  " +
                      "  ".join(f"{u:8.2f} rgb@{a:#x} depth@{b:#x}" for u, a, b in res), flush=True)
