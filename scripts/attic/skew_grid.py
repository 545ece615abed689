"""Tuning aid: output tensor skews (MRX_OUT_SKEW_DEPTH_KB x MRX_OUT_SKEW_IDS_KB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
ids = len(sys.argv) > 1 and sys.argv[1] == "ids"
if ids:
    os.environ["MADRONA_MI355_VISIBILITY"] = "1"
d = scenes.synthetic_scene(4096)
for dk in ([0, 128, 256, 384] if ids else [0, 64, 128, 192, 256, 320, 384, 448]):
    row = []
    for ik in ([0, 64, 128, 192, 256, 320, 384, 448] if ids else [0]):
        os.environ["MRX_OUT_SKEW_DEPTH_KB"] = str(dk)
        os.environ["MRX_OUT_SKEW_IDS_KB"] = str(ik)
        r = scenes.make_renderer(d)
        r.time_renders(2000)
        row.append(min(r.time_renders(400) / 400 * 1000 for _ in range(3)))
        del r
    print(f"depth +{dk:3d} KiB:", " ".join(f"{x:.2f}" for x in row), flush=True)
