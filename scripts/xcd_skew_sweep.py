"""XCD-skew tuning aid: MRX_XCD_SKEW (0 off, else 1 + log2 period) on a batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from madrona_renderer_amd import scenes
from oracle import oracle

worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kw = {}
if len(sys.argv) > 2:
    kw = dict(width=int(sys.argv[2]), height=int(sys.argv[2]))
desc = scenes.synthetic_scene(worlds, **kw)
o = oracle.FlatScene(desc).render()
for skew in ["0", "1", "2", "3", "4", "0", "1", "2", "3", "4"]:
    os.environ.pop("MRX_XCD_SKEW", None)
    if skew:
        os.environ["MRX_XCD_SKEW"] = skew
    r = scenes.make_renderer(desc)
    r.sync()
    for _ in range(3):
        r.step()
    r.sync()
    rgb = r.rgb_tensor().to_torch().cpu().numpy()
    bad = int((rgb != o["rgb"]).any(axis=-1).sum())
    if os.environ.get("MADRONA_MI355_VISIBILITY") == "1":
        vis = r.visibility_tensor().to_torch().cpu().numpy()
        bad += int((vis != o["tri_id"]).sum())
    r.time_renders(3000)                        # clocks settle
    us = sorted(r.time_renders(400) / 400 * 1000 for _ in range(7))
    print(f"skew {skew or 'auto'}: mismatches {bad}  us/step " + " ".join(f"{u:.2f}" for u in us), flush=True)
    del r
