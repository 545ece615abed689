"""Headline batch after an allocation history: create / destroy a few large
renderers first (as a long test session does), then time fresh headline
renderers with and without the placement search."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRACE"] = "1"
hl = scenes.synthetic_scene(4096)


def t(label, tries):
    if tries is None:
        os.environ.pop("MRX_PLACEMENT_TRIES", None)
    else:
        os.environ["MRX_PLACEMENT_TRIES"] = tries
    r = scenes.make_renderer(hl)
    r.time_renders(2000)
    us = min(r.time_renders(400) for _ in range(3)) / 400 * 1000
    print(f"{label:40s} {us:7.2f} us", flush=True)
    del r


t("fresh process, one try", "1")
t("fresh process, search", None)
x = [torch.empty(int(g * 2**30), dtype=torch.uint8, device="cuda") for g in (0.3, 1.1, 2.5)]
big = scenes.make_renderer(scenes.synthetic_scene(2048, width=256, height=256, textured=True, render_mode="Raytracer"))
del big
del x[1]
mid = scenes.make_renderer(scenes.synthetic_scene(4096, width=128, height=128, with_wall=True))
del mid
torch.cuda.empty_cache()
for i in range(4):
    t("after history, one try", "1")
    t("after history, search", None)
