"""Null-stream launches vs other streams in the process: headline us/render."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
hl = scenes.synthetic_scene(4096)
r = scenes.make_renderer(hl)


def t(label, rr=None):
    rr = rr or r
    t0 = time.time()
    while time.time() - t0 < 0.2:
        rr.time_renders(50)
    us = min(rr.time_renders(400) for _ in range(5)) / 400 * 1000
    print(f"{label:70s} {us:7.2f} us", flush=True)


t("fresh (renderer on the null stream)")
streams = []
for i in range(5):
    s = torch.cuda.Stream()
    streams.append(s)
    t(f"{i + 1} torch stream(s) created, idle")
    with torch.cuda.stream(s):
        x = torch.ones(1024, device="cuda") * 2
    torch.cuda.synchronize()
    t(f"{i + 1} torch stream(s), the newest did some work")
own = torch.cuda.Stream()
r.set_stream(own.cuda_stream)
t("the renderer on a torch side stream of its own")
r.set_stream(0)
t("back on the null stream")
del streams, s, x
gc.collect()
torch.cuda.synchronize()
t("torch stream objects dropped")
