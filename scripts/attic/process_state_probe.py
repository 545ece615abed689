"""What in a long-lived process slows the headline render from 22.5 to 24.9 us?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
from tests import meshes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
hl = scenes.synthetic_scene(4096)


def t(label):
    r = scenes.make_renderer(hl)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        r.time_renders(50)
    us = min(r.time_renders(400) for _ in range(5)) / 400 * 1000
    print(f"{label:60s} {us:7.2f} us", flush=True)
    del r


t("fresh")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    x = torch.randn(2048, 2048, device="cuda")
    x = x @ x
torch.cuda.synchronize()
t("after a torch side stream did some work")
os.environ["MADRONA_MI355_KERNEL"] = "2"
d = meshes.cube_field(8, 40, textured=True, mode="Raytracer")
r = scenes.make_renderer(d); r.step(); r.sync(); del r
os.environ.pop("MADRONA_MI355_KERNEL")
t("after a kernel that uses scratch ran (BVH, textured)")
os.environ["MADRONA_MI355_KERNEL"] = "3"
d = meshes.cube_field(8, 40)
r = scenes.make_renderer(d); r.step(); r.sync(); del r
os.environ.pop("MADRONA_MI355_KERNEL")
t("after the chunked raster kernel ran")
r2 = scenes.make_renderer(scenes.synthetic_scene(64)); s2 = torch.cuda.Stream(); r2.set_stream(s2.cuda_stream); r2.step(); r2.sync(); del r2
t("after a renderer was moved to a side stream")
y = torch.empty(3 * 2**30, dtype=torch.uint8, device="cuda"); z = y.cpu(); del y, z; torch.cuda.empty_cache()
t("after a 3 GiB device->host copy")
