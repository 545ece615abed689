"""Quick GPU-vs-oracle comparison (development aid; the real parity tests
live in tests/)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MADRONA_MI355_VISIBILITY", "1")
import numpy as np
import torch

from madrona_renderer_amd import scenes
from oracle import oracle


def compare(name, desc):
    r = scenes.make_renderer(desc)
    r.sync()
    rgb = r.rgb_tensor().to_torch().cpu().numpy()
    dep = r.depth_tensor().to_torch().cpu().numpy()
    vis = r.visibility_tensor().to_torch().cpu().numpy()
    o = oracle.FlatScene(desc).render()
    dep = dep.reshape(o["depth"].shape)
    vis_bad = int((vis != o["tri_id"]).sum())
    rgb_bad = int((rgb != o["rgb"]).any(axis=-1).sum())
    dmax = float(np.abs(dep - o["depth"]).max())
    dexact = bool(np.array_equal(dep, o["depth"]))
    print(f"{name}: views={rgb.shape[0]} vis_mismatch={vis_bad} rgb_mismatch={rgb_bad} "
          f"depth_maxabs={dmax:.3g} depth_bitexact={dexact} "
          f"covered={(o['tri_id'] >= 0).mean():.3f}", flush=True)
    return r


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    compare("demo-raster", scenes.demo_scene(render_mode="Rasterizer"))
    compare("demo-rt", scenes.demo_scene(render_mode="Raytracer"))
    compare("syn64", scenes.synthetic_scene(64))
    compare("syn64-tex-wall", scenes.synthetic_scene(64, with_wall=True, textured=True))
    compare("syn16-128", scenes.synthetic_scene(16, width=128, height=128, with_wall=True))
    compare("syn8-96x40", scenes.synthetic_scene(8, width=96, height=40))
    r = compare("syn4096", scenes.synthetic_scene(4096))
    for _ in range(3):
        ms = r.time_renders(50)
        print(f"4096x64x64: {ms / 50 * 1000:.1f} us/step, "
              f"{4096 / (ms / 50 / 1000):.3e} views/s, "
              f"{4096 * 64 * 64 * 8 / (ms / 50 / 1000) / 1e12:.3f} TB/s", flush=True)
