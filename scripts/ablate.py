"""Timing-only ablation of the raster kernel (MRX_DEBUG_SKIP bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
desc = scenes.synthetic_scene(worlds)
for skip in (0, 1, 2, 4, 8, 3, 7, 15, 6, 14):
    os.environ["MRX_DEBUG_SKIP"] = str(skip)
    r = scenes.make_renderer(desc)
    r.time_renders(20)
    ms = min(r.time_renders(50) for _ in range(3))
    print(f"skip={skip:2d}: {ms / 50 * 1000:7.1f} us/step", flush=True)
    del r
