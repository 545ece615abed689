import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
desc = scenes.synthetic_scene(4096)
for skip in (0, 2, 6, 14, 1, 3, 15, 31):
    os.environ["MRX_DEBUG_SKIP"] = str(skip)
    r = scenes.make_renderer(desc)
    r.time_renders(20)
    ms = sorted(r.time_renders(50) for _ in range(5))
    print(f"skip={skip:2d}: min {ms[0] / 50 * 1000:6.1f}  med {ms[2] / 50 * 1000:6.1f} us/step", flush=True)
    del r
