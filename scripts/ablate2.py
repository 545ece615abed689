import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
for worlds in (64, 256, 1024, 2048, 4096, 8192):
    desc = scenes.synthetic_scene(worlds)
    row = []
    for skip in (15, 14, 1, 0):
        os.environ["MRX_DEBUG_SKIP"] = str(skip)
        r = scenes.make_renderer(desc)
        r.time_renders(20)
        ms = min(r.time_renders(50) for _ in range(3))
        row.append(f"skip{skip}={ms / 50 * 1000:6.1f}")
        del r
    print(f"worlds={worlds:5d}: " + "  ".join(row) + " us/step", flush=True)
