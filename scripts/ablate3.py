import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
desc = scenes.synthetic_scene(4096)
for slots in (16, 32, 64):
    os.environ["MRX_DEBUG_SLOTS"] = str(slots)
    row = []
    for skip in (15, 1, 0):
        os.environ["MRX_DEBUG_SKIP"] = str(skip)
        r = scenes.make_renderer(desc)
        r.time_renders(20)
        ms = min(r.time_renders(50) for _ in range(3))
        row.append(f"skip{skip}={ms / 50 * 1000:6.1f}")
        del r
    print(f"slots={slots}: " + "  ".join(row), flush=True)
