import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
base = scenes.synthetic_scene(4096)
def timed(desc, label):
    row = []
    for skip in (1, 0):
        os.environ["MRX_DEBUG_SKIP"] = str(skip)
        r = scenes.make_renderer(desc)
        r.time_renders(20)
        ms = min(r.time_renders(50) for _ in range(3))
        row.append(f"skip{skip}={ms / 50 * 1000:6.1f}")
        del r
    print(f"{label:28s}: " + "  ".join(row), flush=True)
timed(base, "4096 random worlds")
for w in (0, 5, 17):
    d = scenes.synthetic_scene(4096)
    d.instances = base.instances[2 * w:2 * w + 2] * 4096
    d.cameras = [base.cameras[w]] * 4096
    timed(d, f"4096 copies of world {w}")
