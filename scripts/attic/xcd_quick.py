import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
hl = scenes.synthetic_scene(4096)
for label, env in (("default", {}), ("trust partner (FAKE=5)", {"MRX_XCD_FAKE": "5"}), ("skew off", {"MRX_XCD_SKEW": "0"}),
                   ("both fast (FAKE=3)", {"MRX_XCD_FAKE": "3"}), ("both slow (FAKE=2)", {"MRX_XCD_FAKE": "2"}),
                   ("by index (FAKE=4)", {"MRX_XCD_FAKE": "4"})):
    for k in ("MRX_XCD_FAKE", "MRX_XCD_SKEW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = scenes.make_renderer(hl)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        r.time_renders(50)
    print(f"{label:28s} {min(r.time_renders(400) for _ in range(5)) / 400 * 1000:7.2f} us", flush=True)
    del r
