"""bench.py's timed region at the driver's K = 20, launch by launch: torch events on the null stream (the renderer's)
around every step of a burst that follows a device sync, as bench.py's does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
r = scenes.make_renderer(scenes.synthetic_scene(int(os.environ.get("WORLDS", "4096"))))
t0 = time.time()
while time.time() - t0 < 0.3:
    r.time_renders(100)
K = 20
for burst in range(8):
    for _ in range(5):
        r.step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        r.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    d = [ev[i].elapsed_time(ev[i + 1]) * 1000 for i in range(K)]
    print("burst %d: wall %.2f us/step, device %.2f us/step; per launch: %s" % (burst, wall / K, sum(d) / K, " ".join("%.1f" % x for x in d)), flush=True)
    time.sleep(0.05 * burst)
