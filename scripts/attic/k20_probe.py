"""Driver-style short timed regions (K = 20 after a sync): per-group device
time, to see what a short region pays over the steady state (start-up after
the sync, the XCC-report variant every 32nd launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
hl = scenes.synthetic_scene(int(os.environ.get("WORLDS", "4096")))
r = scenes.make_renderer(hl)
t0 = time.time()
while time.time() - t0 < 0.3:
    r.time_renders(100)
for K in (20, 32, 64, 640):
    out = []
    for g in range(24):
        torch.cuda.synchronize()
        r.mark(0)
        for _ in range(K):
            r.step()
        r.mark(1)
        torch.cuda.synchronize()
        out.append(r.elapsed_ms() * 1000 / K)
    print("K=%4d us/step:" % K, " ".join("%.2f" % x for x in out), flush=True)
