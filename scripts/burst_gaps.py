"""Bursts of K launches after a device sync (the shape of bench.py's timed region at the driver's K = 20):
run under `rocprofv3 --kernel-trace` by scripts/burst_gaps.sh, which prints where in a burst the device idles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
r = scenes.make_renderer(scenes.synthetic_scene(int(os.environ.get("WORLDS", "4096"))))
t0 = time.time()
while time.time() - t0 < 0.3:
    r.time_renders(100)
K = int(os.environ.get("K", "40"))
for burst in range(6):
    torch.cuda.synchronize()
    time.sleep(0.002)                 # (a visible gap in the trace between bursts)
    r.mark(0)
    t0 = time.perf_counter()
    for _ in range(K):
        r.step()
    t1 = time.perf_counter()
    r.mark(1)
    torch.cuda.synchronize()
    print("burst %d: host enqueue %.1f us for %d launches, device %.2f us/launch" % (burst, (t1 - t0) * 1e6, K, r.elapsed_ms() * 1000 / K), flush=True)
