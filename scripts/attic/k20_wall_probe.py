"""What the wall clock of a K = 20 timed region pays beyond the kernels (driver-style bench):
variants of the bracket, 40 repetitions each, median us per step."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
r = scenes.make_renderer(scenes.synthetic_scene(4096))
t0 = time.time()
while time.time() - t0 < 0.3:
    r.time_renders(100)
K = 20
def run(marks, presync):
    out = []
    for _ in range(40):
        for _ in range(5):
            r.step()
        torch.cuda.synchronize()
        a = time.perf_counter()
        if marks: r.mark(0)
        for _ in range(K):
            r.step()
        if marks: r.mark(1)
        if presync: r.sync()
        torch.cuda.synchronize()
        b = time.perf_counter()
        out.append((b - a) * 1e6 / K)
    return statistics.median(out), min(out)
for name, m, p in (("marks + torch sync (bench.py)", True, False), ("marks + stream sync first", True, True),
                   ("no marks, torch sync", False, False), ("no marks, stream sync first", False, True)):
    med, lo = run(m, p)
    print("%-34s median %.2f  min %.2f us/step" % (name, med, lo), flush=True)
