"""Times the BASELINE.json configurations on one GPU (render only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_renderer_amd import scenes
CONFIGS = [
    ("C2  1024 x 64^2  cube+plane", dict(num_worlds=1024)),
    ("HL  4096 x 64^2  cube+plane", dict(num_worlds=4096)),
    ("C4  2048 x 64^2  (per GPU of 8)", dict(num_worlds=2048)),
    ("C3  4096 x 128^2 +wall", dict(num_worlds=4096, width=128, height=128, with_wall=True)),
    ("C5  4096 x 256^2 RT textured", dict(num_worlds=4096, width=256, height=256, textured=True,
                                          render_mode="Raytracer")),
    ("    4096 x 64^2  textured +wall", dict(num_worlds=4096, textured=True, with_wall=True)),
]
for name, kw in CONFIGS:
    t0 = time.time()
    d = scenes.synthetic_scene(**kw)
    r = scenes.make_renderer(d)
    t1 = time.time()
    while time.time() - t1 < 0.25:               # let the clocks settle
        r.time_renders(50)
    n = 400 if d.num_views * d.width * d.height < 2 ** 27 else 50
    ms = sorted(r.time_renders(n) for _ in range(5))[2] / n
    b = r.bytes_per_step()
    print(f"{name:34s} {ms * 1000:9.1f} us/step  {d.num_views / ms * 1000:11.3e} views/s  "
          f"{b / ms / 1e9:6.2f} TB/s  ({b / 2**20:7.0f} MiB/step, setup {time.time() - t0:.1f}s)", flush=True)
    del r
