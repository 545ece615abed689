"""Where the BVH kernel's time goes: the same scene with phases switched off
(MRX_DEBUG_SKIP: 1 stores, 2 pixel tests, 8 triangle setup, 4 traversal).  Needs a library built
with the kernel's diagnostics:  MRX_EXTRA_HIPCC_FLAGS=-DMRX_BVH_DIAG=1 python -m madrona_renderer_amd.build --force"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from madrona_renderer_amd import scenes  # noqa: E402
from tests import meshes  # noqa: E402

os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
which = sys.argv[1] if len(sys.argv) > 1 else "cubes40"
if len(sys.argv) > 2:
    os.environ["MRX_BVH_SMALL_AREA"] = sys.argv[2]
desc = {"cubes40": lambda: meshes.cube_field(1024, 40),
        "cubes100": lambda: meshes.cube_field(1024, 100),
        "meshes": lambda: meshes.mesh_worlds(256),
        "hl": lambda: scenes.synthetic_scene(4096)}[which]()
for skip, name in ((0, "full"), (1, "no stores"), (32, "no small-triangle walk"), (64, "no large-triangle pass"),
                   (2, "no pixel tests"), (2 | 8, "no setup, no pixel tests"),
                   (4, "no traversal (phase I + background stores)"), (4 | 1, "phase I only"),
                   (16, "bare launch")):
    os.environ["MRX_DEBUG_SKIP"] = str(skip)
    r = scenes.make_renderer(desc)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        r.time_renders(5)
    n = 10 if which == "meshes" else 100
    us = min(r.time_renders(n) for _ in range(3)) * 1000.0 / n
    print("%-46s %8.1f us" % (name, us), flush=True)
    del r
