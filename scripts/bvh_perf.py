"""Device time per render of the BVH path against the tiled raster kernels on
many-instance worlds (GPU box):  python scripts/bvh_perf.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from madrona_renderer_amd import scenes  # noqa: E402
from tests import meshes  # noqa: E402


def timed(desc, variant, steps):
    os.environ["MADRONA_MI355_KERNEL"] = str(variant)
    os.environ["MRX_PLACEMENT_TRIES"] = "1"
    r = scenes.make_renderer(desc)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        r.time_renders(20)
    best = min(r.time_renders(steps) for _ in range(3)) / steps * 1000.0
    del r
    return best


def main():
    cases = [
        ("1024 x 64^2, 40 cubes (482 tris)", meshes.cube_field(1024, 40), 200),
        ("1024 x 64^2, 100 cubes (1202 tris)", meshes.cube_field(1024, 100), 100),
        ("256 x 64^2, 416 cubes (4994 tris)", meshes.cube_field(256, 416), 50),
        ("256 x 128^2 RT, 100 cubes", meshes.cube_field(256, 100, width=128, height=128, mode="Raytracer"), 50),
        ("1024 x 64^2, cube+plane (14 tris)", scenes.synthetic_scene(1024), 500),
        ("4096 x 64^2, cube+plane (14 tris)", scenes.synthetic_scene(4096), 200),
        ("512 x 256^2 RT textured cube+plane", scenes.synthetic_scene(512, width=256, height=256, textured=True,
                                                                     render_mode="Raytracer"), 50),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cases = cases[:2]
    if len(sys.argv) > 1 and sys.argv[1] == "crossover":
        cases = [("1024 x 64^2, %d cubes (%d tris)" % (c, 12 * c + 2), meshes.cube_field(1024, c), 200)
                 for c in ((10, 11, 12, 13, 14) if len(sys.argv) > 2 and sys.argv[2] == "fine" else (2, 5, 10, 15, 21, 30, 40))]
    if len(sys.argv) > 1 and sys.argv[1] == "bvhonly":
        cases = cases[:4] + [("1024 x 64^2, 40 textured cubes", meshes.cube_field(1024, 40, textured=True), 200),
                             ("64 x 256^2 RT, 416 cubes", meshes.cube_field(64, 416, width=256, height=256, mode="Raytracer"), 50),
                             ("256 x 128^2, 40 cubes", meshes.cube_field(256, 40, width=128, height=128), 100),
                             ("256 x 64^2, meshes (14152 tris, BLAS)", meshes.mesh_worlds(256), 50),
                             ("64 x 256^2 RT, meshes (BLAS)", meshes.mesh_worlds(64, 256, 256, "Raytracer"), 30),
                             # close-ups: every triangle is large
                             ("1024 x 64^2, cube+plane (14 tris)", scenes.synthetic_scene(1024), 200),
                             ("1024 x 128^2, cube+plane+wall", scenes.synthetic_scene(1024, width=128, height=128, with_wall=True), 50)]
        for name, desc, steps in cases:
            print("%-44s bvh %9.2f us" % (name, timed(desc, 2, steps)), flush=True)
        return
    for name, desc, steps in cases:
        row = []
        for variant in (2, 3):
            row.append(timed(desc, variant, steps))
        print("%-44s bvh %9.1f us   raster %9.1f us   ratio %.2f" % (name, row[0], row[1], row[1] / row[0]), flush=True)


if __name__ == "__main__":
    main()
