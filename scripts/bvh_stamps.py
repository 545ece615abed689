"""In-kernel timeline of the BVH kernel (MRX_DEBUG_STAMPS=1): per-phase
durations of the workgroups, first and second generation.  Needs a library built with the
kernel's diagnostics:  MRX_EXTRA_HIPCC_FLAGS=-DMRX_BVH_DIAG=1 python -m madrona_renderer_amd.build --force"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_DEBUG_STAMPS"] = "1"
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
import numpy as np
import torch  # noqa
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
from tests import meshes
cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if cubes > 0:
    desc = meshes.cube_field(worlds, cubes, textured=os.environ.get('TEXTURED') == '1')
elif os.environ.get("SHAPE") == "c3":
    desc = scenes.synthetic_scene(worlds, width=128, height=128, with_wall=True)
else:           # cubes = 0: the BASELINE configs[4] shape (256x256 Raytracer, textured cube + plane), `worlds` views
    desc = scenes.synthetic_scene(worlds, width=256, height=256, textured=True, render_mode="Raytracer")
r = scenes.make_renderer(desc)
print('%.1f us/step' % (r.time_renders(100) / 100 * 1000))
for _ in range(5):
    r.step()
r.sync()
lib = pkg.load_capi()
lib.mrx_debug_stamps.restype = ctypes.c_int64
lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
tiles = ((desc.width + 63) // 64) * ((desc.height + 63) // 64) if cubes == 0 else 1
buf = np.zeros(worlds * tiles * 4 * 8, np.uint64)
n = lib.mrx_debug_stamps(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size)
st = buf[:n].reshape(-1, 4, 8).astype(np.int64)
st = st[st[:, 0, 0] != 0]                 # (groups of tiles: fewer workgroups than tiles)
if os.environ.get("FLAT") == "1":
    # bvhFlatKernel: 0 entry, 7 wave reaches the set-up barrier, 1 barrier passed, then of the LAST tile of the group:
    # 2 classification + small walk done, 3 large pass done, 4 barrier passed, 5 resolve + stores issued, 6 exit
    t0 = st[:, :, 0].min()
    us = (st - t0) / 100.0
    order = [0, 7, 1, 2, 3, 4, 5, 6]
    names = ["entry", "at set-up barrier", "set-up barrier passed", "last tile: small walk done", "large pass done",
             "barrier passed", "resolve done", "exit"]
    for w in range(4):
        print("wave %d" % w)
        for i, nm in zip(order, names):
            v = us[:, w, i]
            print(f"  {nm:28s} p10 {np.percentile(v, 10):7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}")
    life = us[:, :, 6] - us[:, :, 0]
    print("workgroup life p50 %.2f us, set-up (entry -> barrier passed) p50 %.2f, per tile (life - set-up) / tiles-per-group" %
          (np.median(life), np.median(us[:, :, 1] - us[:, :, 0])))
    d = {"walk->large": us[:, :, 3] - us[:, :, 2], "large->barrier": us[:, :, 4] - us[:, :, 3],
         "barrier->resolve": us[:, :, 5] - us[:, :, 4]}
    for k, v in d.items():
        print(f"  last tile {k:18s} p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}")
    print("kernel span %.1f us" % us[:, :, 6].max())
    sys.exit(0)
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0
names = ["entry", "TLAS built (barrier)", "produce done", "barrier 1", "large pass done", "resolve done (barrier 3)", "exit"]
if os.environ.get("MRX_DEBUG_SKIP") == "128":
    names = ["entry", "TLAS built (barrier)", "batch queued", "batch set up", "small walk done", "produce done", "exit"]
entry = us[:, 0, 0]
gen1 = entry < np.median(entry) - 1.0 if entry.max() - entry.min() > 4 else np.ones(len(entry), bool)
for label, sel in (("first generation", gen1), ("second generation", ~gen1)):
    if sel.sum() == 0:
        continue
    print("==", label, int(sel.sum()), "workgroups")
    u = us[sel]
    for i, nm in enumerate(names):
        v = u[:, :, i].reshape(-1)
        print(f"  {nm:26s} p10 {np.percentile(v, 10):6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
    d = np.diff(u[:, :, :7], axis=2)
    print("  phase durations p50:", " ".join(f"{nm.split()[0]}:{np.median(d[:, :, i]):.2f}" for i, nm in enumerate(names[1:])))
    print("  phase durations p90:", " ".join(f"{nm.split()[0]}:{np.percentile(d[:, :, i], 90):.2f}" for i, nm in enumerate(names[1:])))
print("kernel span %.1f us" % us[:, :, 6].max())
