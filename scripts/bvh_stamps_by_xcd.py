"""Where the BVH kernel's slow workgroups run (diagnostics build, MRX_DEBUG_STAMPS=1): life of a workgroup by
the XCD it ran on.  scripts/ab_build.sh diag -DMRX_BVH_DIAG=1; scripts/ab_run.sh "python scripts/bvh_stamps_by_xcd.py" diag"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_DEBUG_STAMPS"] = "1"
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
import numpy as np
import torch  # noqa
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
desc = scenes.cube_field(worlds, cubes, textured=os.environ.get("TEXTURED") == "1")
r = scenes.make_renderer(desc)
print("%.1f us/step" % (r.time_renders(100) / 100 * 1000))
lib = pkg.load_capi()
lib.mrx_debug_stamps.restype = ctypes.c_int64
lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
for rep in range(3):
    for _ in range(5):
        r.step()
    r.sync()
    buf = np.zeros(worlds * 4 * 8, np.uint64)
    n = lib.mrx_debug_stamps(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size)
    st = buf[:n].reshape(-1, 4, 8)
    st = st[st[:, 0, 0] != 0]
    t = st[:, :, :7].astype(np.int64)
    t0 = t[:, :, 0].min()
    entry = (t[:, 0, 0] - t0) / 100.0
    exit_ = (t[:, :, 6].max(axis=1) - t0) / 100.0
    xcc = (st[:, 0, 7] >> np.uint64(32)).astype(np.int64) & 15
    cu = ((st[:, 0, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 8) & 15
    print("render %d: %d workgroups, span %.1f us" % (rep, len(st), exit_.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            life = exit_[m] - entry[m]
            print("  XCD %d: %4d workgroups  entry p50 %5.2f  life p50 %5.2f p90 %5.2f max %5.2f  exit p50 %5.2f max %5.2f"
                  % (x, m.sum(), np.median(entry[m]), np.median(life), np.percentile(life, 90), life.max(),
                     np.median(exit_[m]), exit_[m].max()))
    slow = np.argsort(exit_)[-8:]
    print("  last to exit: " + ", ".join("wg %d (xcd %d, life %.1f)" % (i, xcc[i], exit_[i] - entry[i]) for i in slow))
