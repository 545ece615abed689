"""Does the group kernel's launch end with its youngest workgroups?  In-kernel stamps (MRX_DEBUG_STAMPS=1: the plain
entry point, not the preloaded-header one) by quarter of the workgroup index: a CU holds four of the headline's
workgroups, dispatched one per CU and round (GPU box).  python scripts/raster_age_order.py [worlds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_DEBUG_STAMPS"] = "1"
import numpy as np
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
r = scenes.make_renderer(scenes.synthetic_scene(worlds))
print("%d worlds: %.2f us/step (stamps compiled in)" % (worlds, min(r.time_renders(200) for _ in range(3)) / 200 * 1000))
for _ in range(5):
    r.step()
r.sync()
lib = pkg.load_capi()
lib.mrx_debug_stamps.restype = ctypes.c_int64
lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
buf = np.zeros(worlds * 4 * 8, np.uint64)
n = lib.mrx_debug_stamps(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size)
st = buf[:n].reshape(-1, 4, 8).astype(np.int64)
ids = np.nonzero(st[:, 0, 0] > 0)[0]
wgs = st[ids]
t0 = wgs[:, :, 0].min()
us = (wgs - t0) / 100.0
print("workgroups %d, span %.2f us" % (len(wgs), us[:, :, 6].max()))
q = len(wgs) // 4
for k in range(4):
    sel = slice(k * q, (k + 1) * q)
    e, b, x = us[sel, :, 0].min(axis=1), us[sel, :, 4].max(axis=1), us[sel, :, 6].max(axis=1)
    print("  workgroups %4d..%4d: entry p50 %5.2f  set-up + classify done p50 %5.2f  exit p50 %5.2f p90 %5.2f max %5.2f  life p50 %5.2f"
          % (ids[k * q], ids[(k + 1) * q - 1], np.median(e), np.median(b), np.median(x), np.percentile(x, 90), x.max(), np.median(x - e)))
