"""(Needs the experiment build of profiles/r03_bvh_priority.txt: the per-view clock and MRX_BVH_BALANCE are not in the product.)
Are the expensive views of the BVH kernel expensive by content or by where they run?  Costs per view as the
kernel records them (mrx_debug_bvh_order) under consecutive groups and under groups dealt by cost (GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MADRONA_MI355_KERNEL"] = "2"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
import numpy as np
import torch  # noqa
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
n = 1024
desc = scenes.cube_field(n, int(os.environ.get("CUBES", "40")), textured=os.environ.get("TEXTURED") == "1")
lib = pkg.load_capi()
lib.mrx_debug_bvh_order.restype = ctypes.c_int64
lib.mrx_debug_bvh_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]


def grab(r):
    buf = np.zeros(2 * n, np.uint32)
    assert lib.mrx_debug_bvh_order(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size) == 2 * n
    return buf[:n].astype(np.int64), buf[n:].astype(np.float64) / 100.0


def stats(tag, v):
    print("%-44s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (tag, v.min(), np.median(v), np.percentile(v, 90), v.max()))


costs = {}
for mode, name in (("3", "consecutive groups (costs recorded only)"), ("2", "groups dealt by cost"), ("p1", "consecutive, young workgroups at priority 1"), ("p2", "consecutive, young at priority 1 in their first view"), ("p3", "consecutive, young at priority 1 in the first view, old in the second")):
    os.environ["MRX_BVH_BALANCE"] = mode if mode[0] != "p" else "3"
    os.environ["MRX_BVH_PRIO"] = mode[1] if mode[0] == "p" else "0"
    r = scenes.make_renderer(desc)
    for _ in range(40):
        r.step()
    r.sync()
    order_used, c0 = grab(r)         # (the order is rebuilt after launches 1, 16, 32, 48: 41 launches so far)
    r.step(); r.sync()
    _, c1 = grab(r)
    us = min(r.time_renders(200) for _ in range(3)) / 200 * 1000
    print("== %s: %.2f us/step" % (name, us))
    stats("  cost of a view (behind phase I, us)", c1)
    print("  correlation of a view's cost between two launches: %.3f" % np.corrcoef(c0, c1)[0, 1])
    if mode != "2":
        pairs = c1[0::2] + c1[1::2]
        print("  workgroups 0..255: mean %.2f   256..511: mean %.2f" % (pairs[:256].mean(), pairs[256:].mean()))
    else:
        pairs = c1[order_used[:n // 2]] + c1[order_used[::-1][:n // 2]]
    stats("  sum over a workgroup's two views", pairs)
    costs[mode] = c1
    del r
print("correlation of a view's cost between the two modes: %.3f" % np.corrcoef(costs["3"], costs["2"])[0, 1])
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bvh_costs_%s.npy" % ("tex" if os.environ.get("TEXTURED") == "1" else "plain")),
        np.stack([costs["3"], costs["2"]]))
