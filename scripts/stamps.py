"""In-kernel timeline (diagnostic build path: MRX_DEBUG_STAMPS=1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_DEBUG_STAMPS"] = "1"
import numpy as np
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wall = len(sys.argv) > 3 and sys.argv[3] == "wall"
tex = len(sys.argv) > 4 and sys.argv[4] == "tex"
mode = sys.argv[5] if len(sys.argv) > 5 else "Rasterizer"
instances = int(sys.argv[6]) if len(sys.argv) > 6 else 1
desc = scenes.synthetic_scene(worlds, width=size, height=size, with_wall=wall, textured=tex, render_mode=mode)

for inst in range(instances):
    r = scenes.make_renderer(desc)
    print('== instance', inst, ' %.1f us/step' % (r.time_renders(50) / 50 * 1000))
    for _ in range(5):
        r.step()
    r.sync()
    lib = pkg.load_capi()
    lib.mrx_debug_stamps.restype = ctypes.c_int64
    lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    buf = np.zeros(worlds * ((size + 63) // 64) ** 2 * 4 * 8, np.uint64)
    n = lib.mrx_debug_stamps(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size)
    st = buf[:n].reshape(-1, 4, 8).astype(np.int64)
    wgs = st[st[:, 0, 0] > 0]
    t0 = wgs[:, :, 0].min()
    us = (wgs - t0) / 100.0
    print("workgroups", len(wgs), " kernel span %.1f us" % us[:, :, 6].max())
    names = ["entry", "S loads issued", "setup done", "classify done", "after barrier", "tile0 done", "exit"]
    for i, nm in enumerate(names):
        sel = us[:, 0, i] if i in (1, 2, 3) else us[:, :, i].reshape(-1)
        print(f"{nm:16s} min {sel.min():6.2f}  p50 {np.median(sel):6.2f}  p90 {np.percentile(sel, 90):6.2f}  max {sel.max():6.2f}")
    life = us[:, :, 6] - us[:, :, 0]
    print("wave lifetime   p50 %.2f  max %.2f" % (np.median(life), life.max()))
    print("S phase (wave0) p50 %.2f" % np.median(us[:, 0, 3] - us[:, 0, 0]))
    print("barrier->exit   p50 %.2f" % np.median(us[:, :, 6] - us[:, :, 4]))

    # where did the spread come from?  group exits by XCC and by CU
    hw = st[st[:, 0, 0] > 0][:, 0, 7]
    xcc = (hw >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    ex = us[:, :, 6].max(axis=1)
    print("exit by XCC:", " ".join(f"{x}:{ex[xcc == x].mean():.1f}/{ex[xcc == x].max():.1f}" for x in sorted(set(xcc.tolist()))))
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    import collections
    per_cu = collections.defaultdict(list)
    for k, e in zip(key.tolist(), ex.tolist()):
        per_cu[k].append(e)
    within = np.mean([max(v) - min(v) for v in per_cu.values() if len(v) > 1])
    cu_last = np.array([max(v) for v in per_cu.values()])
    print(f"distinct CUs seen {len(per_cu)}; WGs per CU {np.mean([len(v) for v in per_cu.values()]):.2f}; "
          f"mean spread of exits within a CU {within:.2f} us; per-CU last exit: min {cu_last.min():.1f} "
          f"p50 {np.median(cu_last):.1f} max {cu_last.max():.1f}")
    for i, nm in ((0, "entry"), (4, "after barrier")):
        col = us[:, :, i].min(axis=1) if i == 0 else us[:, :, i].max(axis=1)
        print(f"{nm} by XCC:", " ".join(f"{x}:{col[xcc == x].mean():.1f}/{col[xcc == x].max():.1f}" for x in sorted(set(xcc.tolist()))))
    print("WGs per XCC:", " ".join(f"{x}:{int((xcc == x).sum())}" for x in sorted(set(xcc.tolist()))))
    ids = np.nonzero(st[:, 0, 0] > 0)[0]
    print("WGs whose XCC differs from blockIdx % 8:", int((xcc != (ids % 8)).sum()), "of", len(ids))
    del r
