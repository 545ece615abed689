import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import test_bvh_gpu as T
from tests.util import fetch, make_product, render_oracle
d = T._mesh_world("Raytracer", 128, 128)
r = make_product(d, visibility=True)
got = fetch(r); ref = render_oracle(d)
bad = (got["rgb"] != ref["rgb"]).any(axis=-1)
print("bad px", bad.sum(), "ids differ", (got["tri_id"] != ref["tri_id"]).sum())
for v in range(bad.shape[0]):
    ys, xs = np.nonzero(bad[v])
    if len(ys) == 0: continue
    print("view", v, "n", len(ys), "tiles", sorted(set(zip((ys // 64).tolist(), (xs // 64).tolist()))))
    ids = ref["tri_id"][v][bad[v]]
    print("  tri ids range", ids.min(), ids.max(), "unique", len(np.unique(ids)))
    for i in range(min(6, len(ys))):
        y, x = ys[i], xs[i]
        print("  ", y, x, "id", ref["tri_id"][v, y, x], "got", got["rgb"][v, y, x], "ref", ref["rgb"][v, y, x], "seg", ref["segmask"][v, y, x])
