"""Regenerates tests/golden/golden.json and golden_views.npz from the CPU
oracle.  The reference ships no golden images, so these
fixtures pin THIS repo's rendering spec: the oracle must keep reproducing them
(guards against silent spec drift) and the HIP path must match them bit for bit.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from madrona_renderer_amd import scenes          # noqa: E402
from oracle import oracle                        # noqa: E402
from tests.util import digest                    # noqa: E402


def cases():
    return {
        "demo_raster_64": scenes.demo_scene(num_worlds=2, render_mode="Rasterizer"),
        "demo_raytracer_64": scenes.demo_scene(num_worlds=2, render_mode="Raytracer"),
        "synthetic_wall_textured_64": scenes.synthetic_scene(4, with_wall=True, textured=True),
        "synthetic_plain_64": scenes.synthetic_scene(8),
        "synthetic_wall_128": scenes.synthetic_scene(2, width=128, height=128, with_wall=True),
        "synthetic_rt_textured_96": scenes.synthetic_scene(
            2, width=96, height=96, textured=True, render_mode="Raytracer"),
    }


PROBES = [(0, 40, 32), (0, 10, 10), (1, 50, 20), (1, 36, 31)]


def main():
    meta, views = {}, {}
    for name, desc in cases().items():
        o = oracle.FlatScene(desc).render()
        meta[name] = {
            "shape": list(o["rgb"].shape),
            "rgb_sha256": digest(o["rgb"]),
            "depth_sha256": digest(o["depth"]),
            "tri_id_sha256": digest(o["tri_id"]),
            "segmask_sha256": digest(o["segmask"]),
            "covered": int((o["tri_id"] >= 0).sum()),
            "probes": [{"at": list(p), "rgb": o["rgb"][p].tolist(),
                        "depth_bits": int(o["depth"][p].view(np.uint32)),
                        "tri_id": int(o["tri_id"][p])} for p in PROBES],
        }
        views[name + "/rgb"] = o["rgb"][0]
        views[name + "/depth"] = o["depth"][0]
        views[name + "/tri_id"] = o["tri_id"][0].astype(np.int16)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_views.npz"), **views)
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
