"""Helpers shared by the parity tests."""
import os

import numpy as np


def render_oracle(desc, **kw):
    from oracle import oracle
    return oracle.FlatScene(desc).render(**kw)


def make_product(desc, visibility=True, gpu_id=0, variant=None):
    from madrona_renderer_amd import scenes
    old = {k: os.environ.get(k) for k in ("MADRONA_MI355_VISIBILITY", "MADRONA_MI355_KERNEL")}
    os.environ["MADRONA_MI355_VISIBILITY"] = "1" if visibility else "0"
    if variant is not None:
        os.environ["MADRONA_MI355_KERNEL"] = str(variant)
    try:
        return scenes.make_renderer(desc, gpu_id=gpu_id)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def fetch(r, visibility=True, raytracer=False):
    """Outputs of a product renderer as numpy arrays in oracle layout."""
    r.sync()
    out = {"rgb": r.rgb_tensor().to_torch().cpu().numpy()}
    d = r.depth_tensor().to_torch().cpu().numpy()
    out["depth"] = d.reshape(d.shape[0], d.shape[1], d.shape[2])
    if visibility:
        out["tri_id"] = r.visibility_tensor().to_torch().cpu().numpy()
    elif raytracer:
        out["segmask"] = r.segmask_tensor().to_torch().cpu().numpy()
    return out


def assert_parity(got, ref, depth_rtol=1e-4):
    """Bit-exact visibility and colour; depth within the north star's 1e-4."""
    if "tri_id" in got:
        bad = int((got["tri_id"] != ref["tri_id"]).sum())
        assert bad == 0, f"{bad} pixels differ in visibility"
    if "segmask" in got:
        bad = int((got["segmask"] != ref["segmask"]).sum())
        assert bad == 0, f"{bad} pixels differ in segmask"
    bad = int((got["rgb"] != ref["rgb"]).any(axis=-1).sum())
    assert bad == 0, f"{bad} pixels differ in colour"
    assert got["depth"].shape == ref["depth"].shape
    np.testing.assert_allclose(got["depth"], ref["depth"], rtol=depth_rtol, atol=0)


def digest(arr):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
