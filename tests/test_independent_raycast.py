"""An independent algorithm against the oracle.

The reference ships no pixels, so the oracle's *semantics* -- which triangle a
pixel sees, how far away -- are checked here by a ray caster that shares
nothing with it beyond the scene conventions of SURVEY.md section 8b: float64
Moeller-Trumbore per pixel centre, world-space rays, no edge functions, no
plane set-up, no S6b culling, NumPy only.  Wherever the nearest hit is
unambiguous in float64 (the runner-up is clearly further, the hit is clearly
inside its triangle and clearly inside the near / far range) the oracle must
name the same triangle, and its depth must agree to 1e-5 (1e-4 where the
+-10000-unit ground quad is in view: its float32 planes are that coarse).  This pins S2-S6,
S6b (image-preserving) and S9; it cannot pin tie-breaking, shading constants or
texel choice -- those remain build-defined.  The colour test at the end of the file does the same
for S4 / S7 / S8: a float64 shading model from first principles against the oracle's RGB bytes."""
import math
import os

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.golden.make_golden import cases
from tests.util import render_oracle


def quat_to_mat(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def raycast_view(fs, v):
    """(tri [H,W] or -1, depth [H,W], margin [H,W]): nearest hit per pixel in
    float64 and how unambiguous it is (>= 0; small = too close to call)."""
    W, H = fs.width, fs.height
    w = int(fs.view_world[v])
    Rc = quat_to_mat(fs.cam_rot[v])
    c = fs.cam_pos[v].astype(np.float64)
    th = math.tan(math.radians(45.0))
    asp = W / H
    px = (np.arange(W) + 0.5) / W * 2 - 1
    py = 1 - (np.arange(H) + 0.5) / H * 2
    X, Z = np.meshgrid(px * th * asp, py * th)                 # camera: +X right, +Y forward, +Z up
    dirs = np.stack([X, np.ones_like(X), Z], axis=-1) @ Rc.T   # world-space ray directions, |d_y(cam)| = 1
    fwd = Rc[:, 1]
    rt = fs.raytracer
    near, far = (0.1, 1000.0) if rt else (0.001, np.inf)
    best = np.full((H, W), np.inf)
    second = np.full((H, W), np.inf)
    tri = np.full((H, W), -1, np.int64)
    inside = np.zeros((H, W))
    k = 0
    for i in range(fs.world_inst_start[w], fs.world_inst_start[w + 1]):
        obj = int(fs.inst_obj[i])
        if obj < 0 or obj >= len(fs.obj_first_tri):
            continue
        M = quat_to_mat(fs.inst_rot[i]) * fs.inst_scale[i].astype(np.float64)[None, :]
        t = fs.inst_pos[i].astype(np.float64)
        f0, n = int(fs.obj_first_tri[obj]), int(fs.obj_num_tris[obj])
        for ti in range(f0, f0 + n):
            P = fs.tri_pos[ti].astype(np.float64) @ M.T + t    # [3, 3] world space
            e1, e2 = P[1] - P[0], P[2] - P[0]
            pvec = np.cross(dirs, e2)
            det = pvec @ e1
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / det
                tvec = c - P[0]
                u = (pvec @ tvec) * inv
                qvec = np.cross(tvec, e1)
                vv = (dirs @ qvec) * inv
                tt = (qvec @ e2) * inv                          # distance along the ray = view depth (d_y = 1)
            bary = np.minimum(np.minimum(u, vv), 1 - u - vv)
            ok = (np.abs(det) > 0) & (bary >= 0) & (tt >= near) & (tt <= far)
            tt = np.where(ok, tt, np.inf)
            closer = tt < best
            second = np.where(closer, best, np.minimum(second, tt))
            inside = np.where(closer, bary / np.maximum(1e-300, np.abs(u) + np.abs(vv) + 1), inside)
            tri = np.where(closer, k, tri)
            best = np.where(closer, tt, best)
            k += 1
    hit = np.isfinite(best)
    # how decisive: relative gap to the runner-up, barycentric margin, distance from the near / far limits
    safe = np.where(hit, best, 1.0)
    gap = np.where(np.isfinite(second), (np.where(np.isfinite(second), second, 0.0) - safe) / safe, np.inf)
    rng = (safe - near) / safe
    if np.isfinite(far):
        rng = np.minimum(rng, (far - safe) / safe)
    margin = np.where(hit, np.minimum(np.minimum(gap, inside), rng), 0.0)
    depth = np.where(hit, best, 0.0)
    if rt:                                                     # Raytracer storage is [x][y]
        return tri.T, depth.T, margin.T
    return tri, depth, margin


@pytest.mark.parametrize("name", sorted(cases()))
def test_oracle_agrees_with_float64_moeller_trumbore(oracle_mod, name):
    desc = cases()[name]
    fs = oracle_mod.FlatScene(desc)
    ref = fs.render()
    checked = hits = 0
    for v in range(min(fs.num_views, 2)):
        tri, depth, margin = raycast_view(fs, v)
        sure = margin > 1e-6
        assert np.array_equal(ref["tri_id"][v][sure], tri[sure]), \
            f"view {v}: {(ref['tri_id'][v][sure] != tri[sure]).sum()} decisive pixels name another triangle"
        # float32 planes of the +-10000-unit ground quad carry ~5e-5 of relative depth
        # error (the north star's bound is 1e-4); scenes without it agree to 1e-5
        np.testing.assert_allclose(ref["depth"][v][sure], depth[sure],
                                   rtol=1e-5 if name.startswith("demo") else 1e-4)
        # coverage (hit / miss) may differ only on the silhouette: a pixel whose
        # float64 neighbourhood holds both hits and misses
        cov_ref, cov_mt = ref["tri_id"][v] >= 0, tri >= 0
        pad = np.pad(cov_mt, 1, mode="edge")
        near_hit = np.zeros_like(cov_mt)
        near_miss = np.zeros_like(cov_mt)
        for dy in range(3):
            for dx in range(3):
                win = pad[dy:dy + cov_mt.shape[0], dx:dx + cov_mt.shape[1]]
                near_hit |= win
                near_miss |= ~win
        silhouette = near_hit & near_miss
        assert not ((cov_ref != cov_mt) & ~silhouette).any()
        assert (cov_ref != cov_mt).sum() <= 0.02 * max(1, cov_ref.sum()) + 2
        checked += int(sure.sum())
        hits += int(cov_ref.sum())
    assert checked > 0.9 * hits > 0, f"only {checked} of {hits} covered pixels were decisive"


def test_back_face_rule_is_image_preserving_on_closed_meshes(oracle_mod):
    # S6b in the oracle vs no culling at all in the ray caster: cubes seen from
    # outside, from inside, mirrored -- every decisive pixel names the same triangle
    ident = (1.0, 0.0, 0.0, 0.0)
    cube = os.path.join(scenes.DATA_DIR, "cube.obj")
    d = scenes.SceneDesc(
        num_worlds=3, width=64, height=64, asset_paths=[(cube, -1)],
        instances=[((0.0, 6.0, 0.0), (0.9238795, 0.0, 0.3826834, 0.0), (2.0, 2.0, 2.0), 0),
                   ((0.3, 0.2, 0.1), ident, (4.0, 4.0, 4.0), 0),
                   ((1.0, 5.0, 0.5), (0.9659258, 0.0, 0.0, 0.2588190), (-2.0, 1.5, 2.5), 0)],
        cameras=[((0.0, 0.0, 0.0), ident)], worlds=[(1, 0, 1, 0), (1, 1, 1, 0), (1, 2, 1, 0)])
    fs = oracle_mod.FlatScene(d)
    ref = fs.render()
    for v in range(3):
        tri, depth, margin = raycast_view(fs, v)
        sure = margin > 1e-6
        assert sure.sum() > 100
        assert np.array_equal(ref["tri_id"][v][sure], tri[sure])
        np.testing.assert_allclose(ref["depth"][v][sure], depth[sure], rtol=1e-5)


def test_meshes_with_hierarchies_agree_with_float64_moeller_trumbore(oracle_mod):
    # curved closed meshes (sphere, torus: S6b culls their back faces), an open terrain, a mirrored
    # instance, cameras on the terrain and inside boxes -- the scenes the BVH path is tested on
    from tests import meshes
    d = meshes.mesh_scene_random_cameras(300, 48, 48, "Rasterizer")
    fs = oracle_mod.FlatScene(d)
    ref = fs.render()
    checked = hits = 0
    for v in (0, 2, 5):
        tri, depth, margin = raycast_view(fs, v)
        sure = margin > 1e-5
        assert np.array_equal(ref["tri_id"][v][sure], tri[sure]), \
            f"view {v}: {(ref['tri_id'][v][sure] != tri[sure]).sum()} decisive pixels name another triangle"
        np.testing.assert_allclose(ref["depth"][v][sure], depth[sure], rtol=1e-4)
        # ... and the colours of those pixels follow from the float64 shading model below
        rgb, sure_tex = raycast_colour(fs, v)
        near_half = np.abs(rgb - np.floor(rgb) - 0.5) < 0.02
        diff = np.abs(ref["rgb"][v][..., :3].astype(np.float64) - np.floor(rgb + 0.5))
        assert not ((sure & sure_tex)[..., None] & (diff > np.where(near_half, 1.0, 0.0))).any()
        checked += int(sure.sum())
        hits += int((ref["tri_id"][v] >= 0).sum())
    assert checked > 0.8 * hits > 0, f"only {checked} of {hits} covered pixels were decisive"


def raycast_colour(fs, v):
    """(rgb [H,W,3] float64 prediction in 0..255 before rounding, decisive [H,W] bool, tri, margin): the
    colour of the nearest hit per pixel from first principles in float64 -- world-space normal turned
    towards the eye, one directional light travelling along (1, -1, -0.05), 0.25 ambient + 0.75
    diffuse, material colour, nearest texel of the perspective-correct (barycentric) uv with v up."""
    W, H = fs.width, fs.height
    w = int(fs.view_world[v])
    Rc = quat_to_mat(fs.cam_rot[v])
    c = fs.cam_pos[v].astype(np.float64)
    th = math.tan(math.radians(45.0))
    asp = W / H
    px = (np.arange(W) + 0.5) / W * 2 - 1
    py = 1 - (np.arange(H) + 0.5) / H * 2
    X, Z = np.meshgrid(px * th * asp, py * th)
    dirs = np.stack([X, np.ones_like(X), Z], axis=-1) @ Rc.T
    to_light = -np.array([1.0, -1.0, -0.05])
    to_light /= np.linalg.norm(to_light)
    rt = fs.raytracer
    near, far = (0.1, 1000.0) if rt else (0.001, np.inf)
    best = np.full((H, W), np.inf)
    rgb = np.zeros((H, W, 3))
    sure_tex = np.ones((H, W), bool)
    for i in range(fs.world_inst_start[w], fs.world_inst_start[w + 1]):
        obj = int(fs.inst_obj[i])
        if obj < 0 or obj >= len(fs.obj_first_tri):
            continue
        M = quat_to_mat(fs.inst_rot[i]) * fs.inst_scale[i].astype(np.float64)[None, :]
        t = fs.inst_pos[i].astype(np.float64)
        f0, n = int(fs.obj_first_tri[obj]), int(fs.obj_num_tris[obj])
        for ti in range(f0, f0 + n):
            P = fs.tri_pos[ti].astype(np.float64) @ M.T + t
            e1, e2 = P[1] - P[0], P[2] - P[0]
            pvec = np.cross(dirs, e2)
            det = pvec @ e1
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / det
                tvec = c - P[0]
                u = (pvec @ tvec) * inv
                qvec = np.cross(tvec, e1)
                vv = (dirs @ qvec) * inv
                tt = (qvec @ e2) * inv
            ok = (np.abs(det) > 0) & (np.minimum(np.minimum(u, vv), 1 - u - vv) >= 0) & (tt >= near) & (tt <= far)
            closer = ok & (tt < best)
            if not closer.any():
                continue
            nrm = np.cross(e1, e2)
            ln = np.linalg.norm(nrm)
            if ln == 0:
                continue
            if nrm @ (c - P[0]) < 0:
                nrm = -nrm
            lit = 0.25 + 0.75 * max(float(nrm @ to_light) / ln, 0.0)
            mi = int(fs.tri_mat[ti])
            col, tex = np.ones(3), -1
            if 0 <= mi < len(fs.mat_color):
                col, tex = fs.mat_color[mi][:3].astype(np.float64), int(fs.mat_tex[mi])
            if not (0 <= tex < fs.num_textures):
                val = np.broadcast_to(255.0 * np.clip(lit * col, 0, 1), (H, W, 3))
                st = np.ones((H, W), bool)
            else:
                uv = fs.tri_uv[ti].astype(np.float64)
                U = (1 - u - vv) * uv[0, 0] + u * uv[1, 0] + vv * uv[2, 0]
                V = (1 - u - vv) * uv[0, 1] + u * uv[1, 1] + vv * uv[2, 1]
                tw_, th_ = int(fs.tex_w[tex]), int(fs.tex_h[tex])
                with np.errstate(invalid="ignore"):
                    fu = (U - np.floor(U)) * tw_
                    fv = (1.0 - (V - np.floor(V))) * th_
                    st = (np.abs(fu - np.round(fu)) > 2e-3) & (np.abs(fv - np.round(fv)) > 2e-3)
                    tx = np.clip(np.nan_to_num(fu).astype(np.int64), 0, tw_ - 1)
                    ty = np.clip(np.nan_to_num(fv).astype(np.int64), 0, th_ - 1)
                texel = fs.tex_data[int(fs.tex_offset[tex]) + ty * tw_ + tx][..., :3].astype(np.float64)
                val = 255.0 * np.clip(texel / 255.0 * lit * col, 0, 1)
            rgb = np.where(closer[..., None], val, rgb)
            sure_tex = np.where(closer, st, sure_tex)
            best = np.where(closer, tt, best)
    if rt:
        return np.transpose(rgb, (1, 0, 2)), sure_tex.T
    return rgb, sure_tex


@pytest.mark.parametrize("name", sorted(cases()))
def test_colours_agree_with_a_float64_shading_model(oracle_mod, name):
    # S4 / S7 / S8 from first principles: where visibility and texel choice are decisive the oracle's
    # bytes are within one level of the float64 colour (the constants -- light, 0.25 / 0.75 -- are the
    # build's; this checks the arithmetic that applies them, with an algorithm that shares none of it)
    desc = cases()[name]
    fs = oracle_mod.FlatScene(desc)
    ref = fs.render()
    checked = 0
    for v in range(min(fs.num_views, 2)):
        tri, depth, margin = raycast_view(fs, v)
        rgb, sure_tex = raycast_colour(fs, v)
        sure = (margin > 1e-4) & sure_tex & (tri >= 0)
        got = ref["rgb"][v][..., :3].astype(np.float64)
        diff = np.abs(got - np.floor(rgb + 0.5))
        # a colour that sits within 0.02 of a rounding boundary may land on either side
        near_half = np.abs(rgb - np.floor(rgb) - 0.5) < 0.02
        bad = sure[..., None] & (diff > np.where(near_half, 1.0, 0.0))
        assert not bad.any(), f"view {v}: {int(bad.any(axis=-1).sum())} decisive pixels differ, max {diff[sure].max()}"
        assert (ref["rgb"][v][..., 3][tri >= 0] == 255).all()
        checked += int(sure.sum())
    assert checked > 100
