"""Pins the CPU oracle (and with it the rendering spec) on known answers that
follow from the reference's own constants, since the reference ships no golden
images:

 * the demo camera / cube / triangle of /root/reference/scripts/test.py:36-55,
 * vfov 90 deg, znear 0.001 (/root/reference/src/sim.cpp:168-171),
 * Raytracer near 0.1 / far 1000 and [x][y] storage
   (/root/reference/src/mgr.cpp:477-478, scripts/test.py:160),
 * output shapes / dtypes (/root/reference/src/mgr.cpp:552-603).
"""
import math
import os

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.util import render_oracle

DEMO_CAM = ((-22.343935, -21.845375, 27.061676),
            (0.913407, -0.112268, 0.047731, -0.388336))


def _one_world(instances, assets, width=64, height=64, cam=DEMO_CAM, mode="Rasterizer",
               materials=(), textures=()):
    return scenes.SceneDesc(
        num_worlds=1, render_mode=mode, width=width, height=height,
        asset_paths=list(assets), materials=list(materials),
        texture_paths=list(textures), instances=list(instances),
        cameras=[cam], worlds=[(len(instances), 0, 1, 0)])


CUBE = os.path.join(scenes.DATA_DIR, "cube.obj")
PLANE = os.path.join(scenes.DATA_DIR, "plane.obj")


def test_cube_centre_projects_where_the_reference_constants_say(oracle_mod):
    # cube centre (0,0,15) -> view (-0.229, 33.239, -4.136) -> pixel (31.78, 35.98)
    d = _one_world([((0.0, 0.0, 15.0), (1.0, 0.0, 0.0, 0.0), (1.2, 1.2, 1.2), 0)],
                   [(CUBE, -1)])
    o = render_oracle(d)
    ys, xs = np.nonzero(o["tri_id"][0] >= 0)
    assert len(xs) >= 1
    assert xs.min() >= 30 and xs.max() <= 33, (xs.min(), xs.max())
    assert ys.min() >= 34 and ys.max() <= 37, (ys.min(), ys.max())
    dep = o["depth"][0][ys, xs]
    assert np.all(np.abs(dep - 33.239) < 1.1)     # within the cube's half diagonal


def test_demo_cube_pixel_box(oracle_mod):
    # scale 3, 90 deg about X: corners span x in [29.71,33.84], y in [33.98,38.19]
    d = scenes.demo_scene(num_worlds=1, render_mode="Rasterizer")
    o = render_oracle(d)
    ids = o["tri_id"][0]
    cube = (ids >= 0) & (ids < 12)
    ys, xs = np.nonzero(cube)
    assert xs.min() >= 29 and xs.max() <= 34
    assert ys.min() >= 33 and ys.max() <= 38
    assert 9 <= cube.sum() <= 25                  # "about 4x4 px"


def test_demo_triangle_needs_no_clipper(oracle_mod):
    # raw triangle (scale 10): view depths 33.24 / 0.855 / 103.1, pixel x up to
    # ~4015: coverage must stay finite and inside that depth range
    d = scenes.demo_scene(num_worlds=1, render_mode="Rasterizer")
    o = render_oracle(d)
    tri = o["tri_id"][0] == 12
    assert tri.sum() > 50
    dep = o["depth"][0][tri]
    assert np.isfinite(dep).all()
    assert dep.min() > 0.855 and dep.max() < 103.1


def test_ground_plane_horizon_row_and_centre_depth(oracle_mod):
    # a z = 0 plane seen from the demo camera: horizon at image row 24.0, the
    # centre ray meets it at forward depth 111.75
    d = _one_world([((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0)],
                   [(PLANE, -1)])
    o = render_oracle(d)
    hit = o["tri_id"][0] >= 0
    for x in (0, 20, 32, 63):
        rows = np.nonzero(hit[:, x])[0]
        assert rows.min() == 24, (x, rows.min())   # no roll: horizon is level
        assert rows.max() == 63
    centre = o["depth"][0][31:33, 31:33]
    assert abs(centre.mean() - 111.75) / 111.75 < 0.02
    # rows nearer the bottom are nearer the camera
    col = o["depth"][0][24:, 32]
    assert np.all(np.diff(col) < 0)


def test_projection_constants_exact_square(oracle_mod):
    # camera 10 above a 10x10 quad, looking straight down (+Y -> -Z): the quad
    # spans exactly pixels [16,48)^2 at 64x64 and every depth is exactly 10
    down = (math.sqrt(0.5), -math.sqrt(0.5), 0.0, 0.0)
    d = _one_world([((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (0.0005, 0.0005, 1.0), 0)],
                   [(PLANE, -1)], cam=((0.0, 0.0, 10.0), down))
    o = render_oracle(d)
    hit = o["tri_id"][0] >= 0
    assert hit.sum() == 32 * 32
    ys, xs = np.nonzero(hit)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (16, 47, 16, 47)
    np.testing.assert_allclose(o["depth"][0][hit], 10.0, rtol=2e-6)
    # both triangles of the quad present, no hole on the shared diagonal
    assert set(np.unique(o["tri_id"][0][hit])) == {0, 1}


def test_image_axes_x_right_row0_up(oracle_mod):
    # camera at origin looking along +Y (identity rotation); a small cube to the
    # right (+X) and above (+Z) must land right of centre and above centre
    ident = (1.0, 0.0, 0.0, 0.0)
    d = _one_world([((3.0, 10.0, 2.0), ident, (1.0, 1.0, 1.0), 0)], [(CUBE, -1)],
                   cam=((0.0, 0.0, 0.0), ident))
    o = render_oracle(d)
    ys, xs = np.nonzero(o["tri_id"][0] >= 0)
    assert xs.mean() > 32 and ys.mean() < 32
    # view depth is distance along +Y of the nearest face: 10 - 0.5
    assert abs(o["depth"][0][ys, xs].min() - 9.5) < 1e-3


def test_background_and_alpha(oracle_mod):
    d = scenes.demo_scene(num_worlds=1, render_mode="Rasterizer")
    o = render_oracle(d)
    miss = o["tri_id"][0] < 0
    assert miss.any()
    assert np.all(o["rgb"][0][..., 3] == 255)
    assert np.all(o["rgb"][0][miss][:, :3] == 0)
    assert np.all(o["depth"][0][miss] == 0.0)
    assert o["rgb"].dtype == np.uint8 and o["depth"].dtype == np.float32


def test_raytracer_storage_is_transposed_raster(oracle_mod):
    # the reference's RT callers read storage as [x][y] (test.py:160,
    # dump.cpp:9-21); with every depth inside (0.1, 1000) both modes agree
    ra = render_oracle(scenes.demo_scene(num_worlds=2, render_mode="Rasterizer"))
    rt = render_oracle(scenes.demo_scene(num_worlds=2, render_mode="Raytracer"))
    assert np.array_equal(rt["rgb"], ra["rgb"].transpose(0, 2, 1, 3))
    assert np.array_equal(rt["tri_id"], ra["tri_id"].transpose(0, 2, 1))
    np.testing.assert_allclose(rt["depth"], ra["depth"].transpose(0, 2, 1), rtol=1e-6)
    # segmask = objectID of the instance hit: cube 0, raw triangle 1
    assert set(np.unique(rt["segmask"])) == {-1, 0, 1}


def test_raytracer_near_and_far_planes(oracle_mod):
    ident = (1.0, 0.0, 0.0, 0.0)
    insts = [((0.0, 0.05, 0.0), ident, (0.01, 0.01, 0.01), 0),     # nearer than 0.1
             ((0.0, 5.0, 0.0), ident, (1.0, 1.0, 1.0), 0),
             ((0.0, 3000.0, 0.0), ident, (4000.0, 1.0, 4000.0), 0)]  # beyond 1000
    for mode, lo, hi in (("Raytracer", 0.1, 1000.0), ("Rasterizer", 0.001, 1e9)):
        d = _one_world(insts, [(CUBE, -1)], cam=((0.0, 0.0, 0.0), ident), mode=mode)
        o = render_oracle(d)
        dep = o["depth"][0]
        hit = dep > 0
        assert hit.any()
        assert dep[hit].min() >= lo and dep[hit].max() <= hi
    # in raster mode the tiny near cube wins the centre, in RT mode it is clipped
    ra = render_oracle(_one_world(insts, [(CUBE, -1)], cam=((0, 0, 0), ident)))
    assert ra["depth"][0][32, 32] < 0.1
    rt = render_oracle(_one_world(insts, [(CUBE, -1)], cam=((0, 0, 0), ident),
                                  mode="Raytracer"))
    assert abs(rt["depth"][0][32, 32] - 4.5) < 1e-3
    assert (rt["depth"][0] == 0).sum() > 0 or rt["depth"][0].max() <= 1000.0


def test_worlds_aliasing_the_same_rows_render_identically(oracle_mod):
    # scripts/test.py:61-67 -- all worlds use offset 0
    o = render_oracle(scenes.demo_scene(num_worlds=4, render_mode="Rasterizer"))
    for v in range(1, 4):
        assert np.array_equal(o["rgb"][0], o["rgb"][v])
        assert np.array_equal(o["depth"][0], o["depth"][v])


def test_two_sided_coverage_and_light_direction(oracle_mod):
    ident = (1.0, 0.0, 0.0, 0.0)
    verts = np.array([[-1, 0, -1], [1, 0, -1], [0, 0, 1]], np.float32)

    def tri_scene(order):
        return scenes.SceneDesc(
            num_worlds=1, width=64, height=64,
            mesh_vertices=verts, mesh_uvs=np.zeros((3, 2), np.float32),
            mesh_indices=np.array(order, np.uint32),
            mesh_vertex_offsets=np.array([0], np.uint32),
            mesh_indices_offsets=np.array([0], np.uint32),
            mesh_materials=np.array([-1], np.int32),
            instances=[((0.0, 4.0, 0.0), ident, (1.0, 1.0, 1.0), 0)],
            cameras=[((0.0, 0.0, 0.0), ident)], worlds=[(1, 0, 1, 0)])
    a = render_oracle(tri_scene([0, 1, 2]))
    b = render_oracle(tri_scene([0, 2, 1]))
    ca, cb = a["tri_id"][0] >= 0, b["tri_id"][0] >= 0
    assert ca.sum() > 100
    assert (ca != cb).sum() <= 4          # same triangle, either winding
    np.testing.assert_allclose(a["depth"][0][ca & cb], 4.0, rtol=1e-6)
    # facing normal is -Y; light travels along (1,-1,-0.05) so the side facing
    # the camera (normal -Y ... towards the light source at (-1,+1)) is unlit:
    # Lambert term 0 -> ambient 0.25 of white = 64
    assert set(np.unique(a["rgb"][0][ca][:, 0])) == {64}
    # seen from the other side the same triangle faces the light source
    back = tri_scene([0, 1, 2])
    back.cameras = [((0.0, 8.0, 0.0), (0.0, 0.0, 0.0, 1.0))]   # 180 deg about Z
    c = render_oracle(back)
    cc = c["tri_id"][0] >= 0
    lit = 0.25 + 0.75 * (1.0 / math.sqrt(2.0 + 0.05 ** 2))
    assert abs(int(c["rgb"][0][cc][0, 0]) - int(lit * 255 + 0.5)) <= 1


# ---- S6b: back faces of closed meshes are culled only where that cannot
#      change the image (eye outside the object's bounding box)
def test_closed_mesh_detection(oracle_mod):
    cube, _ = oracle_mod.parse_obj(CUBE)
    plane, _ = oracle_mod.parse_obj(PLANE)

    def orient(tris):
        return oracle_mod.shell_orientation(tris)[0].tolist()
    assert orient(cube) == [1.0] * 12
    assert orient(plane) == [0.0] * 2                             # open
    assert orient(cube[:, ::-1]) == [-1.0] * 12                   # inward winding
    assert orient(cube[:-1]) == [0.0] * 11                        # a hole
    mixed = cube.copy()
    mixed[0] = mixed[0, ::-1]                                     # one flipped face
    assert orient(mixed) == [0.0] * 12
    # two shells in one object, wound oppositely: each is judged on its own
    # and carries its own bounding box
    far = cube[:, ::-1] + np.float32(5.0)
    o, lo, hi = oracle_mod.shell_orientation(np.concatenate([cube, far, plane]))
    assert o.tolist() == [1.0] * 12 + [-1.0] * 12 + [0.0] * 2
    assert np.all(hi[:12] < 1.0) and np.all(lo[12:24] > 4.0)


def test_two_shells_of_opposite_winding_render_like_two_sided(oracle_mod, monkeypatch):
    # one orientation per object culled the near faces of the
    # minority shell.  Per-shell orientation must leave the image of a
    # two-shell, mixed-winding object equal to the render without S6b.
    cube, cuv = oracle_mod.parse_obj(CUBE)
    shifted = cube[:, ::-1] + np.array([2.5, 0.0, 0.4], np.float32)
    verts = np.concatenate([cube, shifted]).reshape(-1, 3)
    d = scenes.SceneDesc(
        num_worlds=1, width=64, height=64,
        mesh_vertices=verts, mesh_uvs=np.zeros((len(verts), 2), np.float32),
        mesh_indices=np.arange(len(verts), dtype=np.uint32),
        mesh_vertex_offsets=np.array([0], np.uint32),
        mesh_indices_offsets=np.array([0], np.uint32),
        mesh_materials=np.array([-1], np.int32),
        instances=[((-1.0, 6.0, 0.0), (0.9238795, 0.0, 0.0, 0.3826834), (1.5, 1.5, 1.5), 0)],
        cameras=[((0.0, 0.0, 0.5), (1.0, 0.0, 0.0, 0.0))], worlds=[(1, 0, 1, 0)])
    fs = oracle_mod.FlatScene(d)
    assert fs.tri_orient.tolist() == [1.0] * 12 + [-1.0] * 12
    with_cull = fs.render()
    fs.tri_orient[:] = 0.0                  # S6b off: plain two-sided rule
    without = fs.render()
    assert (with_cull["tri_id"][0] >= 12).any() and (with_cull["tri_id"][0] >= 0).sum() > 100
    assert np.array_equal(with_cull["rgb"], without["rgb"])
    assert np.array_equal(with_cull["depth"], without["depth"])


def test_back_faces_stay_when_the_near_plane_can_cut_the_front(oracle_mod):
    # the eye sits just outside the cube's box, closer to its front face than
    # znear (Raytracer: 0.1): the front face fails the near test, so the far
    # wall must show -- culling it would leave background
    ident = (1.0, 0.0, 0.0, 0.0)
    d = _one_world([((0.0, 1.05, 0.0), ident, (2.0, 2.0, 2.0), 0)], [(CUBE, -1)],
                   cam=((0.0, 0.0, 0.0), ident))
    d.render_mode = "Raytracer"
    fs = oracle_mod.FlatScene(d)
    a = fs.render()
    fs.tri_orient[:] = 0.0
    b = fs.render()
    assert np.array_equal(a["tri_id"], b["tri_id"]) and np.array_equal(a["depth"], b["depth"])
    centre = a["depth"][0, 32, 32]
    assert abs(centre - 2.05) < 1e-5          # the far wall (y = 2.05), not the near one (0.05)


def test_back_faces_culled_outside_but_not_inside(oracle_mod):
    ident = (1.0, 0.0, 0.0, 0.0)
    box = [((0.0, 6.0, 0.0), ident, (2.0, 2.0, 2.0), 0)]
    outside = render_oracle(_one_world(box, [(CUBE, -1)], cam=((0.0, 0.0, 0.0), ident)))
    ids = outside["tri_id"][0]
    # seen face-on from -Y only the two triangles of the y = -0.5 face survive
    assert set(np.unique(ids[ids >= 0])) == {0, 1}
    np.testing.assert_allclose(outside["depth"][0][ids >= 0], 5.0, rtol=1e-6)
    # eye inside the cube: every pixel sees an inner wall (those are back faces)
    inside = render_oracle(_one_world(box, [(CUBE, -1)], cam=((0.0, 6.0, 0.0), ident)))
    assert (inside["tri_id"][0] >= 0).all()
    # a mirrored instance (negative scale) flips the winding, not the image
    mirrored = [((0.0, 6.0, 0.0), ident, (-2.0, 2.0, 2.0), 0)]
    m = render_oracle(_one_world(mirrored, [(CUBE, -1)], cam=((0.0, 0.0, 0.0), ident)))
    assert np.array_equal(m["tri_id"][0] >= 0, ids >= 0)
    np.testing.assert_allclose(m["depth"][0], outside["depth"][0], rtol=1e-6)
