/* Sanitizer driver for the CPU oracle (test infrastructure, like the oracle itself):
 * builds a small scene in memory -- a closed cube, an open quad, a degenerate triangle, a
 * textured material, worlds with hidden, unbound and out-of-range instances, cameras inside
 * and outside the geometry -- and renders it in both storage orders with every output
 * enabled.  Built with -fsanitize=address,undefined by `make -C oracle asan`; run by
 * tests/test_sanitizers.py (CPU suite).  Prints a checksum so the work cannot be elided. */
#include "raster_oracle.c"

#include <stdio.h>
#include <stdlib.h>

static const float CUBE[8][3] = { { -1, -1, -1 }, { 1, -1, -1 }, { 1, 1, -1 }, { -1, 1, -1 },
                                  { -1, -1, 1 },  { 1, -1, 1 },  { 1, 1, 1 },  { -1, 1, 1 } };
static const int QUADS[6][4] = { { 0, 3, 2, 1 }, { 4, 5, 6, 7 }, { 0, 1, 5, 4 },
                                 { 2, 3, 7, 6 }, { 1, 2, 6, 5 }, { 3, 0, 4, 7 } };

int main(void)
{
    enum { T = 12 + 2 + 1, O = 3, I = 9, V = 4, W = 3, RES = 48 };
    float tri_pos[T][3][3], tri_uv[T][3][2], orient[T], bbmin[T][3], bbmax[T][3];
    int32_t tri_mat[T];
    int t = 0;
    for (int q = 0; q < 6; ++q)
        for (int h = 0; h < 2; ++h, ++t) {
            const int idx[3] = { QUADS[q][0], QUADS[q][1 + h], QUADS[q][2 + h] };
            for (int c = 0; c < 3; ++c) {
                for (int a = 0; a < 3; ++a)
                    tri_pos[t][c][a] = CUBE[idx[c]][a];
                tri_uv[t][c][0] = 0.5f * (CUBE[idx[c]][0] + 1.0f) * 3.0f;
                tri_uv[t][c][1] = 0.5f * (CUBE[idx[c]][1] + CUBE[idx[c]][2]);
            }
            tri_mat[t] = 1;
            orient[t] = 1.0f;
            for (int a = 0; a < 3; ++a) { bbmin[t][a] = -1.0002f; bbmax[t][a] = 1.0002f; }
        }
    /* an open quad (two-sided) ... */
    const float Q[4][3] = { { -50, -50, 0 }, { 50, -50, 0 }, { 50, 50, 0 }, { -50, 50, 0 } };
    for (int h = 0; h < 2; ++h, ++t) {
        const int idx[3] = { 0, 1 + h, 2 + h };
        for (int c = 0; c < 3; ++c) {
            for (int a = 0; a < 3; ++a)
                tri_pos[t][c][a] = Q[idx[c]][a];
            tri_uv[t][c][0] = Q[idx[c]][0];
            tri_uv[t][c][1] = Q[idx[c]][1];
        }
        tri_mat[t] = 0;
        orient[t] = 0.0f;
        for (int a = 0; a < 3; ++a) { bbmin[t][a] = -51.0f; bbmax[t][a] = 51.0f; }
    }
    /* ... and a degenerate triangle (all three corners on a line), material out of range */
    for (int c = 0; c < 3; ++c) {
        tri_pos[t][c][0] = (float)c; tri_pos[t][c][1] = (float)c; tri_pos[t][c][2] = (float)c;
        tri_uv[t][c][0] = tri_uv[t][c][1] = 0.0f;
    }
    tri_mat[t] = 7;
    orient[t] = 0.0f;
    for (int a = 0; a < 3; ++a) { bbmin[t][a] = 0.0f; bbmax[t][a] = 2.0f; }
    const int32_t obj_first[O] = { 0, 12, 14 }, obj_num[O] = { 12, 2, 1 };
    const float mat_color[2][4] = { { 0.3f, 0.6f, 0.3f, 1.0f }, { 1.0f, 0.9f, 0.8f, 1.0f } };
    const int32_t mat_tex[2] = { -1, 0 };
    uint8_t tex[4 * 4][4];
    for (int i = 0; i < 16; ++i) {
        tex[i][0] = (uint8_t)(i * 16); tex[i][1] = (uint8_t)(255 - i * 13); tex[i][2] = (uint8_t)(i * 7); tex[i][3] = 255;
    }
    const int64_t tex_off[1] = { 0 };
    const int32_t tex_w[1] = { 4 }, tex_h[1] = { 4 };
    /* three worlds: (plane, cube, degenerate), (cube hidden, cube mirrored, unbound row), (object id out of range, ...) */
    const float ipos[I][3] = { { 0, 0, 0 }, { 0, 6, 1 }, { 1, 5, 1 }, { 0, 6, 1 }, { -2, 7, 1.5f }, { 0, 0, 0 },
                               { 0, 5, 0 }, { 0, 0, 0 }, { 0.2f, 0.1f, 0.3f } };
    const float irot[I][4] = { { 1, 0, 0, 0 }, { 0.9239f, 0, 0, 0.3827f }, { 1, 0, 0, 0 }, { 1, 0, 0, 0 },
                               { 0.7071f, 0.7071f, 0, 0 }, { 1, 0, 0, 0 }, { 1, 0, 0, 0 }, { 1, 0, 0, 0 }, { 0.5f, 0.5f, 0.5f, 0.5f } };
    const float iscl[I][3] = { { 1, 1, 1 }, { 1, 2, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, -1.5f, 1 }, { 1, 1, 1 },
                               { 1, 1, 1 }, { 1, 1, 1 }, { 40, 40, 40 } };
    const int32_t iobj0[I] = { 1, 0, 2, 0, 0, -1, 9, 1, 0 };
    const int32_t iobj[I] = { 1, 0, 2, -1, 0, -1, 9, 1, 0 };
    const int32_t wstart[W + 1] = { 0, 3, 6, 9 };
    const float cpos[V][3] = { { 0, -4, 2 }, { 0, 6, 1 }, { 3, 1, 4 }, { 0, 0, 0.5f } };   /* [1]: inside the cube */
    const float crot[V][4] = { { 1, 0, 0, 0 }, { 1, 0, 0, 0 }, { 0.9659f, -0.2588f, 0, 0 }, { 0.7071f, 0, 0, 0.7071f } };
    const int32_t vworld[V] = { 0, 0, 1, 2 };
    unsigned long long sum = 0;
    for (int transposed = 0; transposed < 2; ++transposed) {
        orc_scene s;
        memset(&s, 0, sizeof s);
        s.tri_pos = &tri_pos[0][0][0]; s.tri_uv = &tri_uv[0][0][0]; s.tri_mat = tri_mat;
        s.obj_first_tri = obj_first; s.obj_num_tris = obj_num; s.num_objects = O;
        s.tri_orient = orient; s.tri_bbmin = &bbmin[0][0]; s.tri_bbmax = &bbmax[0][0];
        s.mat_color = &mat_color[0][0]; s.mat_tex = mat_tex; s.num_materials = 2;
        s.tex_data = &tex[0][0]; s.tex_offset = tex_off; s.tex_w = tex_w; s.tex_h = tex_h; s.num_textures = 1;
        s.inst_pos = &ipos[0][0]; s.inst_rot = &irot[0][0]; s.inst_scale = &iscl[0][0];
        s.inst_obj = iobj; s.inst_obj0 = iobj0; s.world_inst_start = wstart;
        s.cam_pos = &cpos[0][0]; s.cam_rot = &crot[0][0]; s.view_world = vworld; s.num_views = V;
        s.width = RES; s.height = transposed ? RES : RES - 11;          /* ragged height in raster mode */
        const float th = 1.0f, asp = (float)s.width / (float)s.height;
        s.sx = 2.0f * th * asp / (float)s.width; s.ox = (1.0f / (float)s.width - 1.0f) * th * asp;
        s.sz = -2.0f * th / (float)s.height;     s.oz = (1.0f - 1.0f / (float)s.height) * th;
        s.inv_near = transposed ? 10.0f : 1000.0f;
        s.inv_far = transposed ? 0.001f : 0.0f;
        s.s6b_pad = 0.002f;
        s.to_light[0] = -0.7f; s.to_light[1] = 0.7f; s.to_light[2] = 0.14f;
        s.ambient = 0.25f; s.diffuse = 0.75f;
        s.default_color[0] = s.default_color[1] = s.default_color[2] = s.default_color[3] = 1.0f;
        s.transposed = transposed;
        const size_t px = (size_t)V * s.width * s.height;
        uint8_t *rgb = malloc(px * 4);
        float *depth = malloc(px * 4);
        int32_t *ids = malloc(px * 4), *seg = malloc(px * 4);
        if (!rgb || !depth || !ids || !seg)
            return 2;
        orc_render(&s, 0, V, rgb, depth, ids, seg, 2);
        orc_render(&s, 1, 3, rgb, depth, NULL, NULL, 1);                /* a view range, no ids */
        int covered = 0;
        for (size_t i = 0; i < px; ++i) {
            sum = sum * 1099511628211ull + rgb[4 * i] + 3u * rgb[4 * i + 1] + (unsigned)(ids[i] + 1) + (unsigned)(seg[i] + 2);
            covered += ids[i] >= 0;
        }
        printf("%s: %d of %zu pixels covered\n", transposed ? "raytracer" : "rasterizer", covered, px);
        if (covered == 0)
            return 3;
        free(rgb); free(depth); free(ids); free(seg);
    }
    printf("checksum %llx\n", sum);
    return 0;
}
