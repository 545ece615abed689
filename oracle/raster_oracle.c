/*
 * raster_oracle.c -- CPU restatement of the batch-render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported
 * by, or executed from the product path (libmrx_hip.so / madrona_renderer).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker.
 *
 * PARITY UNPINNED: the reference (llGuy/madrona_renderer @ 2024_10_08) holds
 * no rendering arithmetic -- every pixel is produced by the un-vendored
 * submodule external/madrona (.gitmodules:1-3, pinned SHA not recoverable)
 * and the reference ships no tests, golden images or fixtures
 * (SURVEY.md section 0 F1/F2/F7, section 8c).  What this file follows instead
 * is the rendering spec of DESIGN.md section 3, whose conventions are anchored
 * on the things the reference does pin:
 *   - table layouts / world assembly   src/sim.hpp:31-50,76-82, src/sim.cpp:135-176
 *   - vfov 90 deg, znear 0.001         src/sim.cpp:168-171
 *   - RT near 0.1 / far 1000, square   src/mgr.cpp:443-479
 *   - one directional light (1,-1,-.05) src/mgr.cpp:356-359
 *   - output shapes and dtypes         src/mgr.cpp:547-605
 *   - object / material / texture order src/mgr.cpp:214-363
 *   - RT callers read storage as [x][y] scripts/test.py:160, src/dump.cpp:9-21
 * and it is pinned by the known-answer anchors derived from the reference's
 * own scene constants (scripts/test.py:36-55), checked
 * in tests/test_oracle_anchors.py.
 *
 * Arithmetic contract: IEEE-754 binary32, round-to-nearest,
 * no contraction (-ffp-contract=off) except where fmaf() is written, no
 * fast-math, correctly rounded / and sqrtf.  The scalar loops below are the
 * literal op order of the spec; the HIP kernels restate the same order in a
 * separate translation unit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    /* object-space triangle soup, shared by every world */
    const float   *tri_pos;        /* [T][3 verts][xyz]                       */
    const float   *tri_uv;         /* [T][3 verts][uv]                        */
    const int32_t *tri_mat;        /* [T] material index, -1 = none           */
    const int32_t *obj_first_tri;  /* [O]                                     */
    const int32_t *obj_num_tris;   /* [O]                                     */
    int32_t        num_objects;
    /* S6b, per triangle, of its shell (edge-connected component of the object):
     * +1 / -1 for a closed, consistently wound shell (volume sign), else 0;
     * the shell's padded object-space bounding box */
    const float   *tri_orient;     /* [T]                                     */
    const float   *tri_bbmin;      /* [T][3]                                  */
    const float   *tri_bbmax;      /* [T][3]                                  */
    /* materials / textures */
    const float   *mat_color;      /* [M][rgba]                               */
    const int32_t *mat_tex;        /* [M] texture index, -1 = none            */
    int32_t        num_materials;
    const uint8_t *tex_data;       /* RGBA8 texels, all textures concatenated */
    const int64_t *tex_offset;     /* [X] texel offset of texture X           */
    const int32_t *tex_w;          /* [X]                                     */
    const int32_t *tex_h;          /* [X]                                     */
    int32_t        num_textures;
    /* per-world pose state, world-major (src/sim.cpp:146-175 copies) */
    const float   *inst_pos;       /* [I][3]                                  */
    const float   *inst_rot;       /* [I][4] w,x,y,z                          */
    const float   *inst_scale;     /* [I][3]                                  */
    const int32_t *inst_obj;       /* [I] ObjectID now: negative = hidden     */
    const int32_t *inst_obj0;      /* [I] ObjectID at creation: binds the geometry and the
                                    * world-local triangle slots (MRX_BUF_INSTANCE_OBJECT) */
    const int32_t *world_inst_start; /* [W+1] prefix sum of numInstances      */
    const float   *cam_pos;        /* [V][3]                                  */
    const float   *cam_rot;        /* [V][4]                                  */
    const int32_t *view_world;     /* [V] world of view v                     */
    int32_t        num_views;
    int32_t        width, height;
    /* pixel -> ray constants, computed on the host */
    float sx, ox, sz, oz;
    float inv_near, inv_far;
    float s6b_pad;                 /* S6b: reach of the near plane            */
    float to_light[3];             /* unit vector towards the light, world    */
    float ambient, diffuse;
    float default_color[4];        /* material for tri_mat == -1              */
    int32_t transposed;            /* 1: Raytracer-mode [x][y] storage        */
} orc_scene;

typedef struct {
    float A[3], B[3], C[3];        /* edge functions, inside <=> all >= 0     */
    float Dx, Dy, Dc;              /* 1/depth plane                           */
    float Ua, Ub, Uc, Va, Vb, Vc;  /* (u,v)/depth planes                      */
    float lit[3];                  /* lighting x material colour              */
    uint32_t rgba;                 /* the same, packed, for untextured tris   */
    int32_t tex;                   /* texture index or -1                     */
    int32_t seg;                   /* objectID of the owning instance         */
    int32_t k;                     /* world-local triangle index              */
} orc_tri;

/* S1: rotation matrix of q = (w,x,y,z), used as given (no normalisation). */
static void quat_to_mat(const float q[4], float R[3][3])
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, yy = y * y2, zz = z * z2;
    float xy = x * y2, xz = x * z2, yz = y * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    R[0][0] = 1.0f - (yy + zz); R[0][1] = xy - wz;          R[0][2] = xz + wy;
    R[1][0] = xy + wz;          R[1][1] = 1.0f - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy;          R[2][1] = yz + wx;          R[2][2] = 1.0f - (xx + yy);
}

/* S2: every 3-term dot product is one rounded product and two fused steps */
static inline float dot3(float ax, float ay, float az,
                         float bx, float by, float bz)
{
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static inline uint32_t to_u8(float c)
{
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint32_t)fmaf(c, 255.0f, 0.5f);
}

/* S3-S8: set up every triangle of the world of view v.  Returns the count of
 * records written; culled triangles keep their world-local index k. */
static int setup_view(const orc_scene *s, int v, orc_tri *out)
{
    const int w = s->view_world[v];
    float Rc[3][3];
    quat_to_mat(&s->cam_rot[4 * v], Rc);
    const float *c = &s->cam_pos[3 * v];
    /* S4: light into view space, lv = Rc^T l */
    float lv[3];
    for (int r = 0; r < 3; ++r)
        lv[r] = dot3(Rc[0][r], Rc[1][r], Rc[2][r],
                     s->to_light[0], s->to_light[1], s->to_light[2]);

    int n = 0, k = 0;
    for (int i = s->world_inst_start[w]; i < s->world_inst_start[w + 1]; ++i) {
        const int obj = s->inst_obj0[i];
        if (obj < 0 || obj >= s->num_objects)
            continue;
        if (s->inst_obj[i] < 0) {                 /* hidden this step: its slots stay empty */
            k += s->obj_num_tris[obj];
            continue;
        }
        float Ri[3][3], M[3][3], MV[3][3], tv[3];
        quat_to_mat(&s->inst_rot[4 * i], Ri);
        const float *sc = &s->inst_scale[3 * i];
        const float *t = &s->inst_pos[3 * i];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)
                M[r][cc] = Ri[r][cc] * sc[cc];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)
                MV[r][cc] = dot3(Rc[0][r], Rc[1][r], Rc[2][r],
                                 M[0][cc], M[1][cc], M[2][cc]);
        float dt[3] = { t[0] - c[0], t[1] - c[1], t[2] - c[2] };
        for (int r = 0; r < 3; ++r)
            tv[r] = dot3(Rc[0][r], Rc[1][r], Rc[2][r], dt[0], dt[1], dt[2]);

        /* S6b: the eye in the instance's unscaled frame, q = Ri^T (c - t) */
        float qo[3];
        for (int r = 0; r < 3; ++r)
            qo[r] = -dot3(Ri[0][r], Ri[1][r], Ri[2][r], dt[0], dt[1], dt[2]);
        const float det = (sc[0] * sc[1]) * sc[2];   /* mirroring flips the winding */
        const int first = s->obj_first_tri[obj];
        const int cnt = s->obj_num_tris[obj];
        for (int ti = first; ti < first + cnt; ++ti, ++k) {
            const float *op = &s->tri_pos[9 * ti];
            const float *uv = &s->tri_uv[6 * ti];
            float P[3][3];
            for (int j = 0; j < 3; ++j)
                for (int r = 0; r < 3; ++r)
                    P[j][r] = fmaf(MV[r][2], op[3 * j + 2],
                                   fmaf(MV[r][1], op[3 * j + 1],
                                        fmaf(MV[r][0], op[3 * j], tv[r])));
            float N[3][3], e1[3], e2[3], nn[3];
            cross3(P[1], P[2], N[0]);
            cross3(P[2], P[0], N[1]);
            cross3(P[0], P[1], N[2]);
            for (int r = 0; r < 3; ++r) {
                e1[r] = P[1][r] - P[0][r];
                e2[r] = P[2][r] - P[0][r];
            }
            cross3(e1, e2, nn);
            const float d = dot3(nn[0], nn[1], nn[2], P[0][0], P[0][1], P[0][2]);
            if (!(fabsf(d) > 0.0f))
                continue;                                   /* S6: cull */
            {
                /* S6b: eye outside the shell's padded box (scaled by s) by more
                 * than the reach of the near plane -> faces turned away from it
                 * can never be the nearest hit */
                int outside = 0;
                for (int r = 0; r < 3; ++r) {
                    const float b0 = s->tri_bbmin[3 * ti + r] * sc[r];
                    const float b1 = s->tri_bbmax[3 * ti + r] * sc[r];
                    if (qo[r] < fminf(b0, b1) - s->s6b_pad || qo[r] > fmaxf(b0, b1) + s->s6b_pad)
                        outside = 1;
                }
                const float handed = s->tri_orient[ti] * det;
                if (outside && ((handed > 0.0f && d > 0.0f) || (handed < 0.0f && d < 0.0f)))
                    continue;
            }
            orc_tri *o = &out[n++];
            const float flip = d < 0.0f ? -1.0f : 1.0f;
            for (int e = 0; e < 3; ++e) {
                float ax = N[e][0] * s->sx;
                float az = N[e][2] * s->sz;
                float cc = fmaf(N[e][2], s->oz, fmaf(N[e][0], s->ox, N[e][1]));
                o->A[e] = (s->transposed ? az : ax) * flip;
                o->B[e] = (s->transposed ? ax : az) * flip;
                o->C[e] = cc * flip;
            }
            const float rd = 1.0f / d;
            {
                float ax = (nn[0] * s->sx) * rd;
                float az = (nn[2] * s->sz) * rd;
                o->Dx = s->transposed ? az : ax;
                o->Dy = s->transposed ? ax : az;
                o->Dc = fmaf(nn[2], s->oz, fmaf(nn[0], s->ox, nn[1])) * rd;
            }
            const float rad = fabsf(rd);
            o->Ua = fmaf(uv[4], o->A[2], fmaf(uv[2], o->A[1], uv[0] * o->A[0])) * rad;
            o->Ub = fmaf(uv[4], o->B[2], fmaf(uv[2], o->B[1], uv[0] * o->B[0])) * rad;
            o->Uc = fmaf(uv[4], o->C[2], fmaf(uv[2], o->C[1], uv[0] * o->C[0])) * rad;
            o->Va = fmaf(uv[5], o->A[2], fmaf(uv[3], o->A[1], uv[1] * o->A[0])) * rad;
            o->Vb = fmaf(uv[5], o->B[2], fmaf(uv[3], o->B[1], uv[1] * o->B[0])) * rad;
            o->Vc = fmaf(uv[5], o->C[2], fmaf(uv[3], o->C[1], uv[1] * o->C[0])) * rad;
            /* S7: flat two-sided Lambert */
            const float len = sqrtf(dot3(nn[0], nn[1], nn[2], nn[0], nn[1], nn[2]));
            float ndl = dot3(nn[0], nn[1], nn[2], lv[0], lv[1], lv[2]) / len;
            if (d > 0.0f)
                ndl = -ndl;
            const float lit = fmaf(s->diffuse, fmaxf(ndl, 0.0f), s->ambient);
            const int m = s->tri_mat[ti];
            const float *col = (m >= 0 && m < s->num_materials)
                                   ? &s->mat_color[4 * m] : s->default_color;
            int tex = (m >= 0 && m < s->num_materials) ? s->mat_tex[m] : -1;
            if (tex < 0 || tex >= s->num_textures)
                tex = -1;
            for (int ch = 0; ch < 3; ++ch)
                o->lit[ch] = lit * col[ch];
            o->rgba = to_u8(o->lit[0]) | (to_u8(o->lit[1]) << 8) |
                      (to_u8(o->lit[2]) << 16) | 0xFF000000u;
            o->tex = tex;
            o->seg = obj;                         /* the bound object: label and geometry always agree */
            o->k = k;
        }
    }
    return n;
}

static int max_world_tris(const orc_scene *s)
{
    int best = 0;
    for (int v = 0; v < s->num_views; ++v) {
        int w = s->view_world[v], cnt = 0;
        for (int i = s->world_inst_start[w]; i < s->world_inst_start[w + 1]; ++i) {
            int obj = s->inst_obj0[i];
            if (obj >= 0 && obj < s->num_objects)
                cnt += s->obj_num_tris[obj];
        }
        if (cnt > best)
            best = cnt;
    }
    return best;
}

__attribute__((target_clones("fma", "default")))
static void render_view(const orc_scene *s, int v, orc_tri *tris,
                        uint8_t *rgb, float *depth, int32_t *tri_id,
                        int32_t *seg)
{
    const int n = setup_view(s, v, tris);
    /* storage is [view][slow][fast]; raster: fast = image x, slow = image y;
     * Raytracer mode: fast = image y, slow = image x (S9). */
    const int nfast = s->transposed ? s->height : s->width;
    const int nslow = s->transposed ? s->width : s->height;
    const size_t base = (size_t)v * nfast * nslow;
    for (int fy = 0; fy < nslow; ++fy) {
        for (int fx = 0; fx < nfast; ++fx) {
            const float px = (float)fx, py = (float)fy;
            float best = s->inv_far;
            int hit = -1;
            for (int t = 0; t < n; ++t) {
                const orc_tri *o = &tris[t];
                float e0 = fmaf(o->A[0], px, fmaf(o->B[0], py, o->C[0]));
                float e1 = fmaf(o->A[1], px, fmaf(o->B[1], py, o->C[1]));
                float e2 = fmaf(o->A[2], px, fmaf(o->B[2], py, o->C[2]));
                if (!(fminf(fminf(e0, e1), e2) >= 0.0f))
                    continue;
                float it = fmaf(o->Dx, px, fmaf(o->Dy, py, o->Dc));
                if (it > best && it <= s->inv_near) {
                    best = it;
                    hit = t;
                }
            }
            const size_t p = base + (size_t)fy * nfast + fx;
            uint32_t rgba = 0xFF000000u;
            float dep = 0.0f;
            int32_t tid = -1, sg = -1;
            if (hit >= 0) {
                const orc_tri *o = &tris[hit];
                const float tt = 1.0f / best;
                dep = tt;
                tid = o->k;
                sg = o->seg;
                if (o->tex < 0) {
                    rgba = o->rgba;
                } else {
                    /* S8: nearest texel, repeat addressing, v up */
                    float u = fmaf(o->Ua, px, fmaf(o->Ub, py, o->Uc)) * tt;
                    float vv = fmaf(o->Va, px, fmaf(o->Vb, py, o->Vc)) * tt;
                    const int tw = s->tex_w[o->tex], th = s->tex_h[o->tex];
                    float uf = u - floorf(u);
                    float vf = vv - floorf(vv);
                    vf = 1.0f - vf;
                    int tx = (int)(uf * (float)tw);
                    int ty = (int)(vf * (float)th);
                    if (tx > tw - 1) tx = tw - 1;
                    if (ty > th - 1) ty = th - 1;
                    if (tx < 0) tx = 0;
                    if (ty < 0) ty = 0;
                    const uint8_t *tp = s->tex_data +
                        4 * (s->tex_offset[o->tex] + (int64_t)ty * tw + tx);
                    uint32_t r = to_u8(((float)tp[0] * (1.0f / 255.0f)) * o->lit[0]);
                    uint32_t g = to_u8(((float)tp[1] * (1.0f / 255.0f)) * o->lit[1]);
                    uint32_t b = to_u8(((float)tp[2] * (1.0f / 255.0f)) * o->lit[2]);
                    rgba = r | (g << 8) | (b << 16) | 0xFF000000u;
                }
            }
            if (rgb) memcpy(rgb + 4 * p, &rgba, 4);
            if (depth) depth[p] = dep;
            if (tri_id) tri_id[p] = tid;
            if (seg) seg[p] = sg;
        }
    }
}

/* Render views [view_begin, view_end).  Output pointers address view 0 of the
 * full tensors; any of them may be NULL.  num_threads <= 0: all cores.
 * Returns the number of threads used. */
int orc_render(const orc_scene *s, int view_begin, int view_end,
               uint8_t *rgb, float *depth, int32_t *tri_id, int32_t *seg,
               int num_threads)
{
    const int maxt = max_world_tris(s);
    int used = 1;
#ifdef _OPENMP
    if (num_threads <= 0)
        num_threads = omp_get_num_procs();
    used = num_threads;
#pragma omp parallel num_threads(num_threads)
#endif
    {
        orc_tri *tris = (orc_tri *)malloc(sizeof(orc_tri) * (size_t)(maxt > 0 ? maxt : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int v = view_begin; v < view_end; ++v)
            render_view(s, v, tris, rgb, depth, tri_id, seg);
        free(tris);
    }
    return used;
}

int orc_num_procs(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

int orc_abi_version(void) { return 1; }
