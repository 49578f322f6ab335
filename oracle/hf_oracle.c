/*
 * hf_oracle.c -- CPU ORACLE (test infrastructure; see hf_oracle.h header).
 *
 * Scalar restatement, in plain C, of the reference's mesh semantics applied
 * to the heightfield tessellation.  Every function cites the reference
 * file:line it follows.  Build: see oracle/Makefile (-ffp-contract=off).
 *
 * Geometry conventions (build decisions, frozen in DESIGN.md section 2):
 *   object space = Rectangle's: x,y in [-1,1], +Z up (src/shapes/rectangle.cpp:47-48)
 *   vertex (row i, col j): x = fma(j, 2/(W-1), -1), y = fma(i, 2/(H-1), -1), z = h[i*W+j]*max_height
 *   cell (cx,cy) has corners vXY = vertex(cy+Y, cx+X); it is split along the
 *   v10-v01 diagonal into
 *       tri 0 = (v00, v10, v01)        tri 1 = (v11, v01, v10)
 *   (both wound so that n.z > 0 in object space); prim_index = 2*(cy*(W-1)+cx)+tri.
 *   prim_uv = Moeller-Trumbore (u,v) = barycentric weights of p1, p2.
 *   vertex texcoord = (j/(W-1), i/(H-1)).
 */
#include "hf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct hfo_field {
    int W, H;
    float *h;
    float s;
    float to_world[12], to_object[12];
    int flip_normals;
    float sx, sy, iu, iv;
    int top;          /* levels 1..top hold (min,max) of 2^l x 2^l cell blocks */
    float **mip;      /* mip[l][2*(iy*mw[l]+ix) + {0,1}] */
    int *mw, *mh;
    float zmin, zmax; /* range of h*s over the whole grid */
};

/* ------------------------------------------------------------------------ */
/* Dr.Jit array-op restatements (see header)                                  */
/* ------------------------------------------------------------------------ */
static inline float dot3(const float a[3], const float b[3]) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline float dot2(const float a[2], const float b[2]) {
    return fmaf(a[1], b[1], a[0] * b[0]);
}
static inline void cross3(const float a[3], const float b[3], float r[3]) {
    float r0 = fmaf(a[1], b[2], -(a[2] * b[1]));
    float r1 = fmaf(a[2], b[0], -(a[0] * b[2]));
    float r2 = fmaf(a[0], b[1], -(a[1] * b[0]));
    r[0] = r0; r[1] = r1; r[2] = r2;
}
static inline float rcpf(float x) { return 1.0f / x; }
static inline float rsqrtf_(float x) { return 1.0f / sqrtf(x); }
static inline void sub3(const float a[3], const float b[3], float r[3]) {
    r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2];
}
static inline void normalize3(const float v[3], float r[3]) {
    float il = rsqrtf_(dot3(v, v));
    r[0] = v[0] * il; r[1] = v[1] * il; r[2] = v[2] * il;
}

/* Transform::transform_affine(Point), include/mitsuba/core/transform.h:104-111:
 * result = column 3; result = fmadd(column i, arg[i], result) for i = 0,1,2.
 * m is row-major 3x4, so column i is (m[i], m[4+i], m[8+i]). */
static inline void xform_point(const float *m, const float p[3], float r[3]) {
    for (int k = 0; k < 3; ++k) {
        float acc = m[4 * k + 3];
        acc = fmaf(m[4 * k + 0], p[0], acc);
        acc = fmaf(m[4 * k + 1], p[1], acc);
        acc = fmaf(m[4 * k + 2], p[2], acc);
        r[k] = acc;
    }
}
/* Transform::operator*(Vector), transform.h:130-138: column 0 * x, then fmadd. */
static inline void xform_vec(const float *m, const float v[3], float r[3]) {
    for (int k = 0; k < 3; ++k) {
        float acc = m[4 * k + 0] * v[0];
        acc = fmaf(m[4 * k + 1], v[1], acc);
        acc = fmaf(m[4 * k + 2], v[2], acc);
        r[k] = acc;
    }
}

/* ------------------------------------------------------------------------ */
/* Geometry                                                                   */
/* ------------------------------------------------------------------------ */
void hfo_vertex(const hfo_field *f, int i, int j, float out[3]) {
    out[0] = fmaf((float) j, f->sx, -1.0f);
    out[1] = fmaf((float) i, f->sy, -1.0f);
    out[2] = f->h[(size_t) i * f->W + j] * f->s;
}

/* grid indices (row, col) of the three vertices of a primitive */
static inline void prim_vertex_ids(const hfo_field *f, uint32_t prim, int vi[3], int vj[3]) {
    uint32_t cell = prim >> 1;
    int cw = f->W - 1;
    int cx = (int) (cell % (uint32_t) cw), cy = (int) (cell / (uint32_t) cw);
    if ((prim & 1u) == 0) {          /* (v00, v10, v01) */
        vi[0] = cy;     vj[0] = cx;
        vi[1] = cy;     vj[1] = cx + 1;
        vi[2] = cy + 1; vj[2] = cx;
    } else {                          /* (v11, v01, v10) */
        vi[0] = cy + 1; vj[0] = cx + 1;
        vi[1] = cy + 1; vj[1] = cx;
        vi[2] = cy;     vj[2] = cx + 1;
    }
}

/* moeller_trumbore(), include/mitsuba/render/mesh.h:357-380 (operation order kept) */
static inline int moeller_trumbore(const float o[3], const float d[3], float maxt,
                                   const float p0[3], const float p1[3], const float p2[3],
                                   float *t_out, float *u_out, float *v_out) {
    float e1[3], e2[3], pvec[3], tvec[3], qvec[3];
    sub3(p1, p0, e1);
    sub3(p2, p0, e2);
    cross3(d, e2, pvec);
    float inv_det = rcpf(dot3(e1, pvec));
    sub3(o, p0, tvec);
    float u = dot3(tvec, pvec) * inv_det;
    int active = (u >= 0.f) && (u <= 1.f);
    cross3(tvec, e1, qvec);
    float v = dot3(d, qvec) * inv_det;
    active = active && (v >= 0.f) && (u + v <= 1.f);
    float t = dot3(e2, qvec) * inv_det;
    active = active && (t >= 0.f) && (t <= maxt);
    *t_out = t; *u_out = u; *v_out = v;
    return active;
}

typedef struct {
    float t, u, v;
    uint32_t prim;
    int hit;
} best_t;

/* Closest-hit update equivalent to the sequential `pi = prim_pi; ray.maxt = prim_pi.t`
 * loop of kdtree.h:2424-2448 visiting primitives in index order: minimum t, and among
 * exactly equal t the HIGHEST prim index (the later primitive passes `t <= maxt`). */
static inline void best_update(best_t *b, float t, float u, float v, uint32_t prim) {
    if (!b->hit || t < b->t || (t == b->t && prim > b->prim)) {
        b->t = t; b->u = u; b->v = v; b->prim = prim; b->hit = 1;
    }
}

/* test both triangles of one cell against the object-space ray */
static inline int test_cell(const hfo_field *f, int cx, int cy, const float o[3],
                            const float d[3], float maxt, best_t *b) {
    float v00[3], v10[3], v01[3], v11[3];
    hfo_vertex(f, cy, cx, v00);
    hfo_vertex(f, cy, cx + 1, v10);
    hfo_vertex(f, cy + 1, cx, v01);
    hfo_vertex(f, cy + 1, cx + 1, v11);
    uint32_t prim = 2u * ((uint32_t) cy * (uint32_t) (f->W - 1) + (uint32_t) cx);
    float t, u, v;
    int any = 0;
    if (moeller_trumbore(o, d, maxt, v00, v10, v01, &t, &u, &v)) { best_update(b, t, u, v, prim); any = 1; }
    if (moeller_trumbore(o, d, maxt, v11, v01, v10, &t, &u, &v)) { best_update(b, t, u, v, prim + 1u); any = 1; }
    return any;
}

/* ------------------------------------------------------------------------ */
/* Construction + min/max mips                                                */
/* ------------------------------------------------------------------------ */
static void free_mips(hfo_field *f) {
    if (f->mip) {
        for (int l = 1; l <= f->top; ++l) free(f->mip[l]);
        free(f->mip);
    }
    free(f->mw); free(f->mh);
    f->mip = NULL; f->mw = f->mh = NULL;
}

static void build_mips(hfo_field *f) {
    int cw = f->W - 1, ch = f->H - 1;
    int top = 0;
    while ((1 << top) < cw || (1 << top) < ch) ++top;
    f->top = top;
    f->mip = (float **) calloc((size_t) top + 1, sizeof(float *));
    f->mw = (int *) calloc((size_t) top + 1, sizeof(int));
    f->mh = (int *) calloc((size_t) top + 1, sizeof(int));
    /* global range */
    float zmin = INFINITY, zmax = -INFINITY;
    for (size_t k = 0; k < (size_t) f->W * f->H; ++k) {
        float z = f->h[k] * f->s;
        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    f->zmin = zmin; f->zmax = zmax;
    for (int l = 1; l <= top; ++l) {
        int w = (cw + (1 << l) - 1) >> l, h = (ch + (1 << l) - 1) >> l;
        f->mw[l] = w; f->mh[l] = h;
        f->mip[l] = (float *) malloc(sizeof(float) * 2 * (size_t) w * h);
        for (int iy = 0; iy < h; ++iy)
            for (int ix = 0; ix < w; ++ix) {
                float mn = INFINITY, mx = -INFINITY;
                if (l == 1) {
                    /* 2x2 cells = 3x3 vertices, clamped to the grid */
                    for (int i = 2 * iy; i <= 2 * iy + 2 && i < f->H; ++i)
                        for (int j = 2 * ix; j <= 2 * ix + 2 && j < f->W; ++j) {
                            float z = f->h[(size_t) i * f->W + j] * f->s;
                            mn = fminf(mn, z); mx = fmaxf(mx, z);
                        }
                } else {
                    int pw = f->mw[l - 1], ph = f->mh[l - 1];
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            int jy = 2 * iy + a, jx = 2 * ix + b;
                            if (jy < ph && jx < pw) {
                                const float *c = &f->mip[l - 1][2 * ((size_t) jy * pw + jx)];
                                mn = fminf(mn, c[0]); mx = fmaxf(mx, c[1]);
                            }
                        }
                }
                f->mip[l][2 * ((size_t) iy * w + ix) + 0] = mn;
                f->mip[l][2 * ((size_t) iy * w + ix) + 1] = mx;
            }
    }
}

hfo_field *hfo_create(int W, int H, const float *heights, float max_height,
                      const float *to_world, const float *to_object, int flip_normals) {
    if (W < 2 || H < 2 || !heights) return NULL; /* bitmap.cpp:280-283: H,W >= 2 */
    hfo_field *f = (hfo_field *) calloc(1, sizeof(hfo_field));
    f->W = W; f->H = H; f->s = max_height; f->flip_normals = flip_normals;
    memcpy(f->to_world, to_world, sizeof(float) * 12);
    memcpy(f->to_object, to_object, sizeof(float) * 12);
    f->sx = 2.0f / (float) (W - 1); f->sy = 2.0f / (float) (H - 1);
    f->iu = 1.0f / (float) (W - 1); f->iv = 1.0f / (float) (H - 1);
    f->h = (float *) malloc(sizeof(float) * (size_t) W * H);
    memcpy(f->h, heights, sizeof(float) * (size_t) W * H);
    build_mips(f);
    return f;
}

void hfo_set_heights(hfo_field *f, const float *heights) {
    memcpy(f->h, heights, sizeof(float) * (size_t) f->W * f->H);
    free_mips(f);
    build_mips(f);
}

void hfo_destroy(hfo_field *f) {
    if (!f) return;
    free_mips(f);
    free(f->h);
    free(f);
}

int hfo_num_levels(const hfo_field *f) { return f->top; }

int hfo_get_mip(const hfo_field *f, int level, float *out, int *w, int *h) {
    if (level < 1 || level > f->top) return 0;
    *w = f->mw[level]; *h = f->mh[level];
    if (out) memcpy(out, f->mip[level], sizeof(float) * 2 * (size_t) (*w) * (*h));
    return (*w) * (*h);
}

/* bbox(): world-space box of the 8 corners of the object-space bound
 * [x0,xW]x[y0,yH]x[zmin,zmax], as src/shapes/rectangle.cpp:114-124 does for its 4. */
void hfo_bbox(const hfo_field *f, float out[6]) {
    float lo[3] = { fmaf(0.f, f->sx, -1.f), fmaf(0.f, f->sy, -1.f), f->zmin };
    float hi[3] = { fmaf((float) (f->W - 1), f->sx, -1.f), fmaf((float) (f->H - 1), f->sy, -1.f), f->zmax };
    for (int k = 0; k < 3; ++k) { out[k] = INFINITY; out[3 + k] = -INFINITY; }
    for (int c = 0; c < 8; ++c) {
        float p[3] = { (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2] }, q[3];
        xform_point(f->to_world, p, q);
        for (int k = 0; k < 3; ++k) { out[k] = fminf(out[k], q[k]); out[3 + k] = fmaxf(out[3 + k], q[k]); }
    }
}

/* ------------------------------------------------------------------------ */
/* Preliminary intersection                                                   */
/* ------------------------------------------------------------------------ */
static inline int finite3(const float a[3]) { return isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2]); }

static void write_result(const best_t *b, float *t, float uv[2], uint32_t *prim) {
    /* miss: zero-initialised record with t = +inf (interaction.h:639-646, mesh.h:247) */
    if (b->hit) { *t = b->t; uv[0] = b->u; uv[1] = b->v; *prim = b->prim; }
    else        { *t = INFINITY; uv[0] = 0.f; uv[1] = 0.f; *prim = 0; }
}

/* ray_intersect_naive, kdtree.h:2424-2448 */
static void trace_naive(const hfo_field *f, const float o[3], const float d[3], float maxt,
                        int any_hit, best_t *b) {
    float oo[3], od[3];
    xform_point(f->to_object, o, oo);   /* rectangle.cpp:212: ray = to_object.transform_affine(ray_) */
    xform_vec(f->to_object, d, od);
    b->hit = 0; b->t = INFINITY; b->u = b->v = 0.f; b->prim = 0;
    for (int cy = 0; cy < f->H - 1; ++cy)
        for (int cx = 0; cx < f->W - 1; ++cx)
            if (test_cell(f, cx, cy, oo, od, maxt, b) && any_hit) return;
}

/*
 * Hierarchical traversal.  The set of cells it visits is a conservative
 * superset ("fat ray", margin m cells / mz in z) of the cells whose triangles
 * the ray can hit, and the per-triangle arithmetic and the tie rule are those
 * of the brute force, so the result is the brute force's -- independent of the
 * visiting order.  Order: cells are enumerated along a Morton curve mirrored so
 * that both direction components are >= 0 ("order space"); a node (X,Y,L) of
 * the implicit quadtree over 2^top x 2^top cells is skipped when the fat ray
 * segment [0, t_hi] misses its box [min z, max z] read from mip level L.
 */
/* traversal statistics of the last trace_hier() call on this thread (debug/tuning) */
static _Thread_local uint32_t g_stat_nodes, g_stat_leaves;

static void trace_hier(const hfo_field *f, const float o[3], const float d[3], float maxt,
                       int any_hit, best_t *b) {
    g_stat_nodes = 0; g_stat_leaves = 0;
    float oo[3], od[3];
    xform_point(f->to_object, o, oo);
    xform_vec(f->to_object, d, od);
    b->hit = 0; b->t = INFINITY; b->u = b->v = 0.f; b->prim = 0;
    if (!finite3(oo) || !finite3(od) || !(maxt >= 0.f)) return;

    const int cw = f->W - 1, ch = f->H - 1, top = f->top;
    const float hx = 0.5f * (float) cw, hy = 0.5f * (float) ch;

    /* The margins first (round 3): `reach` is measured with the entry into the bound inflated by the fixed amounts
     * only, also for a ray that misses that bound; then the bound itself (bbox.h:302-327) is inflated by what the
     * triangle test's noise can reach -- m cells in xy, m x (height span) in z -- like every node test below.
     * (Without it a ray traced from 40 units away lost a noise hit the brute force reports 1 % of the height span
     * above the bound: the one mismatch of 4e9 fuzz rays against the band brute force.) */
    const float zspan = fmaxf(f->zmax - f->zmin, fmaxf(fabsf(f->zmin), fabsf(f->zmax)));
    const float mz0 = 1e-5f * zspan + 1e-30f;
    float rr[3];
    float tin0 = 0.f;
    {
        const float lo[3] = { -1.f - 1e-4f, -1.f - 1e-4f, f->zmin - mz0 };
        const float hi[3] = {  1.f + 1e-4f,  1.f + 1e-4f, f->zmax + mz0 };
        for (int k = 0; k < 3; ++k) {
            rr[k] = 1.0f / od[k];
            if (od[k] != 0.f) tin0 = fmaxf(tin0, fminf((lo[k] - oo[k]) * rr[k], (hi[k] - oo[k]) * rr[k]));
        }
        tin0 = tin0 - fabsf(tin0) * 1e-6f;
        if (tin0 < 0.f) tin0 = 0.f;
    }
    /* The xy margin has to cover the noise of
     * the fp32 triangle test itself -- how far beside the exact ray a triangle can lie and still be reported hit --
     * which grows faster than linearly with the distance of the origin (measured against float64 geometry and the
     * brute force over ALL cells: 0.001 cell from 3 units away, 0.02 .. 1.7 cells from 50 units away on needle terrain
     * at N = 285 .. 4096).  Round 3 (a fuzz mismatch of the linear term, tests/test_oracle_band.py::
     * test_far_origin_needle_regression): beyond a reach of 8 units the distance term grows with the square of
     * reach / 8 (the walk only gets slower); within 8 units -- every BASELINE configuration -- it is what it was.
     * The constant part is slack on top (1/64 cell until round 3, 1/256 since: the distance term alone is never
     * below 8 eps x the grid's side, the rounding of the slab arithmetic). */
    const float reach = fabsf(oo[0]) + fabsf(oo[1]) + tin0 * (fabsf(od[0]) + fabsf(od[1])) + 2.f;
    const float far = fmaxf(1.f, 0.125f * reach);
    /* capped at 8 cells: the strip the walk visits stays bounded however far the origin (a ray from 200 units away
     * at N = 4096 would otherwise ask for 120 cells each side) */
    const float m = 0.00390625f + fminf(8.f, 4.8e-7f * reach * fmaxf(hx, hy) * (far * far));
    float tin = 0.f, tout = maxt;
    {
        const float ex = 1e-4f + m * f->sx, ey = 1e-4f + m * f->sy, ez = fmaf(2.f * m, zspan, mz0); /* (sx = 1 / hx: a cell; 2 m: as the node tests) */
        const float lo[3] = { -1.f - ex, -1.f - ey, f->zmin - ez };
        const float hi[3] = {  1.f + ex,  1.f + ey, f->zmax + ez };
        for (int k = 0; k < 3; ++k) {
            if (od[k] == 0.f) {
                if (oo[k] < lo[k] || oo[k] > hi[k]) return;
            } else {
                float t1 = (lo[k] - oo[k]) * rr[k], t2 = (hi[k] - oo[k]) * rr[k];
                tin = fmaxf(tin, fminf(t1, t2));
                tout = fminf(tout, fmaxf(t1, t2));
            }
        }
        tin = tin - fabsf(tin) * 1e-6f;
        if (tin < 0.f) tin = 0.f;
        tout = tout + fabsf(tout) * 1e-6f;
        if (!(tin <= tout)) return;
    }

    /* traversal ray: grid units (cell = 1), re-based at t = tin, mirrored into order space */
    const int fx = od[0] < 0.f, fy = od[1] < 0.f;
    const float Wp = (float) (1 << top);
    float gx = (fmaf(tin, od[0], oo[0]) + 1.f) * hx, gy = (fmaf(tin, od[1], oo[1]) + 1.f) * hy;
    float gz = fmaf(tin, od[2], oo[2]);
    float dx = od[0] * hx, dy = od[1] * hy, dz = od[2];
    if (fx) { gx = Wp - gx; dx = -dx; }
    if (fy) { gy = Wp - gy; dy = -dy; }
    /* |.|: a negative-zero component is not mirrored above and would give -inf; +inf for axis-parallel rays */
    const float idx = 1.0f / fabsf(dx), idy = 1.0f / fabsf(dy);
    const float mz = mz0 + 4.8e-7f * (fabsf(oo[2]) + tin * fabsf(od[2]) + zspan);
    float thi = (tout - tin);
    thi = thi + thi * 1e-6f + 1e-30f;

    int X = 0, Y = 0, L = top;
    for (;;) {
        ++g_stat_nodes;
        const float S = (float) (1 << L);
        const float bx0 = (float) X * S - m, bx1 = (float) (X + 1) * S + m;
        const float by0 = (float) Y * S - m, by1 = (float) (Y + 1) * S + m;
        /* direction components are >= 0 in order space: no min/max swap needed;
         * fmaxf/fminf drop the NaN of 0*inf (origin exactly on a slab plane) */
        float t0 = fmaxf(fmaxf((bx0 - gx) * idx, (by0 - gy) * idy), 0.f);
        float t1 = fminf(fminf((bx1 - gx) * idx, (by1 - gy) * idy), thi);
        int overlap = 0;
        const int ix = fx ? ((1 << (top - L)) - 1 - X) : X;
        const int iy = fy ? ((1 << (top - L)) - 1 - Y) : Y;
        if (t0 <= t1) {
            float zlo, zhi;
            int inside;
            if (L == 0) {
                inside = ix < cw && iy < ch;
                if (inside) {
                    const float *r0 = &f->h[(size_t) iy * f->W + ix], *r1 = r0 + f->W;
                    float a = r0[0] * f->s, bb = r0[1] * f->s, c = r1[0] * f->s, e = r1[1] * f->s;
                    zlo = fminf(fminf(a, bb), fminf(c, e));
                    zhi = fmaxf(fmaxf(a, bb), fmaxf(c, e));
                }
            } else {
                inside = ix < f->mw[L] && iy < f->mh[L];
                if (inside) {
                    const float *c = &f->mip[L][2 * ((size_t) iy * f->mw[L] + ix)];
                    zlo = c[0]; zhi = c[1];
                }
            }
            if (inside) {
                float z0 = fmaf(t0, dz, gz), z1 = fmaf(t1, dz, gz);
                /* The triangle test may report a hit up to m cells beside the walk's ray.  On a needle triangle that
                 * is up to  m x (|dz/dx| + |dz/dy|)  above or below it, and each partial derivative of a triangle is
                 * bounded by the height range of its cell (one cell wide), i.e. by the range of any node that
                 * contains the cell: 2 m x (height range of the node).  (Until round 4 the factor was 1: 9 % short of
                 * a noise hit the full brute force reports at N = 4096 from 8 units away -- the HIP walk's records
                 * carry (|a| + |b| + r) m and found it; tests/test_oracle_band.py::test_walk_needle_term_regression.) */
                const float mzz = fmaf(2.f * (m + 1e-6f * (float) (1 << top)), zhi - zlo, mz); /* (+ the slack the HIP walk's sheared line carries) */
                float rlo = fminf(z0, z1) - mzz, rhi = fmaxf(z0, z1) + mzz;
                overlap = rlo <= zhi && rhi >= zlo;
            }
        }
        if (overlap) {
            if (L > 0) { X <<= 1; Y <<= 1; --L; continue; }
            ++g_stat_leaves;
            if (test_cell(f, ix, iy, oo, od, maxt, b)) {
                if (any_hit) return;
                float tb = b->t - tin;
                tb = tb + fabsf(tb) * 1e-6f + 1e-30f;
                if (tb < thi) thi = tb;
            }
        }
        /* advance to the next node in mirrored Morton order, climbing while the
         * current node is the last (k = 3) child of its parent */
        while (L < top && (X & 1) && (Y & 1)) { X >>= 1; Y >>= 1; ++L; }
        if (L >= top) break;
        if ((X & 1) == 0) X |= 1;          /* k = 0 -> 1,  k = 2 -> 3 */
        else { X &= ~1; Y |= 1; }          /* k = 1 -> 2 */
    }
}


/*
 * BAND brute force: an intersector that shares NOTHING with the hierarchical walk -- no mips, no xy margin m, no
 * z margin mz, none of their constants.  In float64: the object-space ray in cell units, its parameter range
 * clipped to the grid's xy extent grown by HFO_BAND cells and to the global height range grown by 1 % of its span
 * (+1e-3; the walk's own clip uses 1e-5), both ends then pushed out by HFO_BAND cells along xy; every cell whose
 * xy box lies within +/-HFO_BAND cells of that segment is tested with the same fp32 test_cell and tie rule as the
 * brute force over all cells (kdtree.h:2424-2448).  A cell farther than two cells from the ray's line cannot pass the
 * fp32 Moeller-Trumbore test at any grid size this library supports, so the result is the full brute force's --
 * checked against it directly on the small grids where that one is affordable (tests/test_oracle_band.py) -- and
 * it stays affordable at the BASELINE grid sizes (N = 1024 ... 4096), where it pins the hierarchical walks of the
 * oracle and of the HIP kernels (accelerated == naive on the real scene: src/render/tests/test_kdtrees.py:52-82).
 */
#define HFO_BAND 2
static void trace_band(const hfo_field *f, const float o[3], const float d[3], float maxt,
                       int any_hit, best_t *b) {
    float oo[3], od[3];
    xform_point(f->to_object, o, oo);
    xform_vec(f->to_object, d, od);
    b->hit = 0; b->t = INFINITY; b->u = b->v = 0.f; b->prim = 0;
    if (!finite3(oo) || !finite3(od) || !(maxt >= 0.f)) return;
    if (od[0] == 0.f && od[1] == 0.f && od[2] == 0.f) return;   /* null direction: every test is NaN */
    const int cw = f->W - 1, ch = f->H - 1;
    const double B = (double) HFO_BAND;
    const double g[3] = { ((double) oo[0] + 1.0) * 0.5 * cw, ((double) oo[1] + 1.0) * 0.5 * ch, (double) oo[2] };
    const double e[3] = { (double) od[0] * 0.5 * cw, (double) od[1] * 0.5 * ch, (double) od[2] };
    const double span = (double) f->zmax - (double) f->zmin;
    const double zpad = 1e-2 * span + 1e-3 * (fabs((double) f->zmin) + fabs((double) f->zmax)) + 1e-30;
    const double lo[3] = { -B, -B, (double) f->zmin - zpad };
    const double hi[3] = { cw + B, ch + B, (double) f->zmax + zpad };
    double t0 = 0.0, t1 = (double) maxt;
    for (int k = 0; k < 3; ++k) {
        if (e[k] == 0.0) {
            if (g[k] < lo[k] || g[k] > hi[k]) return;
        } else {
            const double a = (lo[k] - g[k]) / e[k], c = (hi[k] - g[k]) / e[k];
            t0 = fmax(t0, fmin(a, c));
            t1 = fmin(t1, fmax(a, c));
        }
    }
    if (!(t0 <= t1)) return;
    const double sxy = hypot(e[0], e[1]);
    if (sxy > 0.0) {                       /* both ends out by B cells along xy (t0 may become negative: the */
        const double ext = B / sxy;        /* triangle test's own t >= 0 decides) */
        if (isfinite(ext)) { t0 -= ext; t1 += ext; }
    }
    if (!isfinite(t1) || !isfinite(t0)) return;  /* (cannot happen: a non-null direction bounds t1 through a slab above) */
    const double xa = g[0] + t0 * e[0], xb = g[0] + t1 * e[0];
    const double ya = g[1] + t0 * e[1], yb = g[1] + t1 * e[1];
    const int xmajor = fabs(xb - xa) >= fabs(yb - ya);
    /* u = major axis, w = minor axis */
    const double ua = xmajor ? xa : ya, ub = xmajor ? xb : yb;
    const double gu = xmajor ? g[0] : g[1], eu = xmajor ? e[0] : e[1];
    const double gw = xmajor ? g[1] : g[0], ew = xmajor ? e[1] : e[0];
    const int nu = xmajor ? cw : ch, nw = xmajor ? ch : cw;
    int cu0 = (int) floor(fmin(ua, ub)) - HFO_BAND, cu1 = (int) floor(fmax(ua, ub)) + HFO_BAND;
    if (cu0 < 0) cu0 = 0;
    if (cu1 > nu - 1) cu1 = nu - 1;
    for (int cu = cu0; cu <= cu1; ++cu) {
        double wa, wb;
        if (eu == 0.0) { wa = gw + t0 * ew; wb = gw + t1 * ew; }
        else {
            double ta = ((double) cu - B - gu) / eu, tb = ((double) cu + 1.0 + B - gu) / eu;
            if (ta > tb) { const double q = ta; ta = tb; tb = q; }
            ta = fmax(ta, t0); tb = fmin(tb, t1);
            if (ta > tb) continue;
            wa = gw + ta * ew; wb = gw + tb * ew;
        }
        int cv0 = (int) floor(fmin(wa, wb)) - HFO_BAND, cv1 = (int) floor(fmax(wa, wb)) + HFO_BAND;
        if (cv0 < 0) cv0 = 0;
        if (cv1 > nw - 1) cv1 = nw - 1;
        for (int cv = cv0; cv <= cv1; ++cv) {
            const int cx = xmajor ? cu : cv, cy = xmajor ? cv : cu;
            if (test_cell(f, cx, cy, oo, od, maxt, b) && any_hit) return;
        }
    }
}

void hfo_intersect_naive(const hfo_field *f, const float o[3], const float d[3], float maxt,
                         float *t, float uv[2], uint32_t *prim) {
    best_t b; trace_naive(f, o, d, maxt, 0, &b); write_result(&b, t, uv, prim);
}
void hfo_intersect(const hfo_field *f, const float o[3], const float d[3], float maxt,
                   float *t, float uv[2], uint32_t *prim) {
    best_t b; trace_hier(f, o, d, maxt, 0, &b); write_result(&b, t, uv, prim);
}
void hfo_trace_stats(const hfo_field *f, const float o[3], const float d[3], float maxt,
                     uint32_t *nodes, uint32_t *leaves) {
    best_t b; trace_hier(f, o, d, maxt, 0, &b); *nodes = g_stat_nodes; *leaves = g_stat_leaves;
}

/* ray_test == ray_intersect_preliminary(...).is_valid(), src/render/shape.cpp:430-434 */
int hfo_ray_test_naive(const hfo_field *f, const float o[3], const float d[3], float maxt) {
    best_t b; trace_naive(f, o, d, maxt, 1, &b); return b.hit;
}
void hfo_intersect_band(const hfo_field *f, const float o[3], const float d[3], float maxt,
                        float *t, float uv[2], uint32_t *prim) {
    best_t b; trace_band(f, o, d, maxt, 0, &b); write_result(&b, t, uv, prim);
}
int hfo_ray_test_band(const hfo_field *f, const float o[3], const float d[3], float maxt) {
    best_t b; trace_band(f, o, d, maxt, 1, &b); return b.hit;
}
int hfo_ray_test(const hfo_field *f, const float o[3], const float d[3], float maxt) {
    best_t b; trace_hier(f, o, d, maxt, 1, &b); return b.hit;
}

/* ------------------------------------------------------------------------ */
/* Surface interaction (src/render/mesh.cpp:672-903 + interaction.h:476-499)  */
/* ------------------------------------------------------------------------ */
static inline float signf_(float x) { return x >= 0.f ? 1.f : -1.f; }              /* dr::sign */
static inline float mulsign(float a, float b) { return b >= 0.f ? a : -a; }
static inline float mulsign_neg(float a, float b) { return b >= 0.f ? -a : a; }

/* coordinate_system(), include/mitsuba/core/vector.h:116-136 */
static void coordinate_system(const float n[3], float s[3], float t[3]) {
    float sign = signf_(n[2]);
    float a = -rcpf(sign + n[2]);
    float b = n[0] * n[1] * a;
    s[0] = mulsign(n[0] * n[0] * a, n[2]) + 1.f;
    s[1] = mulsign(b, n[2]);
    s[2] = mulsign_neg(n[0], n[2]);
    t[0] = b;
    t[1] = fmaf(n[1], n[1] * a, sign);
    t[2] = -n[1];
}

/* world-space vertices + texcoords of a primitive */
static void prim_world(const hfo_field *f, uint32_t prim, float P[3][3], float UV[3][2],
                       int vi[3], int vj[3]) {
    prim_vertex_ids(f, prim, vi, vj);
    for (int k = 0; k < 3; ++k) {
        float q[3];
        hfo_vertex(f, vi[k], vj[k], q);
        xform_point(f->to_world, q, P[k]);
        UV[k][0] = (float) vj[k] * f->iu;
        UV[k][1] = (float) vi[k] * f->iv;
    }
}

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

/* Which of the hit triangle's three edges (k = 0: P0-P1, 1: P1-P2, 2: P2-P0) are SILHOUETTE edges for a ray of
 * object-space direction od: the neighbour across the edge does not exist (border of the grid), or faces the
 * ray the other way.  A triangle of the grid with slopes (zx, zy) faces the ray by sign(od.z - zx od.x - zy od.y)
 * (its upward normal is (-zx, -zy, 1)).  Interior edges between two triangles that face the ray the same way are
 * no visibility boundary, whatever the per-triangle SDF of mesh.cpp:863-890 says. */
static inline int tri_faces(float zx, float zy, const float od[3]) { return fmaf(-zy, od[1], fmaf(-zx, od[0], od[2])) >= 0.f; }
static uint32_t silhouette_edges(const hfo_field *f, uint32_t prim, const float od[3]) {
    const uint32_t cw = (uint32_t) (f->W - 1);
    const int cy = (int) ((prim >> 1) / cw), cx = (int) ((prim >> 1) - (uint32_t) cy * cw);
    const float *h = f->h;
    const float s = f->s, isx = 1.0f / f->sx, isy = 1.0f / f->sy;
#define HZ(i, j) (h[(size_t) (i) * f->W + (j)] * s)
    const float z00 = HZ(cy, cx), z10 = HZ(cy, cx + 1), z01 = HZ(cy + 1, cx), z11 = HZ(cy + 1, cx + 1);
    const int f0 = tri_faces((z10 - z00) * isx, (z01 - z00) * isy, od);   /* tri 0 = (v00, v10, v01) */
    const int f1 = tri_faces((z11 - z01) * isx, (z11 - z10) * isy, od);   /* tri 1 = (v11, v01, v10) */
    uint32_t m = 0;
    if ((prim & 1u) == 0) {
        /* edge 0 = v00-v10 (bottom): tri 1 of cell (cx, cy-1) */
        if (cy == 0 || tri_faces((z10 - z00) * isx, (z10 - HZ(cy - 1, cx + 1)) * isy, od) != f0) m |= 1u;
        if (f1 != f0) m |= 2u;                                           /* edge 1 = the diagonal */
        /* edge 2 = v01-v00 (left): tri 1 of cell (cx-1, cy) */
        if (cx == 0 || tri_faces((z01 - HZ(cy + 1, cx - 1)) * isx, (z01 - z00) * isy, od) != f0) m |= 4u;
    } else {
        /* edge 0 = v11-v01 (top): tri 0 of cell (cx, cy+1) */
        if (cy + 2 > f->H - 1 || tri_faces((z11 - z01) * isx, (HZ(cy + 2, cx) - z01) * isy, od) != f1) m |= 1u;
        if (f0 != f1) m |= 2u;
        /* edge 2 = v10-v11 (right): tri 0 of cell (cx+1, cy) */
        if (cx + 2 > f->W - 1 || tri_faces((HZ(cy, cx + 2) - z10) * isx, (z11 - z10) * isy, od) != f1) m |= 4u;
    }
#undef HZ
    return m;
}

/* boundary test of the height field: the triangle SDF of mesh.cpp:845-890 (distance of the hit point to the
 * triangle's edges in an equilateral reference triangle, scaled so that the incentre maps to 1), restricted to
 * the silhouette edges `edges`; 1 when the triangle has none.  On the outer border this is the rectangle's
 * border distance (rectangle.cpp:318-319) in triangle units; at self-occlusion silhouettes it vanishes where the
 * facing flips, like the grazing term sqr(dot(n, -d)) of the smooth-normal branch (mesh.cpp:892-898). */
static float boundary_test_flat(const float p[3], const float p0[3], const float dp0[3], const float dp1[3], uint32_t edges) {
    if (edges == 0u) return 1.0f;
    float rel[3];
    sub3(p, p0, rel);
    float bb1 = dot3(dp0, rel), bb2 = dot3(dp1, rel);
    float a11 = dot3(dp0, dp0), a12 = dot3(dp0, dp1), a22 = dot3(dp1, dp1);
    float inv_det = rcpf(a11 * a22 - a12 * a12);
    float u = fmaf(a22, bb1, -(a12 * bb2)) * inv_det;
    float v = fmaf(-a12, bb1, a11 * bb2) * inv_det;
    float w = 1.f - u - v;
    const float tp0[2] = { 0.f, 0.f }, tp1[2] = { 1.f, 0.f }, tp2[2] = { 0.5f, 0.5f * sqrtf(3.f) };
    float q[2] = { tp0[0] * w + tp1[0] * u + tp2[0] * v, tp0[1] * w + tp1[1] * u + tp2[1] * v };
    const float e[3][2] = { { tp1[0] - tp0[0], tp1[1] - tp0[1] },
                            { tp2[0] - tp1[0], tp2[1] - tp1[1] },
                            { tp0[0] - tp2[0], tp0[1] - tp2[1] } };
    const float *tp[3] = { tp0, tp1, tp2 };
    float s = signf_(e[0][0] * e[2][1] - e[0][1] * e[2][0]);
    float dmin0 = INFINITY, dmin1 = INFINITY;
    for (int k = 0; k < 3; ++k) {
        if (!((edges >> k) & 1u)) continue;
        float vv[2] = { q[0] - tp[k][0], q[1] - tp[k][1] };
        float c = clamp01(dot2(vv, e[k]) / dot2(e[k], e[k]));
        float pq[2] = { vv[0] - e[k][0] * c, vv[1] - e[k][1] * c };
        dmin0 = fminf(dmin0, dot2(pq, pq));
        dmin1 = fminf(dmin1, s * (vv[0] * e[k][1] - vv[1] * e[k][0]));
    }
    (void) dmin1;
    float dist = sqrtf(dmin0);
    dist /= sqrtf(3.f) / 6.f;
    return dist;
}

int hfo_compute_si(const hfo_field *f, const float o[3], const float d[3], float t_in,
                   const float uv_in[2], uint32_t prim, uint32_t flags, int active, hfo_si *si) {
    if ((flags & HFO_RAY_DETACHSHAPE) && (flags & HFO_RAY_FOLLOWSHAPE)) return -1; /* mesh.cpp:709-711 */
    memset(si, 0, sizeof(*si));
    /* pi.compute_surface_interaction: active &= is_valid() (interaction.h:667) */
    active = active && (t_in != INFINITY);
    if (!active) {
        si->t = INFINITY;                       /* interaction.h:479 */
        si->wi[0] = -d[0]; si->wi[1] = -d[1]; si->wi[2] = -d[2]; /* interaction.h:493 */
        if (flags & HFO_RAY_BOUNDARYTEST) si->boundary_test = 1e8f; /* interaction.h:497-498 */
        return 0;
    }
    float P[3][3], UV[3][2];
    int vi[3], vj[3];
    prim_world(f, prim, P, UV, vi, vj);

    float t = t_in, b1 = uv_in[0], b2 = uv_in[1], b0 = 1.f - b1 - b2;
    float dp0[3], dp1[3];
    sub3(P[1], P[0], dp0);
    sub3(P[2], P[0], dp1);
    /* si.p = fmadd(p0, b0, fmadd(p1, b1, p2 * b2)), mesh.cpp:745 */
    for (int k = 0; k < 3; ++k)
        si->p[k] = fmaf(P[0][k], b0, fmaf(P[1][k], b1, P[2][k] * b2));
    /* FollowShape re-derives t from p (mesh.cpp:748-752): same primal value up to rounding */
    if (flags & HFO_RAY_FOLLOWSHAPE) {
        float po[3];
        sub3(si->p, o, po);
        t = sqrtf(dot3(po, po) / dot3(d, d));
    }
    si->t = t;
    float N[3];
    cross3(dp0, dp1, N);
    normalize3(N, si->n);                       /* mesh.cpp:757 */
    si->uv[0] = b1; si->uv[1] = b2;             /* mesh.cpp:760 */
    coordinate_system(si->n, si->dp_du, si->dp_dv);
    if (flags & (HFO_RAY_UV | HFO_RAY_DPDUV)) { /* mesh.cpp:764-789 (vertex texcoords present) */
        for (int k = 0; k < 2; ++k)
            si->uv[k] = fmaf(UV[2][k], b2, fmaf(UV[1][k], b1, UV[0][k] * b0));
        if (flags & HFO_RAY_DPDUV) {
            float duv0[2] = { UV[1][0] - UV[0][0], UV[1][1] - UV[0][1] };
            float duv1[2] = { UV[2][0] - UV[0][0], UV[2][1] - UV[0][1] };
            float det = fmaf(duv0[0], duv1[1], -(duv0[1] * duv1[0]));
            float inv_det = rcpf(det);
            if (det != 0.f)
                for (int k = 0; k < 3; ++k) {
                    si->dp_du[k] = fmaf(duv1[1], dp0[k], -(duv0[1] * dp1[k])) * inv_det;
                    si->dp_dv[k] = fmaf(-duv1[0], dp0[k], duv0[0] * dp1[k]) * inv_det;
                }
        }
    }
    for (int k = 0; k < 3; ++k) si->sh_n[k] = si->n[k]; /* flat shading, mesh.cpp:834 */
    if (f->flip_normals)                                  /* mesh.cpp:837-840 */
        for (int k = 0; k < 3; ++k) { si->n[k] = -si->n[k]; si->sh_n[k] = -si->sh_n[k]; }
    if (flags & HFO_RAY_BOUNDARYTEST) {
        float od[3];
        xform_vec(f->to_object, d, od);
        /* 0x10000: the reference Mesh's per-triangle SDF over all three edges (mesh.cpp:845-890) */
        si->boundary_test = boundary_test_flat(si->p, P[0], dp0, dp1, (flags & 0x10000u) ? 7u : silhouette_edges(f, prim, od));
    }

    /* finalize_surface_interaction, interaction.h:476-499 */
    if (flags & HFO_RAY_SHADINGFRAME) {        /* initialize_sh_frame, interaction.h:257-267 */
        float nd = -dot3(si->sh_n, si->dp_du), tmp[3];
        for (int k = 0; k < 3; ++k) tmp[k] = fmaf(si->sh_n[k], nd, si->dp_du[k]);
        normalize3(tmp, si->sh_s);
        if (si->dp_du[0] == 0.f && si->dp_du[1] == 0.f && si->dp_du[2] == 0.f) {
            float dummy[3];
            coordinate_system(si->sh_n, si->sh_s, dummy);
        }
        cross3(si->sh_n, si->sh_s, si->sh_t);
    }
    float md[3] = { -d[0], -d[1], -d[2] };
    si->wi[0] = dot3(md, si->sh_s); si->wi[1] = dot3(md, si->sh_t); si->wi[2] = dot3(md, si->sh_n);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Adjoint: reverse mode of hfo_compute_si.  The reference has no hand-written */
/* counterpart (Dr.Jit AD produces it, prb_reparam.py:586-587); semantics of   */
/* the three modes: mesh.cpp:695-752, known answers: test_mesh.py:562-638.     */
/* ------------------------------------------------------------------------ */
static inline void axpy3(float a, const float x[3], float y[3]) {
    y[0] += a * x[0]; y[1] += a * x[1]; y[2] += a * x[2];
}

int hfo_adjoint(const hfo_field *f, const float o[3], const float d[3], float t_in,
                const float uv_in[2], uint32_t prim, uint32_t flags, int active,
                const hfo_si_grad *g, float *grad_h, float *grad_o, float *grad_d) {
    if ((flags & HFO_RAY_DETACHSHAPE) && (flags & HFO_RAY_FOLLOWSHAPE)) return -1;
    active = active && (t_in != INFINITY);
    if (!active) return 0;
    const int follow = (flags & HFO_RAY_FOLLOWSHAPE) != 0, detach = (flags & HFO_RAY_DETACHSHAPE) != 0;

    float P[3][3], UV[3][2];
    int vi[3], vj[3];
    prim_world(f, prim, P, UV, vi, vj);
    const float b1 = uv_in[0], b2 = uv_in[1], b0 = 1.f - b1 - b2;
    const float bw[3] = { b0, b1, b2 };
    float dp0[3], dp1[3], p[3];
    sub3(P[1], P[0], dp0);
    sub3(P[2], P[0], dp1);
    for (int k = 0; k < 3; ++k) p[k] = fmaf(P[0][k], b0, fmaf(P[1][k], b1, P[2][k] * b2));

    float gP[3][3] = { { 0 } }, gdp0[3] = { 0, 0, 0 }, gdp1[3] = { 0, 0, 0 };
    float go[3] = { 0, 0, 0 }, gd[3] = { 0, 0, 0 };
    float gb[3] = { 0, 0, 0 };
    float gp[3] = { g->p[0], g->p[1], g->p[2] };
    float gt = g->t;

    /* dp_du, dp_dv (only when they come from the texcoords; constant duv) */
    if ((flags & HFO_RAY_DPDUV)) {
        float duv0[2] = { UV[1][0] - UV[0][0], UV[1][1] - UV[0][1] };
        float duv1[2] = { UV[2][0] - UV[0][0], UV[2][1] - UV[0][1] };
        float det = fmaf(duv0[0], duv1[1], -(duv0[1] * duv1[0]));
        float inv_det = rcpf(det);
        if (det != 0.f) {
            axpy3(duv1[1] * inv_det, g->dp_du, gdp0);
            axpy3(-duv0[1] * inv_det, g->dp_du, gdp1);
            axpy3(-duv1[0] * inv_det, g->dp_dv, gdp0);
            axpy3(duv0[0] * inv_det, g->dp_dv, gdp1);
        }
    }
    /* n = sh_n = +-normalize(cross(dp0, dp1)) */
    {
        float N[3], n[3], gn[3];
        cross3(dp0, dp1, N);
        float r = rsqrtf_(dot3(N, N));
        for (int k = 0; k < 3; ++k) n[k] = N[k] * r;
        float sgn = f->flip_normals ? -1.f : 1.f;
        for (int k = 0; k < 3; ++k) gn[k] = sgn * (g->n[k] + g->sh_n[k]);
        float proj = dot3(n, gn), gN[3];
        for (int k = 0; k < 3; ++k) gN[k] = (gn[k] - n[k] * proj) * r;
        float c0[3], c1[3];
        cross3(dp1, gN, c0);   /* d/d(dp0) of <gN, dp0 x dp1> */
        cross3(gN, dp0, c1);   /* d/d(dp1) */
        axpy3(1.f, c0, gdp0);
        axpy3(1.f, c1, gdp1);
    }
    /* FollowShape: t = sqrt(|p-o|^2 / |d|^2) carries gradient into p, o, d (mesh.cpp:751-752) */
    if (follow) {
        float po[3];
        sub3(p, o, po);
        float dd = dot3(d, d), t = sqrtf(dot3(po, po) / dd);
        float c = gt / (t * dd);
        axpy3(c, po, gp);
        axpy3(-c, po, go);
        axpy3(-gt * t / dd, d, gd);
    }
    /* p = sum b_k P_k ; uv = sum b_k uv_k */
    for (int k = 0; k < 3; ++k) {
        gb[k] += dot3(gp, P[k]);
        if (flags & (HFO_RAY_UV | HFO_RAY_DPDUV)) gb[k] += g->uv[0] * UV[k][0] + g->uv[1] * UV[k][1];
        axpy3(bw[k], gp, gP[k]);
    }
    float gu = gb[1] - gb[0], gv = gb[2] - gb[0];
    if (!(flags & (HFO_RAY_UV | HFO_RAY_DPDUV))) { gu += g->uv[0]; gv += g->uv[1]; } /* si.uv = (b1,b2) */

    if (!follow) {
        /* reverse of moeller_trumbore(ray, P0, P1, P2): t_d, prim_uv_d carry the gradient
         * (replace_grad, mesh.cpp:728-735) */
        float e1[3], e2[3], pvec[3], tvec[3], qvec[3];
        sub3(P[1], P[0], e1);
        sub3(P[2], P[0], e2);
        cross3(d, e2, pvec);
        float det = dot3(e1, pvec), inv = rcpf(det);
        sub3(o, P[0], tvec);
        cross3(tvec, e1, qvec);
        float a_u = dot3(tvec, pvec), a_v = dot3(d, qvec), a_t = dot3(e2, qvec);
        float g_au = gu * inv, g_av = gv * inv, g_at = gt * inv;
        float g_inv = gu * a_u + gv * a_v + gt * a_t;
        float g_det = -g_inv * inv * inv;
        float ge1[3] = { 0, 0, 0 }, ge2[3] = { 0, 0, 0 }, gq[3] = { 0, 0, 0 }, gtv[3] = { 0, 0, 0 }, gpv[3] = { 0, 0, 0 };
        axpy3(g_at, qvec, ge2); axpy3(g_at, e2, gq);
        axpy3(g_av, qvec, gd);  axpy3(g_av, d, gq);
        float c[3];
        cross3(e1, gq, c);   axpy3(1.f, c, gtv);   /* qvec = tvec x e1 */
        cross3(gq, tvec, c); axpy3(1.f, c, ge1);
        axpy3(g_au, pvec, gtv); axpy3(g_au, tvec, gpv);
        axpy3(g_det, pvec, ge1); axpy3(g_det, e1, gpv);
        cross3(e2, gpv, c);  axpy3(1.f, c, gd);    /* pvec = d x e2 */
        cross3(gpv, d, c);   axpy3(1.f, c, ge2);
        axpy3(1.f, gtv, go); axpy3(-1.f, gtv, gP[0]);
        axpy3(1.f, ge1, gP[1]); axpy3(-1.f, ge1, gP[0]);
        axpy3(1.f, ge2, gP[2]); axpy3(-1.f, ge2, gP[0]);
    }
    /* dp0 = P1 - P0, dp1 = P2 - P0 */
    axpy3(1.f, gdp0, gP[1]); axpy3(-1.f, gdp0, gP[0]);
    axpy3(1.f, gdp1, gP[2]); axpy3(-1.f, gdp1, gP[0]);

    if (!detach && grad_h) {
        /* P_k = to_world * (x, y, s*h): dP_k/dh = s * (third column of to_world) */
        const float ez[3] = { f->to_world[2], f->to_world[6], f->to_world[10] };
        for (int k = 0; k < 3; ++k) {
            float gh = f->s * dot3(ez, gP[k]);
#ifdef _OPENMP
#pragma omp atomic
#endif
            grad_h[(size_t) vi[k] * f->W + vj[k]] += gh;
        }
    }
    if (grad_o) for (int k = 0; k < 3; ++k) grad_o[k] += go[k];
    if (grad_d) for (int k = 0; k < 3; ++k) grad_d[k] += gd[k];
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Batched SoA wrappers (OpenMP over rays; scalar per ray like               */
/* kdtree_trace_func_wrapper, src/render/scene_native.inl:130-172)           */
/* ------------------------------------------------------------------------ */
static void set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void) nthreads;
#endif
}

void hfo_intersect_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                         const uint8_t *active, int mode, int nthreads,
                         float *t, float *u, float *v, uint32_t *prim) {
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        best_t b;
        b.hit = 0;
        if (!active || active[i]) {
            float o[3] = { rays[0][i], rays[1][i], rays[2][i] }, d[3] = { rays[3][i], rays[4][i], rays[5][i] };
            if (mode == 1)      trace_naive(f, o, d, rays[6][i], 0, &b);
            else if (mode == 2) trace_band(f, o, d, rays[6][i], 0, &b);
            else                trace_hier(f, o, d, rays[6][i], 0, &b);
        }
        float uv[2];
        write_result(&b, &t[i], uv, &prim[i]);
        u[i] = uv[0]; v[i] = uv[1];
    }
}

void hfo_ray_test_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                        const uint8_t *active, int mode, int nthreads, uint8_t *hit) {
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        best_t b;
        b.hit = 0;
        if (!active || active[i]) {
            float o[3] = { rays[0][i], rays[1][i], rays[2][i] }, d[3] = { rays[3][i], rays[4][i], rays[5][i] };
            if (mode == 1)      trace_naive(f, o, d, rays[6][i], 1, &b);
            else if (mode == 2) trace_band(f, o, d, rays[6][i], 1, &b);
            else                trace_hier(f, o, d, rays[6][i], 1, &b);
        }
        hit[i] = (uint8_t) b.hit;
    }
}

int hfo_compute_si_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                         const float *t, const float *u, const float *v, const uint32_t *prim,
                         const uint8_t *active, uint32_t flags, int nthreads, float *const out[28]) {
    if ((flags & HFO_RAY_DETACHSHAPE) && (flags & HFO_RAY_FOLLOWSHAPE)) return -1;
    set_threads(nthreads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float o[3] = { rays[0][i], rays[1][i], rays[2][i] }, d[3] = { rays[3][i], rays[4][i], rays[5][i] };
        float uv[2] = { u[i], v[i] };
        hfo_si si;
        hfo_compute_si(f, o, d, t[i], uv, prim[i], flags, !active || active[i], &si);
        const float *src = (const float *) &si;
        for (int k = 0; k < 28; ++k)
            if (out[k]) out[k][i] = src[k];
    }
    return 0;
}

int hfo_adjoint_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                      const float *t, const float *u, const float *v, const uint32_t *prim,
                      const uint8_t *active, uint32_t flags, int nthreads,
                      const float *const gin[18], float *grad_h,
                      float *const grad_o[3], float *const grad_d[3]) {
    if ((flags & HFO_RAY_DETACHSHAPE) && (flags & HFO_RAY_FOLLOWSHAPE)) return -1;
    set_threads(nthreads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float o[3] = { rays[0][i], rays[1][i], rays[2][i] }, d[3] = { rays[3][i], rays[4][i], rays[5][i] };
        float uv[2] = { u[i], v[i] };
        hfo_si_grad g;
        float *dst = (float *) &g;
        for (int k = 0; k < 18; ++k) dst[k] = gin[k] ? gin[k][i] : 0.f;
        float go[3] = { 0, 0, 0 }, gd[3] = { 0, 0, 0 };
        hfo_adjoint(f, o, d, t[i], uv, prim[i], flags, !active || active[i], &g, grad_h,
                    grad_o ? go : NULL, grad_d ? gd : NULL);
        if (grad_o) for (int k = 0; k < 3; ++k) grad_o[k][i] = go[k];
        if (grad_d) for (int k = 0; k < 3; ++k) grad_d[k][i] = gd[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Synthetic workload                                  */
/* ------------------------------------------------------------------------ */
void hfo_make_sine_heights(int W, int H, float fx, float fy, float *out) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            double u = (double) j / (double) (W - 1), v = (double) i / (double) (H - 1);
            out[(size_t) i * W + j] = (float) (0.5 + 0.25 * sin(two_pi * fx * u) * cos(two_pi * fy * v)
                                               + 0.125 * sin(two_pi * 7.0 * (u + v)));
        }
}

/* sample_tea_32, include/mitsuba/core/random.h:76-91 */
void hfo_sample_tea_32(uint32_t v0, uint32_t v1, int rounds, uint32_t out[2]) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    out[0] = v0; out[1] = v1;
}
