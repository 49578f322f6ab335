/*
 * hf_oracle.h -- CPU ORACLE for the differentiable heightfield hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: the reference snapshot contains no heightfield source
 * (SURVEY.md section 0), so there is no reference heightfield output to be
 * bit-exact against ("parity unpinned" for the heightfield itself).  What
 * this oracle restates, and what IS pinned by the reference's own code and
 * tests, is the reference's *triangle-mesh* semantics applied to the
 * two-triangles-per-cell tessellation of the height grid:
 *
 *   - affine point/vector transform, FMA chain
 *       include/mitsuba/core/transform.h:104-111, 130-138, 168-171
 *   - Moeller-Trumbore, exact operation order
 *       include/mitsuba/render/mesh.h:357-380
 *   - closest-hit bookkeeping of the brute force loop (ties on t go to the
 *     LAST primitive tested because the test is `t <= ray.maxt`)
 *       include/mitsuba/render/kdtree.h:2424-2448, 2270-2277
 *   - surface interaction, FollowShape / DetachShape / default modes,
 *     flat-shaded boundary test
 *       src/render/mesh.cpp:672-903
 *   - finalize (shading frame, wi)
 *       include/mitsuba/render/interaction.h:257-267, 476-499
 *   - coordinate_system()      include/mitsuba/core/vector.h:116-136
 *   - bbox slab test           include/mitsuba/core/bbox.h:302-327
 *
 * Dr.Jit 0.4.2 (pyproject.toml:2) supplies dot/cross/normalize/rcp/rsqrt but
 * its source is absent (ext/drjit is an empty submodule).  Restated here as:
 *   dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   cross(a,b) = fmsub(a.yzx, b.zxy, a.zxy*b.yzx)
 *   rcp(x)     = 1.0f/x (IEEE), rsqrt(x) = 1.0f/sqrtf(x), normalize(v) = v*rsqrt(dot(v,v))
 * Compile with -ffp-contract=off so that only the fmaf() calls written below
 * fuse; the HIP kernels use the same explicit operation order, which is what
 * makes `prim_index` bit-exact between the two.
 */
#ifndef HF_ORACLE_H
#define HF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RayFlags, include/mitsuba/render/interaction.h:19-69 */
enum {
    HFO_RAY_MINIMAL       = 0x1,
    HFO_RAY_UV            = 0x2,
    HFO_RAY_DPDUV         = 0x4,
    HFO_RAY_SHADINGFRAME  = 0x8,
    HFO_RAY_BOUNDARYTEST  = 0x40,
    HFO_RAY_FOLLOWSHAPE   = 0x80,
    HFO_RAY_DETACHSHAPE   = 0x100,
    HFO_RAY_ALL           = 0x2 | 0x4 | 0x8
};

typedef struct hfo_field hfo_field;

/* One surface interaction record (array-of-struct; the oracle is scalar). */
typedef struct hfo_si {
    float t;
    float p[3];
    float n[3];
    float uv[2];
    float sh_n[3];
    float dp_du[3];
    float dp_dv[3];
    float boundary_test;
    /* finalize_surface_interaction() outputs */
    float sh_s[3];
    float sh_t[3];
    float wi[3];
} hfo_si;

/* Upstream gradient of a scalar loss w.r.t. the differentiable SI fields. */
typedef struct hfo_si_grad {
    float t;
    float p[3];
    float n[3];
    float uv[2];
    float sh_n[3];
    float dp_du[3];
    float dp_dv[3];
} hfo_si_grad;

/* heights: H rows x W columns, row-major, row 0 at object y = -1.
 * to_world / to_object: row-major 3x4 affine matrices. */
hfo_field *hfo_create(int W, int H, const float *heights, float max_height,
                      const float *to_world, const float *to_object,
                      int flip_normals);
void hfo_destroy(hfo_field *f);
/* replace heights (rebuilds min/max mips) */
void hfo_set_heights(hfo_field *f, const float *heights);
int  hfo_num_levels(const hfo_field *f);
/* copy mip level l (1..top) as interleaved (min,max) pairs; returns w*h */
int  hfo_get_mip(const hfo_field *f, int level, float *out, int *w, int *h);
/* world-space bounding box {minx,miny,minz,maxx,maxy,maxz} */
void hfo_bbox(const hfo_field *f, float out[6]);

/* object-space vertex position of grid vertex (row i, column j) */
void hfo_vertex(const hfo_field *f, int i, int j, float out[3]);

/* --- preliminary intersection (scalar) ---------------------------------- */
/* brute force over all 2(W-1)(H-1) triangles in prim_index order */
void hfo_intersect_naive(const hfo_field *f, const float o[3], const float d[3],
                         float maxt, float *t, float uv[2], uint32_t *prim);
/* hierarchical min/max-mip traversal; same result as the brute force */
void hfo_intersect(const hfo_field *f, const float o[3], const float d[3],
                   float maxt, float *t, float uv[2], uint32_t *prim);
/* number of quadtree nodes visited / cells tested by hfo_intersect for one ray (tuning aid) */
void hfo_trace_stats(const hfo_field *f, const float o[3], const float d[3], float maxt,
                     uint32_t *nodes, uint32_t *leaves);
/* brute force over the cells within +/-2 cells of the ray's xy segment (float64 geometry, no mips, no margins shared
 * with the hierarchical walk): the independent check at grid sizes where the full brute force is unaffordable */
void hfo_intersect_band(const hfo_field *f, const float o[3], const float d[3],
                        float maxt, float *t, float uv[2], uint32_t *prim);
int  hfo_ray_test_band(const hfo_field *f, const float o[3], const float d[3], float maxt);
int  hfo_ray_test_naive(const hfo_field *f, const float o[3], const float d[3], float maxt);
int  hfo_ray_test(const hfo_field *f, const float o[3], const float d[3], float maxt);

/* --- surface interaction + adjoint (scalar) ------------------------------ */
/* returns 0, or -1 for DetachShape|FollowShape (mesh.cpp:709-711) */
int hfo_compute_si(const hfo_field *f, const float o[3], const float d[3],
                   float t, const float uv[2], uint32_t prim,
                   uint32_t ray_flags, int active, hfo_si *si);
/* reverse mode of hfo_compute_si w.r.t. heights (accumulated into grad_h,
 * H*W floats) and, if non-NULL, the ray (grad_o, grad_d: 3 floats, accumulated) */
int hfo_adjoint(const hfo_field *f, const float o[3], const float d[3],
                float t, const float uv[2], uint32_t prim,
                uint32_t ray_flags, int active, const hfo_si_grad *g,
                float *grad_h, float *grad_o, float *grad_d);

/* --- batched SoA entry points (OpenMP over rays) -------------------------- */
/* rays: 7 arrays of n floats (ox,oy,oz,dx,dy,dz,maxt). mode: 0 = hierarchical,
 * 1 = brute force, 2 = band brute force.  active may be NULL. */
void hfo_intersect_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                         const uint8_t *active, int mode, int nthreads,
                         float *t, float *u, float *v, uint32_t *prim);
void hfo_ray_test_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                        const uint8_t *active, int mode, int nthreads, uint8_t *hit);
/* si_out: 28 arrays of n floats in the order of hfo_si's members
 * (t, p[3], n[3], uv[2], sh_n[3], dp_du[3], dp_dv[3], boundary_test,
 *  sh_s[3], sh_t[3], wi[3]); entries may be NULL. */
int hfo_compute_si_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                         const float *t, const float *u, const float *v,
                         const uint32_t *prim, const uint8_t *active,
                         uint32_t ray_flags, int nthreads, float *const si_out[28]);
/* grad_in: 18 arrays of n floats in the order of hfo_si_grad's members;
 * NULL entries are taken as zero.  grad_o/grad_d: 3 arrays each or NULL. */
int hfo_adjoint_batch(const hfo_field *f, int64_t n, const float *const rays[7],
                      const float *t, const float *u, const float *v,
                      const uint32_t *prim, const uint8_t *active,
                      uint32_t ray_flags, int nthreads,
                      const float *const grad_in[18], float *grad_h,
                      float *const grad_o[3], float *const grad_d[3]);

/* Synthetic workload generators of SURVEY.md section 8(d) --------------------- */
/* h[i,j] = 0.5 + 0.25 sin(2 pi fx u) cos(2 pi fy v) + 0.125 sin(2 pi 7 (u+v)) */
void hfo_make_sine_heights(int W, int H, float fx, float fy, float *out);
/* sample_tea_32, include/mitsuba/core/random.h:76-91 */
void hfo_sample_tea_32(uint32_t v0, uint32_t v1, int rounds, uint32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
