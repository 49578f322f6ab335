// hf_device.h -- device-side arithmetic shared by the gfx950 kernels.
//
// "Spec arithmetic": the per-triangle test, the transforms and the surface
// interaction use an explicit operation order (mul / fma placement written out,
// built with -ffp-contract=off) so that hit decisions are bit-identical to the
// CPU oracle's.  Reference semantics restated (not copied):
//   transform_affine        include/mitsuba/core/transform.h:104-111,130-138
//   moeller_trumbore        include/mitsuba/render/mesh.h:357-380
//   closest-hit tie rule    include/mitsuba/render/kdtree.h:2424-2448
//   surface interaction     src/render/mesh.cpp:672-903
//   finalize                include/mitsuba/render/interaction.h:257-267,476-499
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HF_MAX_LEVELS 24

struct v3 {
    float x, y, z;
};

__device__ __forceinline__ v3 mk3(float x, float y, float z) { return v3{ x, y, z }; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return v3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return v3{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ v3 neg3(v3 a) { return v3{ -a.x, -a.y, -a.z }; }

// Dr.Jit dot(): x product first, then fmadd of y and z terms
__device__ __forceinline__ float dot3(v3 a, v3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
// Dr.Jit cross(): fmsub(a.yzx, b.zxy, a.zxy * b.yzx)
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
    return v3{ __builtin_fmaf(a.y, b.z, -(a.z * b.y)),
               __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)) };
}
__device__ __forceinline__ float rcp_ieee(float x) { return 1.0f / x; }
__device__ __forceinline__ float rsqrt_ieee(float x) { return 1.0f / __builtin_sqrtf(x); }
__device__ __forceinline__ v3 normalize3(v3 v) { return v * rsqrt_ieee(dot3(v, v)); }
// y += a * x, plain mul + add (adjoint accumulation)
__device__ __forceinline__ void axpy3(float a, v3 x, v3 &y) {
    y.x += a * x.x; y.y += a * x.y; y.z += a * x.z;
}
__device__ __forceinline__ v3 fma3(v3 a, float s, v3 c) {
    return v3{ __builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y), __builtin_fmaf(a.z, s, c.z) };
}

// row-major 3x4 affine; point: start from the translation column, fmadd columns 0..2
__device__ __forceinline__ v3 xform_point(const float *m, v3 p) {
    v3 r;
    r.x = __builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, __builtin_fmaf(m[0], p.x, m[3])));
    r.y = __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, __builtin_fmaf(m[4], p.x, m[7])));
    r.z = __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, __builtin_fmaf(m[8], p.x, m[11])));
    return r;
}
// vector: column 0 * x, then fmadd columns 1, 2
__device__ __forceinline__ v3 xform_vec(const float *m, v3 v) {
    v3 r;
    r.x = __builtin_fmaf(m[2], v.z, __builtin_fmaf(m[1], v.y, m[0] * v.x));
    r.y = __builtin_fmaf(m[6], v.z, __builtin_fmaf(m[5], v.y, m[4] * v.x));
    r.z = __builtin_fmaf(m[10], v.z, __builtin_fmaf(m[9], v.y, m[8] * v.x));
    return r;
}

// Device view of one heightfield: heights + min/max mip pyramid.
//
// Quadtree over 2^top x 2^top cells (top >= 1; cells beyond the grid do not exist).
// A node of level l covers 2^l x 2^l cells.  Levels 1..top are stored COARSE-FIRST in one
// padded array: depth k = top - l holds 2^k x 2^k nodes, row-major with pitch 2^k, at
// offset (4^k - 1)/3 + 1 (entry 0 is padding, the root is entry 1: the global range that bbox and the slab clip
// read; the beam sweep of the traversal kernel reads one depth of it, the level of its hand-off nodes, and parks
// the entries of the nodes along the beam in LDS).  Node (ix,iy) of level l holds
// (min z, max z) over the vertices of its existing cells; nodes without any existing cell hold
// (+inf, -inf) and fail every overlap test.
struct hf_dev_field {
    const float *h;    // W*H heights, row-major
    const float2 *mip; // (4^top - 1)/3 nodes
    const float4 *shear; // sheared bounds of the fine levels, 3 float4 per node (see hf_shear_rec)
    int32_t W, H;
    int32_t top;  // max(ceil(log2(max(W-1,H-1))), 1); mip[1] is the global (min,max)
    float s, sx, sy, iu, iv;
    float hx, hy; // 0.5 (W - 1), 0.5 (H - 1): cells per object unit (host-computed: a kernarg scalar instead of a hoisted -- and spilled -- vector register)
    int32_t flip;
    float to_world[12], to_object[12];
};

// number of existing nodes per row / column at level l
__host__ __device__ __forceinline__ int hf_level_w(int cells, int l) { return (cells + (1 << l) - 1) >> l; }
// offset of pyramid depth k: (4^k - 1)/3 + 1 (entry 0 is padding so that every depth >= 1 starts
// on an even index: the two children of a node that share a row are one aligned 16-byte load)
__host__ __device__ __forceinline__ uint32_t hf_depth_off(int k) { return (0x55555555u & ((1u << (2 * k)) - 1u)) + 1u; }

// Sheared bounds.  Min/max boxes are loose on steep terrain: a node on a slope spans a large
// z range although the surface stays close to a plane.  Nodes of levels 1..HF_SHEAR_TOP therefore
// also carry a plane  z = c + a (x - xc) + b (y - yc)  through their corner heights (x, y in cell
// units, (xc,yc) the node centre) and, per child, the exact range of  z - plane  over the child's
// vertices: the box test of the walk then runs in the sheared coordinate  w = z - plane, in which
// the ray is still a straight line.  Any plane is valid (the ranges are exact for the plane that is
// stored); record = { (a, b, c, |a|+|b|), (lo0, hi0, lo1, hi1), (lo2, hi2, lo3, hi3) }, children in
// actual order j = 2 jy + jx, absent children (+inf, -inf).  Same coarse-first indexing as the
// pyramid: node (ix,iy) of level L = depth k = top - L is record  hf_depth_off(k) - 1 + (iy << k) + ix.
#ifndef HF_SHEAR_TOP
#define HF_SHEAR_TOP 5
#endif
__host__ __device__ __forceinline__ size_t hf_shear_records(int top) { // depths 0 .. top-1 (levels 1 .. top)
    return (size_t) hf_depth_off(top) - 1u;
}

struct hf_hit {
    float t, u, v;
    uint32_t prim;
    bool hit;
};

// Moeller-Trumbore with the reference's operation order and inclusive tests.
__device__ __forceinline__ bool moeller_trumbore(v3 o, v3 d, float maxt, v3 p0, v3 p1, v3 p2,
                                                 float &t, float &u, float &v) {
    v3 e1 = p1 - p0, e2 = p2 - p0;
    v3 pvec = cross3(d, e2);
    float inv_det = rcp_ieee(dot3(e1, pvec));
    v3 tvec = o - p0;
    u = dot3(tvec, pvec) * inv_det;
    bool ok = (u >= 0.f) & (u <= 1.f);
    v3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * inv_det;
    ok = ok & (v >= 0.f) & (u + v <= 1.f);
    t = dot3(e2, qvec) * inv_det;
    ok = ok & (t >= 0.f) & (t <= maxt);
    return ok;
}

// minimum t; among exactly equal t the higher prim index (the brute-force loop's
// `t <= ray.maxt` lets the later primitive replace the earlier one)
__device__ __forceinline__ void best_update(hf_hit &b, float t, float u, float v, uint32_t prim) {
    if (!b.hit || t < b.t || (t == b.t && prim > b.prim)) {
        b.t = t; b.u = u; b.v = v; b.prim = prim; b.hit = true;
    }
}

// Both triangles of cell (cx,cy):  tri 0 = (v00, v10, v01),  tri 1 = (v11, v01, v10).
__device__ __forceinline__ bool test_cell(const hf_dev_field &f, int cx, int cy, float z00, float z10,
                                          float z01, float z11, v3 o, v3 d, float maxt, hf_hit &b) {
    const float x0 = __builtin_fmaf((float) cx, f.sx, -1.0f), x1 = __builtin_fmaf((float) (cx + 1), f.sx, -1.0f);
    const float y0 = __builtin_fmaf((float) cy, f.sy, -1.0f), y1 = __builtin_fmaf((float) (cy + 1), f.sy, -1.0f);
    const v3 v00 = mk3(x0, y0, z00), v10 = mk3(x1, y0, z10), v01 = mk3(x0, y1, z01), v11 = mk3(x1, y1, z11);
    const uint32_t prim = 2u * ((uint32_t) cy * (uint32_t) (f.W - 1) + (uint32_t) cx);
    float t, u, v;
    bool any = false;
    if (moeller_trumbore(o, d, maxt, v00, v10, v01, t, u, v)) { best_update(b, t, u, v, prim); any = true; }
    if (moeller_trumbore(o, d, maxt, v11, v01, v10, t, u, v)) { best_update(b, t, u, v, prim + 1u); any = true; }
    return any;
}

// (row, col) of the three vertices of a primitive
__device__ __forceinline__ void prim_vertex_ids(const hf_dev_field &f, uint32_t prim, int vi[3], int vj[3]) {
    const uint32_t cell = prim >> 1, cw = (uint32_t) (f.W - 1);
    const int cy = (int) (cell / cw), cx = (int) (cell - (uint32_t) cy * cw);
    if ((prim & 1u) == 0) {
        vi[0] = cy;     vj[0] = cx;
        vi[1] = cy;     vj[1] = cx + 1;
        vi[2] = cy + 1; vj[2] = cx;
    } else {
        vi[0] = cy + 1; vj[0] = cx + 1;
        vi[1] = cy + 1; vj[1] = cx;
        vi[2] = cy;     vj[2] = cx + 1;
    }
}

__device__ __forceinline__ float signf_(float x) { return x >= 0.f ? 1.f : -1.f; }
__device__ __forceinline__ float mulsign(float a, float b) { return b >= 0.f ? a : -a; }

// coordinate_system(), include/mitsuba/core/vector.h:116-136
__device__ __forceinline__ void coordinate_system(v3 n, v3 &s, v3 &t) {
    const float sign = signf_(n.z);
    const float a = -rcp_ieee(sign + n.z);
    const float b = n.x * n.y * a;
    s = mk3(mulsign(n.x * n.x * a, n.z) + 1.f, mulsign(b, n.z), mulsign(-n.x, n.z));
    t = mk3(b, __builtin_fmaf(n.y, n.y * a, sign), -n.y);
}

struct hf_si_rec {
    float t;
    v3 p, n;
    float uv0, uv1;
    v3 sh_n, dp_du, dp_dv;
    float boundary_test;
    v3 sh_s, sh_t, wi;
};

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// Which of the hit triangle's three edges (k = 0: P0-P1, 1: P1-P2, 2: P2-P0) are SILHOUETTE edges for a ray of
// object-space direction od: the neighbour across the edge does not exist (border of the grid) or faces the ray
// the other way.  A grid triangle with slopes (zx, zy) faces the ray by sign(od.z - zx od.x - zy od.y).  Interior
// edges between triangles that face the ray the same way are no visibility boundary.
__device__ __forceinline__ bool tri_faces(float zx, float zy, v3 od) {
    return __builtin_fmaf(-zy, od.y, __builtin_fmaf(-zx, od.x, od.z)) >= 0.f;
}
__device__ __forceinline__ uint32_t silhouette_edges(const hf_dev_field &f, uint32_t prim, v3 od) {
    const uint32_t cw = (uint32_t) (f.W - 1);
    const int cy = (int) ((prim >> 1) / cw), cx = (int) ((prim >> 1) - (uint32_t) cy * cw);
    const float s = f.s, isx = 1.0f / f.sx, isy = 1.0f / f.sy;
    auto hz = [&](int i, int j) { return f.h[(size_t) i * f.W + j] * s; };
    const float z00 = hz(cy, cx), z10 = hz(cy, cx + 1), z01 = hz(cy + 1, cx), z11 = hz(cy + 1, cx + 1);
    const bool f0 = tri_faces((z10 - z00) * isx, (z01 - z00) * isy, od); // tri 0 = (v00, v10, v01)
    const bool f1 = tri_faces((z11 - z01) * isx, (z11 - z10) * isy, od); // tri 1 = (v11, v01, v10)
    uint32_t m = 0;
    if ((prim & 1u) == 0) {
        if (cy == 0 || tri_faces((z10 - z00) * isx, (z10 - hz(cy - 1, cx + 1)) * isy, od) != f0) m |= 1u; // bottom
        if (f1 != f0) m |= 2u;                                                                             // diagonal
        if (cx == 0 || tri_faces((z01 - hz(cy + 1, cx - 1)) * isx, (z01 - z00) * isy, od) != f0) m |= 4u; // left
    } else {
        if (cy + 2 > f.H - 1 || tri_faces((z11 - z01) * isx, (hz(cy + 2, cx) - z01) * isy, od) != f1) m |= 1u; // top
        if (f0 != f1) m |= 2u;
        if (cx + 2 > f.W - 1 || tri_faces((hz(cy, cx + 2) - z10) * isx, (z11 - z10) * isy, od) != f1) m |= 4u; // right
    }
    return m;
}

// boundary test of the height field = the SDF of the hit point in an equilateral reference triangle
// (src/render/mesh.cpp:845-890), restricted to the silhouette edges; 1 (the incentre value) when there is none
__device__ __forceinline__ float boundary_test_flat(v3 p, v3 p0, v3 dp0, v3 dp1, uint32_t edges) {
    if (edges == 0u) return 1.0f;
    const v3 rel = p - p0;
    const float bb1 = dot3(dp0, rel), bb2 = dot3(dp1, rel);
    const float a11 = dot3(dp0, dp0), a12 = dot3(dp0, dp1), a22 = dot3(dp1, dp1);
    const float inv_det = rcp_ieee(a11 * a22 - a12 * a12);
    const float u = __builtin_fmaf(a22, bb1, -(a12 * bb2)) * inv_det;
    const float v = __builtin_fmaf(-a12, bb1, a11 * bb2) * inv_det;
    const float w = 1.f - u - v;
    const float h3 = 0.5f * __builtin_sqrtf(3.f);
    const float tpx[3] = { 0.f, 1.f, 0.5f }, tpy[3] = { 0.f, 0.f, h3 };
    const float qx = tpx[0] * w + tpx[1] * u + tpx[2] * v, qy = tpy[0] * w + tpy[1] * u + tpy[2] * v;
    float dmin = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!((edges >> k) & 1u)) continue;
        const int k1 = (k + 1) % 3;
        const float ex = tpx[k1] - tpx[k], ey = tpy[k1] - tpy[k];
        const float vx = qx - tpx[k], vy = qy - tpy[k];
        const float c = clamp01(__builtin_fmaf(vy, ey, vx * ex) / __builtin_fmaf(ey, ey, ex * ex));
        const float px = vx - ex * c, py = vy - ey * c;
        dmin = fminf(dmin, __builtin_fmaf(py, py, px * px));
    }
    float dist = __builtin_sqrtf(dmin);
    dist /= __builtin_sqrtf(3.f) / 6.f;
    return dist;
}

// world-space vertices + texcoords of a primitive
__device__ __forceinline__ void prim_world(const hf_dev_field &f, uint32_t prim, v3 P[3], float U[3], float V[3],
                                           int vi[3], int vj[3]) {
    prim_vertex_ids(f, prim, vi, vj);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const v3 q = mk3(__builtin_fmaf((float) vj[k], f.sx, -1.0f), __builtin_fmaf((float) vi[k], f.sy, -1.0f),
                         f.h[(size_t) vi[k] * f.W + vj[k]] * f.s);
        P[k] = xform_point(f.to_world, q);
        U[k] = (float) vj[k] * f.iu;
        V[k] = (float) vi[k] * f.iv;
    }
}

// Shape::compute_surface_interaction + finalize_surface_interaction for one valid hit.
// The fields are handed to `out` as soon as they are final (out.t(..), out.p(..), ...): the record sink below collects
// them into an hf_si_rec; the fused traversal kernel stores each one straight away, so that the whole record is
// never live in registers at once.
template <typename Out>
__device__ __forceinline__ void compute_si_to(const hf_dev_field &f, v3 o, v3 d, float t_in, float b1, float b2,
                                              uint32_t prim, uint32_t flags, Out &out) {
    v3 P[3];
    float U[3], V[3];
    int vi[3], vj[3];
    prim_world(f, prim, P, U, V, vi, vj);
    const float b0 = 1.f - b1 - b2;
    const v3 dp0 = P[1] - P[0], dp1 = P[2] - P[0];
    const v3 p = mk3(__builtin_fmaf(P[0].x, b0, __builtin_fmaf(P[1].x, b1, P[2].x * b2)),
                     __builtin_fmaf(P[0].y, b0, __builtin_fmaf(P[1].y, b1, P[2].y * b2)),
                     __builtin_fmaf(P[0].z, b0, __builtin_fmaf(P[1].z, b1, P[2].z * b2)));
    float t = t_in;
    if (flags & 0x80u) { // FollowShape: t re-derived from the glued point (mesh.cpp:748-752)
        const v3 po = p - o;
        t = __builtin_sqrtf(dot3(po, po) / dot3(d, d));
    }
    out.t(t);
    out.p(p);
    if (flags & 0x40u)
        // 0x10000 (HF_RAY_BOUNDARY_ALL_EDGES, a libhf extension bit): the reference Mesh's per-triangle SDF over all
        // three edges (mesh.cpp:845-890, values in [0, 1]) instead of the silhouette edges only
        out.boundary_test(boundary_test_flat(p, P[0], dp0, dp1, (flags & 0x10000u) ? 7u : silhouette_edges(f, prim, xform_vec(f.to_object, d))));
    else
        out.boundary_test(0.f);
    v3 n = normalize3(cross3(dp0, dp1));
    float uv0 = b1, uv1 = b2;
    v3 dp_du, dp_dv;
    coordinate_system(n, dp_du, dp_dv);
    if (flags & (0x2u | 0x4u)) {
        uv0 = __builtin_fmaf(U[2], b2, __builtin_fmaf(U[1], b1, U[0] * b0));
        uv1 = __builtin_fmaf(V[2], b2, __builtin_fmaf(V[1], b1, V[0] * b0));
        if (flags & 0x4u) {
            const float du0 = U[1] - U[0], dv0 = V[1] - V[0], du1 = U[2] - U[0], dv1 = V[2] - V[0];
            const float det = __builtin_fmaf(du0, dv1, -(dv0 * du1));
            const float inv_det = rcp_ieee(det);
            if (det != 0.f) {
                dp_du = mk3(__builtin_fmaf(dv1, dp0.x, -(dv0 * dp1.x)) * inv_det,
                            __builtin_fmaf(dv1, dp0.y, -(dv0 * dp1.y)) * inv_det,
                            __builtin_fmaf(dv1, dp0.z, -(dv0 * dp1.z)) * inv_det);
                dp_dv = mk3(__builtin_fmaf(-du1, dp0.x, du0 * dp1.x) * inv_det,
                            __builtin_fmaf(-du1, dp0.y, du0 * dp1.y) * inv_det,
                            __builtin_fmaf(-du1, dp0.z, du0 * dp1.z) * inv_det);
            }
        }
    }
    out.uv(uv0, uv1);
    out.dp_dv(dp_dv);
    if (f.flip) n = neg3(n);
    out.n(n); // n and sh_n
    v3 sh_s = mk3(0.f, 0.f, 0.f), sh_t = mk3(0.f, 0.f, 0.f);
    if (flags & 0x8u) { // initialize_sh_frame: Gram-Schmidt on dp_du
        const float nd = -dot3(n, dp_du);
        sh_s = normalize3(fma3(n, nd, dp_du));
        if (dp_du.x == 0.f && dp_du.y == 0.f && dp_du.z == 0.f) {
            v3 dummy;
            coordinate_system(n, sh_s, dummy);
        }
        sh_t = cross3(n, sh_s);
    }
    out.dp_du(dp_du);
    out.sh_s(sh_s);
    out.sh_t(sh_t);
    const v3 md = neg3(d);
    out.wi(mk3(dot3(md, sh_s), dot3(md, sh_t), dot3(md, n)));
}

struct hf_si_rec_sink {
    hf_si_rec &si;
    __device__ __forceinline__ void t(float v) { si.t = v; }
    __device__ __forceinline__ void p(v3 v) { si.p = v; }
    __device__ __forceinline__ void boundary_test(float v) { si.boundary_test = v; }
    __device__ __forceinline__ void uv(float a, float b) { si.uv0 = a; si.uv1 = b; }
    __device__ __forceinline__ void dp_du(v3 v) { si.dp_du = v; }
    __device__ __forceinline__ void dp_dv(v3 v) { si.dp_dv = v; }
    __device__ __forceinline__ void n(v3 v) { si.n = v; si.sh_n = v; }
    __device__ __forceinline__ void sh_s(v3 v) { si.sh_s = v; }
    __device__ __forceinline__ void sh_t(v3 v) { si.sh_t = v; }
    __device__ __forceinline__ void wi(v3 v) { si.wi = v; }
};
__device__ __forceinline__ void compute_si(const hf_dev_field &f, v3 o, v3 d, float t_in, float b1, float b2,
                                           uint32_t prim, uint32_t flags, hf_si_rec &si) {
    hf_si_rec_sink sink = { si };
    compute_si_to(f, o, d, t_in, b1, b2, prim, flags, sink);
}
