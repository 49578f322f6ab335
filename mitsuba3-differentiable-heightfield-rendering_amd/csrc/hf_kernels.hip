// hf_kernels.hip -- gfx950 kernels of libhf + their launchers.
//
//   hf_mip_level1_kernel / hf_mip_reduce_kernel   min/max mip pyramid
//   hf_trace_kernel<MODE>                          hierarchical min/max-mip traversal:
//        MODE 0 closest hit  -> PreliminaryIntersection   (row a1)
//        MODE 1 any hit      -> ray_test                   (row a2)
//        MODE 2 closest hit + fused surface interaction    (rows a1+a4)
//   hf_si_kernel                                   compute_surface_interaction (row a4)
//   hf_adjoint_kernel                              reverse mode of a4, atomic scatter (row a5)
//
// Traversal = scan of the cells along a Morton curve mirrored so the ray direction
// is non-negative on both axes ("order space"), with whole quadtree nodes (X,Y,L)
// skipped when the *fat* ray segment [0,t_hi] misses the node's box
// [x..x+2^L] x [y..y+2^L] x [min z, max z] (mip level L).  The visited set is a
// conservative superset of the cells the ray can hit; the per-triangle test and
// the tie rule are order independent, so the result equals the brute force's.
#include "hf_device.h"
#include "hf_launch.h"

#define HF_BLOCK 256
#define HF_LDS_LEVELS 6           // top mip levels staged in LDS: <= 1+4+16+64+256+1024 nodes
#define HF_LDS_NODES 1365

// ---------------------------------------------------------------------------------
// min/max mip pyramid
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(HF_BLOCK) void hf_mip_level1_kernel(const float *__restrict__ h, int W, int H, float s,
                                                                float2 *__restrict__ out, int mw, int mh) {
    const int idx = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (idx >= mw * mh) return;
    const int iy = idx / mw, ix = idx - iy * mw;
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = 2 * iy + a;
        if (i >= H) break;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int j = 2 * ix + b;
            if (j < W) {
                const float z = h[(size_t) i * W + j] * s;
                mn = fminf(mn, z); mx = fmaxf(mx, z);
            }
        }
    }
    out[idx] = make_float2(mn, mx);
}

__global__ __launch_bounds__(HF_BLOCK) void hf_mip_reduce_kernel(const float2 *__restrict__ in, int pw, int ph,
                                                                float2 *__restrict__ out, int mw, int mh) {
    const int idx = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (idx >= mw * mh) return;
    const int iy = idx / mw, ix = idx - iy * mw;
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int jy = 2 * iy + a, jx = 2 * ix + b;
            if (jy < ph && jx < pw) {
                const float2 c = in[(size_t) jy * pw + jx];
                mn = fminf(mn, c.x); mx = fmaxf(mx, c.y);
            }
        }
    out[idx] = make_float2(mn, mx);
}

void hf_launch_build_mips(const hf_dev_field &f, float2 *mip, hipStream_t stream) {
    for (int l = 1; l <= f.nlev; ++l) {
        const int n = f.mw[l] * f.mh[l];
        const int grid = (n + HF_BLOCK - 1) / HF_BLOCK;
        if (l == 1)
            hipLaunchKernelGGL(hf_mip_level1_kernel, dim3(grid), dim3(HF_BLOCK), 0, stream, f.h, f.W, f.H, f.s,
                               mip + f.moff[1], f.mw[1], f.mh[1]);
        else
            hipLaunchKernelGGL(hf_mip_reduce_kernel, dim3(grid), dim3(HF_BLOCK), 0, stream,
                               (const float2 *) (mip + f.moff[l - 1]), f.mw[l - 1], f.mh[l - 1], mip + f.moff[l],
                               f.mw[l], f.mh[l]);
    }
}

// ---------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------
struct hf_lds_mips {
    float2 node[HF_LDS_NODES];
    uint32_t goff[HF_MAX_LEVELS + 1]; // global offset of level l
    uint32_t loff[HF_MAX_LEVELS + 1]; // LDS offset of level l (valid for l >= lo)
    int lo;                           // lowest staged level
};

__device__ __forceinline__ void stage_mips(const hf_dev_field &f, hf_lds_mips &s) {
    const int tid = threadIdx.x;
    int lo = f.nlev - (HF_LDS_LEVELS - 1);
    if (lo < 1) lo = 1;
    if (tid == 0) {
        uint32_t acc = 0;
        for (int l = f.nlev; l >= 1; --l) {
            s.goff[l] = f.moff[l];
            s.loff[l] = acc;
            if (l >= lo) acc += (uint32_t) (f.mw[l] * f.mh[l]);
        }
        s.lo = lo;
    }
    uint32_t acc = 0;
    for (int l = f.nlev; l >= lo; --l) {
        const int cnt = f.mw[l] * f.mh[l];
        for (int k = tid; k < cnt; k += HF_BLOCK) s.node[acc + k] = f.mip[f.moff[l] + k];
        acc += (uint32_t) cnt;
    }
    __syncthreads();
}

template <bool ANY>
__device__ __forceinline__ void trace_ray(const hf_dev_field &f, const hf_lds_mips &s, v3 o, v3 d, float maxt,
                                          hf_hit &best) {
    best.hit = false; best.t = __builtin_inff(); best.u = 0.f; best.v = 0.f; best.prim = 0u;
    const v3 oo = xform_point(f.to_object, o), od = xform_vec(f.to_object, d);
    // non-finite input or NaN/negative maxt: miss (also bounds the scan below)
    {
        const float chk = (oo.x + oo.y + oo.z) + (od.x + od.y + od.z);
        if (!(__builtin_fabsf(chk) < __builtin_inff()) || !(maxt >= 0.f)) return;
    }
    const int cw = f.W - 1, ch = f.H - 1, top = f.top;
    const float hx = 0.5f * (float) cw, hy = 0.5f * (float) ch;
    const float2 zr = s.node[0]; // root of the pyramid = global (min z, max z)
    const float zspan = fmaxf(zr.y - zr.x, fmaxf(__builtin_fabsf(zr.x), __builtin_fabsf(zr.y)));
    const float mz0 = 1e-5f * zspan + 1e-30f;

    // clip against the inflated object-space bound (slab test, bbox.h:302-327)
    float tin = 0.f, tout = maxt;
    {
        const float lo[3] = { -1.f - 1e-4f, -1.f - 1e-4f, zr.x - mz0 };
        const float hi[3] = { 1.f + 1e-4f, 1.f + 1e-4f, zr.y + mz0 };
        const float oc[3] = { oo.x, oo.y, oo.z }, dc[3] = { od.x, od.y, od.z };
        bool outside = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (dc[k] == 0.f) {
                outside |= (oc[k] < lo[k]) | (oc[k] > hi[k]);
            } else {
                const float r = 1.0f / dc[k];
                const float t1 = (lo[k] - oc[k]) * r, t2 = (hi[k] - oc[k]) * r;
                tin = fmaxf(tin, fminf(t1, t2));
                tout = fminf(tout, fmaxf(t1, t2));
            }
        }
        tin = tin - __builtin_fabsf(tin) * 1e-6f;
        tin = fmaxf(tin, 0.f);
        tout = tout + __builtin_fabsf(tout) * 1e-6f;
        if (outside || !(tin <= tout)) return;
    }

    // traversal ray in cell units, re-based at t = tin, mirrored into order space
    const bool fx = od.x < 0.f, fy = od.y < 0.f;
    const float Wp = (float) (1 << top);
    float gx = (__builtin_fmaf(tin, od.x, oo.x) + 1.f) * hx, gy = (__builtin_fmaf(tin, od.y, oo.y) + 1.f) * hy;
    const float gz = __builtin_fmaf(tin, od.z, oo.z);
    float dx = od.x * hx, dy = od.y * hy;
    const float dz = od.z;
    if (fx) { gx = Wp - gx; dx = -dx; }
    if (fy) { gy = Wp - gy; dy = -dy; }
    const float idx = 1.0f / dx, idy = 1.0f / dy; // +inf for axis-parallel rays
    const float reach = __builtin_fabsf(oo.x) + __builtin_fabsf(oo.y) +
                        tin * (__builtin_fabsf(od.x) + __builtin_fabsf(od.y)) + 2.f;
    const float m = 0.015625f + 4.8e-7f * reach * fmaxf(hx, hy);
    const float mz = mz0 + 4.8e-7f * (__builtin_fabsf(oo.z) + tin * __builtin_fabsf(od.z) + zspan);
    const float gxm = gx + m, gxp = gx - m, gym = gy + m, gyp = gy - m;
    float thi = tout - tin;
    thi = thi + thi * 1e-6f + 1e-30f;
    const uint32_t fxm = fx ? ((1u << top) - 1u) : 0u, fym = fy ? ((1u << top) - 1u) : 0u;

    uint32_t X = 0, Y = 0;
    int L = top;
    bool done = false;
    for (;;) {
        // ---- phase 1: walk quadtree nodes until this lane holds a candidate cell ----
        bool leaf = false;
        int lx = 0, ly = 0;
        float z00 = 0.f, z10 = 0.f, z01 = 0.f, z11 = 0.f;
        while (!done && !leaf) {
            const float S = (float) (1u << L);
            const float fX = (float) X * S, fY = (float) Y * S;
            // order-space direction is >= 0: entry = low faces, exit = high faces.
            // fmaxf/fminf drop the NaN of 0 * inf (origin exactly on a face plane).
            const float t0 = fmaxf(fmaxf((fX - gxm) * idx, (fY - gym) * idy), 0.f);
            const float t1 = fminf(fminf((fX + S - gxp) * idx, (fY + S - gyp) * idy), thi);
            bool overlap = false;
            const int ix = (int) (X ^ (fxm >> L)), iy = (int) (Y ^ (fym >> L));
            if (t0 <= t1) {
                float zlo = 0.f, zhi = 0.f;
                bool inside;
                if (L == 0) {
                    inside = (ix < cw) & (iy < ch);
                    if (inside) {
                        const float *r0 = f.h + (size_t) iy * f.W + ix;
                        z00 = r0[0] * f.s; z10 = r0[1] * f.s;
                        z01 = r0[f.W] * f.s; z11 = r0[f.W + 1] * f.s;
                        zlo = fminf(fminf(z00, z10), fminf(z01, z11));
                        zhi = fmaxf(fmaxf(z00, z10), fmaxf(z01, z11));
                    }
                } else {
                    const int w = (cw + (1 << L) - 1) >> L, hh = (ch + (1 << L) - 1) >> L;
                    inside = (ix < w) & (iy < hh);
                    if (inside) {
                        float2 c;
                        if (L >= s.lo) c = s.node[s.loff[L] + (uint32_t) (iy * w + ix)];
                        else           c = f.mip[s.goff[L] + (uint32_t) iy * (uint32_t) w + (uint32_t) ix];
                        zlo = c.x; zhi = c.y;
                    }
                }
                if (inside) {
                    const float za = __builtin_fmaf(t0, dz, gz), zb = __builtin_fmaf(t1, dz, gz);
                    overlap = (fminf(za, zb) - mz <= zhi) & (fmaxf(za, zb) + mz >= zlo);
                }
            }
            if (overlap && L > 0) { // descend to the first child in order space
                X <<= 1; Y <<= 1; --L;
                continue;
            }
            if (overlap) { leaf = true; lx = ix; ly = iy; }
            // advance: climb while this is the last (k=3) child, then step to the next sibling
            const int c = __builtin_ctz(~(X & Y));
            X >>= c; Y >>= c; L += c;
            if (L >= top) done = true;
            else if ((X & 1u) == 0u) X |= 1u;
            else { X &= ~1u; Y |= 1u; }
        }
        if (!leaf) break;
        // ---- phase 2: the two triangles of the candidate cell (spec arithmetic) ----
        if (test_cell(f, lx, ly, z00, z10, z01, z11, oo, od, maxt, best)) {
            if (ANY) break;
            float tb = best.t - tin;
            tb = tb + __builtin_fabsf(tb) * 1e-6f + 1e-30f;
            thi = fminf(thi, tb);
        }
    }
}

struct hf_rays_dev {
    const float *o[3];
    const float *d[3];
    const float *maxt;
};
struct hf_pi_dev {
    float *t, *u, *v;
    uint32_t *prim;
};
struct hf_si_dev {
    float *t, *p[3], *n[3], *uv[2], *sh_n[3], *dp_du[3], *dp_dv[3], *bt, *sh_s[3], *sh_t[3], *wi[3];
};

__device__ __forceinline__ void st(float *p, size_t i, float v) { if (p) p[i] = v; }
__device__ __forceinline__ void st3(float *const p[3], size_t i, v3 v) {
    if (p[0]) p[0][i] = v.x;
    if (p[1]) p[1][i] = v.y;
    if (p[2]) p[2][i] = v.z;
}

__device__ __forceinline__ void store_si(const hf_si_dev &out, size_t i, const hf_si_rec &si, uint32_t flags) {
    st(out.t, i, si.t);
    st3(out.p, i, si.p);
    st3(out.n, i, si.n);
    st(out.uv[0], i, si.uv0); st(out.uv[1], i, si.uv1);
    st3(out.sh_n, i, si.sh_n);
    st3(out.dp_du, i, si.dp_du);
    st3(out.dp_dv, i, si.dp_dv);
    if (flags & 0x40u) st(out.bt, i, si.boundary_test);
    st3(out.sh_s, i, si.sh_s);
    st3(out.sh_t, i, si.sh_t);
    st3(out.wi, i, si.wi);
}

// zero-initialised record for inactive / missed lanes (interaction.h:479-499, 667-673)
__device__ __forceinline__ void miss_si(hf_si_rec &si, v3 d, uint32_t flags) {
    const v3 z = mk3(0.f, 0.f, 0.f);
    si.t = __builtin_inff();
    si.p = z; si.n = z; si.uv0 = 0.f; si.uv1 = 0.f; si.sh_n = z; si.dp_du = z; si.dp_dv = z;
    si.boundary_test = (flags & 0x40u) ? 1e8f : 0.f;
    si.sh_s = z; si.sh_t = z;
    si.wi = neg3(d);
}

template <int MODE>
__global__ __launch_bounds__(HF_BLOCK) void hf_trace_kernel(hf_dev_field f, size_t n, hf_rays_dev rays,
                                                            const uint8_t *__restrict__ active, hf_pi_dev pi,
                                                            uint8_t *__restrict__ hit_out, hf_si_dev sio,
                                                            uint32_t flags) {
    __shared__ hf_lds_mips s;
    stage_mips(f, s);
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        const v3 o = mk3(rays.o[0][i], rays.o[1][i], rays.o[2][i]);
        const v3 d = mk3(rays.d[0][i], rays.d[1][i], rays.d[2][i]);
        const float maxt = rays.maxt[i];
        hf_hit best;
        best.hit = false; best.t = __builtin_inff(); best.u = 0.f; best.v = 0.f; best.prim = 0u;
        const bool act = active ? (active[i] != 0) : true;
        if (act) trace_ray<MODE == 1>(f, s, o, d, maxt, best);
        if (MODE == 1) {
            hit_out[i] = best.hit ? 1 : 0;
        } else {
            if (pi.t) pi.t[i] = best.hit ? best.t : __builtin_inff();
            if (pi.u) pi.u[i] = best.hit ? best.u : 0.f;
            if (pi.v) pi.v[i] = best.hit ? best.v : 0.f;
            if (pi.prim) pi.prim[i] = best.hit ? best.prim : 0u;
            if (MODE == 2) {
                hf_si_rec si;
                if (best.hit) compute_si(f, o, d, best.t, best.u, best.v, best.prim, flags, si);
                else          miss_si(si, d, flags);
                store_si(sio, i, si, flags);
            }
        }
    }
}

static int grid_for(size_t n) {
    size_t blocks = (n + HF_BLOCK - 1) / HF_BLOCK;
    const size_t cap = 256 * 8; // 256 CUs x 8 resident blocks of 256 threads
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int) blocks;
}

static hf_rays_dev to_dev(const hf_rays_t *r) {
    hf_rays_dev d;
    for (int k = 0; k < 3; ++k) { d.o[k] = r->o[k]; d.d[k] = r->d[k]; }
    d.maxt = r->maxt;
    return d;
}
static hf_si_dev to_dev(const hf_si_t *s) {
    hf_si_dev d;
    d.t = s->t; d.bt = s->boundary_test;
    for (int k = 0; k < 3; ++k) {
        d.p[k] = s->p[k]; d.n[k] = s->n[k]; d.sh_n[k] = s->sh_n[k]; d.dp_du[k] = s->dp_du[k];
        d.dp_dv[k] = s->dp_dv[k]; d.sh_s[k] = s->sh_s[k]; d.sh_t[k] = s->sh_t[k]; d.wi[k] = s->wi[k];
    }
    d.uv[0] = s->uv[0]; d.uv[1] = s->uv[1];
    return d;
}

void hf_launch_trace(int mode, const hf_dev_field &f, size_t n, const hf_rays_t *rays, const uint8_t *active,
                     const hf_pi_t *pi, uint8_t *hit, const hf_si_t *si, uint32_t flags, hipStream_t stream) {
    if (n == 0) return;
    hf_pi_dev p = { nullptr, nullptr, nullptr, nullptr };
    if (pi) { p.t = pi->t; p.u = pi->prim_uv[0]; p.v = pi->prim_uv[1]; p.prim = pi->prim_index; }
    hf_si_dev sd;
    memset(&sd, 0, sizeof(sd));
    if (si) sd = to_dev(si);
    const hf_rays_dev r = to_dev(rays);
    const dim3 grid(grid_for(n)), block(HF_BLOCK);
    if (mode == 0)
        hipLaunchKernelGGL(hf_trace_kernel<0>, grid, block, 0, stream, f, n, r, active, p, hit, sd, flags);
    else if (mode == 1)
        hipLaunchKernelGGL(hf_trace_kernel<1>, grid, block, 0, stream, f, n, r, active, p, hit, sd, flags);
    else
        hipLaunchKernelGGL(hf_trace_kernel<2>, grid, block, 0, stream, f, n, r, active, p, hit, sd, flags);
}

// ---------------------------------------------------------------------------------
// surface interaction from (ray, pi)
// ---------------------------------------------------------------------------------
struct hf_pi_cdev {
    const float *t, *u, *v;
    const uint32_t *prim;
};

__global__ __launch_bounds__(HF_BLOCK) void hf_si_kernel(hf_dev_field f, size_t n, hf_rays_dev rays,
                                                         hf_pi_cdev pi, const uint8_t *__restrict__ active,
                                                         hf_si_dev sio, uint32_t flags) {
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        const v3 o = mk3(rays.o[0][i], rays.o[1][i], rays.o[2][i]);
        const v3 d = mk3(rays.d[0][i], rays.d[1][i], rays.d[2][i]);
        const float t = pi.t[i];
        const bool act = (active ? (active[i] != 0) : true) && (t != __builtin_inff());
        hf_si_rec si;
        if (act) compute_si(f, o, d, t, pi.u[i], pi.v[i], pi.prim[i], flags, si);
        else     miss_si(si, d, flags);
        store_si(sio, i, si, flags);
    }
}

void hf_launch_si(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                  const uint8_t *active, const hf_si_t *si, uint32_t flags, hipStream_t stream) {
    if (n == 0) return;
    const hf_pi_cdev p = { pi->t, pi->prim_uv[0], pi->prim_uv[1], pi->prim_index };
    hipLaunchKernelGGL(hf_si_kernel, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, f, n, to_dev(rays), p, active,
                       to_dev(si), flags);
}

// ---------------------------------------------------------------------------------
// adjoint: reverse mode of compute_si, atomic scatter of dL/dheight
// ---------------------------------------------------------------------------------
struct hf_grad_dev {
    const float *t, *p[3], *n[3], *uv[2], *sh_n[3], *dp_du[3], *dp_dv[3];
};
__device__ __forceinline__ float ld(const float *p, size_t i) { return p ? p[i] : 0.f; }
__device__ __forceinline__ v3 ld3(const float *const p[3], size_t i) { return mk3(ld(p[0], i), ld(p[1], i), ld(p[2], i)); }

__global__ __launch_bounds__(HF_BLOCK) void hf_adjoint_kernel(hf_dev_field f, size_t n, hf_rays_dev rays,
                                                              hf_pi_cdev pi, const uint8_t *__restrict__ active,
                                                              hf_grad_dev g, uint32_t flags,
                                                              float *__restrict__ grad_h, float *go0, float *go1,
                                                              float *go2, float *gd0, float *gd1, float *gd2) {
    const bool follow = (flags & 0x80u) != 0, detach = (flags & 0x100u) != 0;
    const bool tex = (flags & (0x2u | 0x4u)) != 0;
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        const float t_in = pi.t[i];
        const bool act = (active ? (active[i] != 0) : true) && (t_in != __builtin_inff());
        v3 go = mk3(0.f, 0.f, 0.f), gd = mk3(0.f, 0.f, 0.f);
        if (act) {
            const v3 o = mk3(rays.o[0][i], rays.o[1][i], rays.o[2][i]);
            const v3 d = mk3(rays.d[0][i], rays.d[1][i], rays.d[2][i]);
            const float b1 = pi.u[i], b2 = pi.v[i], b0 = 1.f - b1 - b2;
            const uint32_t prim = pi.prim[i];
            v3 P[3];
            float U[3], V[3];
            int vi[3], vj[3];
            prim_world(f, prim, P, U, V, vi, vj);
            const v3 dp0 = P[1] - P[0], dp1 = P[2] - P[0];
            const v3 p = mk3(__builtin_fmaf(P[0].x, b0, __builtin_fmaf(P[1].x, b1, P[2].x * b2)),
                             __builtin_fmaf(P[0].y, b0, __builtin_fmaf(P[1].y, b1, P[2].y * b2)),
                             __builtin_fmaf(P[0].z, b0, __builtin_fmaf(P[1].z, b1, P[2].z * b2)));
            const v3 z3 = mk3(0.f, 0.f, 0.f);
            v3 gP0 = z3, gP1 = z3, gP2 = z3, gdp0 = z3, gdp1 = z3;
            v3 gp = ld3(g.p, i);
            const float gt = ld(g.t, i);

            // dp_du / dp_dv from the (constant) texcoord differences
            if (flags & 0x4u) {
                const float du0 = U[1] - U[0], dv0 = V[1] - V[0], du1 = U[2] - U[0], dv1 = V[2] - V[0];
                const float det = __builtin_fmaf(du0, dv1, -(dv0 * du1));
                const float inv_det = rcp_ieee(det);
                if (det != 0.f) {
                    const v3 gu_ = ld3(g.dp_du, i), gv_ = ld3(g.dp_dv, i);
                    axpy3(dv1 * inv_det, gu_, gdp0);
                    axpy3(-dv0 * inv_det, gu_, gdp1);
                    axpy3(-du1 * inv_det, gv_, gdp0);
                    axpy3(du0 * inv_det, gv_, gdp1);
                }
            }
            // n = sh_n = +-normalize(cross(dp0, dp1))
            {
                const v3 N = cross3(dp0, dp1);
                const float r = rsqrt_ieee(dot3(N, N));
                const v3 nn = N * r;
                const float sgn = f.flip ? -1.f : 1.f;
                const v3 a = ld3(g.n, i), b = ld3(g.sh_n, i);
                const v3 gn = mk3(sgn * (a.x + b.x), sgn * (a.y + b.y), sgn * (a.z + b.z));
                const float proj = dot3(nn, gn);
                const v3 gN = mk3((gn.x - nn.x * proj) * r, (gn.y - nn.y * proj) * r, (gn.z - nn.z * proj) * r);
                axpy3(1.f, cross3(dp1, gN), gdp0);
                axpy3(1.f, cross3(gN, dp0), gdp1);
            }
            // FollowShape: t = sqrt(|p-o|^2/|d|^2) feeds p, o, d
            if (follow) {
                const v3 po = p - o;
                const float dd = dot3(d, d), tt = __builtin_sqrtf(dot3(po, po) / dd);
                const float c = gt / (tt * dd);
                axpy3(c, po, gp);
                axpy3(-c, po, go);
                axpy3(-gt * tt / dd, d, gd);
            }
            // p = sum b_k P_k, uv = sum b_k uv_k
            const float guv0 = ld(g.uv[0], i), guv1 = ld(g.uv[1], i);
            float gb0 = dot3(gp, P[0]), gb1 = dot3(gp, P[1]), gb2 = dot3(gp, P[2]);
            if (tex) {
                gb0 += guv0 * U[0] + guv1 * V[0];
                gb1 += guv0 * U[1] + guv1 * V[1];
                gb2 += guv0 * U[2] + guv1 * V[2];
            }
            axpy3(b0, gp, gP0); axpy3(b1, gp, gP1); axpy3(b2, gp, gP2);
            float gu = gb1 - gb0, gv = gb2 - gb0;
            if (!tex) { gu += guv0; gv += guv1; }

            if (!follow) { // reverse of the differentiable Moeller-Trumbore (t_d, prim_uv_d)
                const v3 e1 = dp0, e2 = dp1;
                const v3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec), inv = rcp_ieee(det);
                const v3 tvec = o - P[0];
                const v3 qvec = cross3(tvec, e1);
                const float a_u = dot3(tvec, pvec), a_v = dot3(d, qvec), a_t = dot3(e2, qvec);
                const float g_au = gu * inv, g_av = gv * inv, g_at = gt * inv;
                const float g_inv = gu * a_u + gv * a_v + gt * a_t;
                const float g_det = -g_inv * inv * inv;
                v3 ge1 = z3, ge2 = z3, gq = z3, gtv = z3, gpv = z3;
                axpy3(g_at, qvec, ge2); axpy3(g_at, e2, gq);
                axpy3(g_av, qvec, gd);  axpy3(g_av, d, gq);
                axpy3(1.f, cross3(e1, gq), gtv);
                axpy3(1.f, cross3(gq, tvec), ge1);
                axpy3(g_au, pvec, gtv); axpy3(g_au, tvec, gpv);
                axpy3(g_det, pvec, ge1); axpy3(g_det, e1, gpv);
                axpy3(1.f, cross3(e2, gpv), gd);
                axpy3(1.f, cross3(gpv, d), ge2);
                axpy3(1.f, gtv, go); axpy3(-1.f, gtv, gP0);
                axpy3(1.f, ge1, gP1); axpy3(-1.f, ge1, gP0);
                axpy3(1.f, ge2, gP2); axpy3(-1.f, ge2, gP0);
            }
            axpy3(1.f, gdp0, gP1); axpy3(-1.f, gdp0, gP0);
            axpy3(1.f, gdp1, gP2); axpy3(-1.f, gdp1, gP0);

            if (!detach && grad_h) { // dP_k/dh_k = s * (third column of to_world)
                const v3 ez = mk3(f.to_world[2], f.to_world[6], f.to_world[10]);
                atomicAdd(grad_h + (size_t) vi[0] * f.W + vj[0], f.s * dot3(ez, gP0));
                atomicAdd(grad_h + (size_t) vi[1] * f.W + vj[1], f.s * dot3(ez, gP1));
                atomicAdd(grad_h + (size_t) vi[2] * f.W + vj[2], f.s * dot3(ez, gP2));
            }
        }
        if (go0) { go0[i] = go.x; go1[i] = go.y; go2[i] = go.z; }
        if (gd0) { gd0[i] = gd.x; gd1[i] = gd.y; gd2[i] = gd.z; }
    }
}

void hf_launch_adjoint(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                       const uint8_t *active, const hf_si_grad_t *gs, uint32_t flags, float *grad_h,
                       float *const grad_o[3], float *const grad_d[3], hipStream_t stream) {
    if (n == 0) return;
    const hf_pi_cdev p = { pi->t, pi->prim_uv[0], pi->prim_uv[1], pi->prim_index };
    hf_grad_dev g;
    g.t = gs->t;
    for (int k = 0; k < 3; ++k) {
        g.p[k] = gs->p[k]; g.n[k] = gs->n[k]; g.sh_n[k] = gs->sh_n[k];
        g.dp_du[k] = gs->dp_du[k]; g.dp_dv[k] = gs->dp_dv[k];
    }
    g.uv[0] = gs->uv[0]; g.uv[1] = gs->uv[1];
    hipLaunchKernelGGL(hf_adjoint_kernel, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, f, n, to_dev(rays), p, active,
                       g, flags, grad_h, grad_o ? grad_o[0] : nullptr, grad_o ? grad_o[1] : nullptr,
                       grad_o ? grad_o[2] : nullptr, grad_d ? grad_d[0] : nullptr, grad_d ? grad_d[1] : nullptr,
                       grad_d ? grad_d[2] : nullptr);
}
