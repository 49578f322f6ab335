// hf_kernels.hip -- gfx950 kernels of libhf + their launchers.
//
//   hf_mip_level1_kernel / hf_mip_reduce_kernel   min/max mip pyramid
//   hf_shear_kernel / hf_shear_minmax_kernel       node records of every level 1..top (hf_device.h)
//   hf_trace_kernel<MODE>                          hierarchical min/max-mip traversal:
//        MODE 0 closest hit  -> PreliminaryIntersection   (row a1)
//        MODE 1 any hit      -> ray_test                   (row a2)
//        MODE 2 closest hit + fused surface interaction    (rows a1+a4)
//   hf_si_kernel                                   compute_surface_interaction (row a4)
//   hf_adjoint_kernel                              reverse mode of a4, atomic scatter (row a5)
//   hf_direct_kernel / hf_direct_adjoint_kernel, hf_adam_kernel     next rows
//
// Traversal = a walk of the implicit quadtree over the cells (the grid is mirrored so the ray direction is
// non-negative on both axes: "order space").  One visit of an inner node fetches its record -- a plane through
// the node's corner heights and, per child, the range of (z - plane) over the child: "sheared
// bounds", tight on slopes; the zero plane with plain min/max above level HF_SHEAR_TOP -- and keeps
// the children the *fat* ray segment [0,t_hi] overlaps; the children of a level-1 node are
// cells, whose two triangles are then tested.
// The 64 rays of a coherent wave (primary rays: one pixel's samples) share the upper levels: the wave's rays are
// bounded by a beam, the nodes of level HF_SUBTREE_LEVEL along the beam are enumerated front to back with the LANES
// acting as node testers (one round trip for all boxes and records, parked in the wave's LDS), and the survivors are
// box-tested per lane and their own record evaluated for all lanes at once (walk_beam).  Everything below such a node is
// a WAVE-WIDE WORK LIST (items_run, round 4): the children a lane is to visit become items (ray, node) on a stack in
// LDS -- ballot-free prefix-sum compaction -- and every round the wave pops up to 64 of them, any lane taking any
// item (the ray's constants come from the owning lane's registers by ds_bpermute), down to (ray, cell) items whose hits
// are merged per ray with a 64-bit minimum in LDS.  An incoherent wave hands the root to every lane instead: per-lane
// depth-first walk, children front to back, pending children as 4-bit-per-level mask stacks in registers, converged
// rounds (lanes walk until they hold candidate cells; the two-triangle test runs for all of them together).
// Fetches of 256 rays that miss the bound as a whole never get that far (the wide path of the kernel).  The visited
// set is a conservative superset of the cells the ray can hit; the per-triangle test and the tie rule are order
// independent, so the result equals the brute force's.
#include <stdlib.h>
#include "hf_device.h"
#include "hf_launch.h"

#define HF_BLOCK 256
#ifndef HF_SUBTREE_LEVEL
#define HF_SUBTREE_LEVEL 4 // the beam sweep hands nodes of this level (16x16 cells) to the per-lane walk (5: +2 %, 3: the beam is too often wider than four nodes)
#endif

// ---------------------------------------------------------------------------------
// min/max mip pyramid (coarse-first, padded -- see hf_dev_field)
// ---------------------------------------------------------------------------------
// level 1: node (ix,iy) bounds cells [2ix, 2ix+1] x [2iy, 2iy+1], i.e. vertices [2ix, 2ix+2]^2
__global__ __launch_bounds__(HF_BLOCK) void hf_mip_level1_kernel(const float *__restrict__ h, int W, int H, float s,
                                                                float2 *__restrict__ out, int sh) {
    const int idx = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (idx >= (1 << (2 * sh))) return;
    const int iy = idx >> sh, ix = idx & ((1 << sh) - 1);
    float mn = __builtin_inff(), mx = -__builtin_inff();
    // cells [2ix, 2ix+1] x [2iy, 2iy+1] that exist, then their vertices
    const int ci0 = 2 * iy, ci1 = min(2 * iy + 1, H - 2);
    const int cj0 = 2 * ix, cj1 = min(2 * ix + 1, W - 2);
    if (ci0 <= ci1 && cj0 <= cj1) {
        for (int i = ci0; i <= ci1 + 1; ++i)
            for (int j = cj0; j <= cj1 + 1; ++j) {
                const float z = h[(size_t) i * W + j] * s;
                mn = fminf(mn, z); mx = fmaxf(mx, z);
            }
    }
    out[idx] = make_float2(mn, mx);
}

// depth k from depth k+1: 2x2 reduce
__global__ __launch_bounds__(HF_BLOCK) void hf_mip_reduce_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                                int sh) {
    const int idx = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (idx >= (1 << (2 * sh))) return;
    const int iy = idx >> sh, ix = idx & ((1 << sh) - 1);
    const float2 *c = in + ((size_t) (2 * iy) << (sh + 1)) + 2 * ix;
    const float2 a = c[0], b = c[1], d = c[(size_t) 1 << (sh + 1)], e = c[((size_t) 1 << (sh + 1)) + 1];
    out[idx] = make_float2(fminf(fminf(a.x, b.x), fminf(d.x, e.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y)));
}

// sheared bounds of level L (see hf_device.h): one thread per (node, child), then one per node
__global__ __launch_bounds__(HF_BLOCK) void hf_shear_kernel(const float *__restrict__ h, int W, int H, float s, int L,
                                                           int sh, float4 *__restrict__ out) {
    const int idx = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (idx >= (4 << (2 * sh))) return;
    const int j = idx & 3, node = idx >> 2, iy = node >> sh, ix = node & ((1 << sh) - 1);
    const int size = 1 << L, S = size >> 1, x0 = ix * size, y0 = iy * size;
    // plane through the (clamped) corner heights; any plane is valid, the ranges below are exact for it
    const int xa = min(x0, W - 1), xb = min(x0 + size, W - 1), ya = min(y0, H - 1), yb = min(y0 + size, H - 1);
    const float z00 = h[(size_t) ya * W + xa] * s, z10 = h[(size_t) ya * W + xb] * s;
    const float z01 = h[(size_t) yb * W + xa] * s, z11 = h[(size_t) yb * W + xb] * s;
    const float inv = 0.5f / (float) size;
    const float a = ((z10 - z00) + (z11 - z01)) * inv, b = ((z01 - z00) + (z11 - z10)) * inv;
    const float c = 0.25f * ((z00 + z10) + (z01 + z11));
    const float xc = (float) (x0 + S), yc = (float) (y0 + S);
    // child j: cells [cj0, cj1] x [ci0, ci1] that exist, then their vertices
    const int cj0 = x0 + (j & 1) * S, cj1 = min(cj0 + S - 1, W - 2);
    const int ci0 = y0 + (j >> 1) * S, ci1 = min(ci0 + S - 1, H - 2);
    float lo = __builtin_inff(), hi = -__builtin_inff();
    if (ci0 <= ci1 && cj0 <= cj1) {
        for (int i = ci0; i <= ci1 + 1; ++i) {
            const float row = __builtin_fmaf(b, (float) i - yc, c);
            for (int jj = cj0; jj <= cj1 + 1; ++jj) {
                const float w = h[(size_t) i * W + jj] * s - __builtin_fmaf(a, (float) jj - xc, row);
                lo = fminf(lo, w); hi = fmaxf(hi, w);
            }
        }
        // rounding of the plane evaluation above
        const float eps = 1e-6f * (__builtin_fabsf(c) + (__builtin_fabsf(a) + __builtin_fabsf(b)) * (float) size);
        lo -= eps; hi += eps;
    }
    // fourth entry of the record (see below): slope of the plane + the largest child range, over the node's four
    // threads (adjacent lanes; an absent child gives -inf)
    float rmax = hi - lo;
    rmax = fmaxf(rmax, __shfl_xor(rmax, 1));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 2));
    float *rec = (float *) (out + (size_t) node * 3);
    if (j == 0) { rec[0] = a; rec[1] = b; rec[2] = c; rec[3] = __builtin_fabsf(a) + __builtin_fabsf(b) + 2.f * fmaxf(rmax, 0.f); }
    rec[4 + 2 * j] = lo; rec[5 + 2 * j] = hi;
}
// Fourth entry of the record: what the walk multiplies its xy uncertainty m by to get a z uncertainty -- the slope
// of the plane, |a|+|b| per cell, PLUS TWICE the largest sheared range of a child: the triangle test may report a hit
// up to m cells beside the walk's ray (that is what m stands for), and where the surface is far steeper than the plane
// (a needle triangle) those m cells are up to m x (|dw/dx| + |dw/dy|) up or down, each partial derivative of a triangle
// bounded by the sheared range of its cell, hence of the child that holds it.  (One range until round 4: the full brute
// force over 16.7 M cells -- found through the oracle walk once IT carried the factor two -- reports a noise hit at
// N = 4096 from 8 units away that (|a|+|b|+r) m fell short of: tests/test_gpu_band.py::
// test_walk_needle_term_regression_on_the_gpu.)
// (computed by hf_shear_kernel / hf_shear_level1_kernel with the ranges.)

// Level 1 (three quarters of all records, 201 MB at N = 4096): one thread per node -- the 3x3 vertex window is read
// once, the four children are the cells themselves, the slope factor is computed in the same pass and the record is
// written as three 16-byte stores.  Same arithmetic as hf_shear_kernel with L = 1.
// The same window also gives the node's entry of the min/max pyramid (hf_mip_level1_kernel's value: clamping only
// repeats vertices of existing cells), written when mip1 is not null.
__global__ __launch_bounds__(HF_BLOCK) void hf_shear_level1_kernel(const float *__restrict__ h, int W, int H, float s,
                                                                  int sh, float4 *__restrict__ out,
                                                                  float2 *__restrict__ mip1) {
    const int node = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (node >= (1 << (2 * sh))) return;
    const int iy = node >> sh, ix = node & ((1 << sh) - 1);
    const int x0 = 2 * ix, y0 = 2 * iy;
    float z[3][3]; // vertex window, clamped to the grid (clamped entries are only used where the kernel above clamps too)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            z[i][j] = h[(size_t) min(y0 + i, H - 1) * W + min(x0 + j, W - 1)] * s;
    if (mip1) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        if (y0 <= H - 2 && x0 <= W - 2) { // at least cell (x0, y0) exists
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { mn = fminf(mn, z[i][j]); mx = fmaxf(mx, z[i][j]); }
        }
        mip1[node] = make_float2(mn, mx);
    }
    const float z00 = z[0][0], z10 = z[0][2], z01 = z[2][0], z11 = z[2][2];
    const float inv = 0.5f / 2.f;
    const float a = ((z10 - z00) + (z11 - z01)) * inv, b = ((z01 - z00) + (z11 - z10)) * inv;
    const float c = 0.25f * ((z00 + z10) + (z01 + z11));
    const float xc = (float) (x0 + 1), yc = (float) (y0 + 1);
    const float eps = 1e-6f * (__builtin_fabsf(c) + (__builtin_fabsf(a) + __builtin_fabsf(b)) * 2.f);
    float lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cj = x0 + (j & 1), ci = y0 + (j >> 1); // the child is cell (cj, ci)
        lo[j] = __builtin_inff(); hi[j] = -__builtin_inff();
        if (cj <= W - 2 && ci <= H - 2) {
#pragma unroll
            for (int di = 0; di < 2; ++di) {
                const float row = __builtin_fmaf(b, (float) (ci + di) - yc, c);
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const float w = z[(j >> 1) + di][(j & 1) + dj] - __builtin_fmaf(a, (float) (cj + dj) - xc, row);
                    lo[j] = fminf(lo[j], w); hi[j] = fmaxf(hi[j], w);
                }
            }
            lo[j] -= eps; hi[j] += eps;
        }
    }
    const float rmax = fmaxf(fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), fmaxf(hi[2] - lo[2], hi[3] - lo[3])), 0.f); // absent: -inf
    float4 *rec = out + (size_t) node * 3;
    rec[0] = make_float4(a, b, c, __builtin_fabsf(a) + __builtin_fabsf(b) + 2.f * rmax);
    rec[1] = make_float4(lo[0], hi[0], lo[1], hi[1]);
    rec[2] = make_float4(lo[2], hi[2], lo[3], hi[3]);
}

// Levels above HF_SHEAR_TOP keep plain min/max boxes, stored in the same record form with the zero plane
// (a = b = c = 0, slope factor 0: w = z, and shear_line leaves the ray's z line untouched bit for bit), so that
// the per-lane walk reads every inner node through ONE code path -- three 16-byte loads from one address.
// Children of node (ix,iy) of level L = nodes (2ix+jx, 2iy+jy) of depth k+1 of the pyramid (L >= 2).
// (a, b, c) = 0 and |a|+|b|+r = the largest child range r: the record of a min/max level.  r is the needle term of
// shear_line's z margin -- a hit reported up to m cells beside the walk's ray sits up to m x (height range) above or
// below it -- which the fitted-plane records carry too; absent children (+inf, -inf) do not count.
__device__ __forceinline__ float4 hf_minmax_plane(float2 a, float2 b, float2 d, float2 e) {
    const float r = fmaxf(fmaxf(fmaxf(a.y - a.x, b.y - b.x), fmaxf(d.y - d.x, e.y - e.x)), 0.f);
    return make_float4(0.f, 0.f, 0.f, 2.f * r);
}
__global__ __launch_bounds__(HF_BLOCK) void hf_shear_minmax_kernel(const float2 *__restrict__ child, int sh,
                                                                  float4 *__restrict__ out) {
    const int node = blockIdx.x * HF_BLOCK + threadIdx.x;
    if (node >= (1 << (2 * sh))) return;
    const int iy = node >> sh, ix = node & ((1 << sh) - 1);
    const float2 *c = child + ((size_t) (2 * iy) << (sh + 1)) + 2 * ix;
    const float2 a = c[0], b = c[1], d = c[(size_t) 1 << (sh + 1)], e = c[((size_t) 1 << (sh + 1)) + 1];
    float4 *rec = out + (size_t) node * 3;
    rec[0] = hf_minmax_plane(a, b, d, e);
    rec[1] = make_float4(a.x, a.y, b.x, b.y);
    rec[2] = make_float4(d.x, d.y, e.x, e.y);
}

// The top of the pyramid in ONE launch: depths kmax..0 (at most 64x64 nodes each) by a single workgroup, level after
// level -- the 2x2 reduce of hf_mip_reduce_kernel and, above HF_SHEAR_TOP, the record of hf_shear_minmax_kernel, both
// from depth k+1, which the previous iteration (or the previous launch, for kmax) has completed.
#define HF_MIP_TOP_DEPTH 6
__global__ __launch_bounds__(1024) void hf_mip_top_kernel(float2 *__restrict__ mip, float4 *__restrict__ shear, int top,
                                                         int kmax) {
    for (int k = kmax; k >= 0; --k) {
        const float2 *c0 = mip + hf_depth_off(k + 1);
        float2 *o = mip + hf_depth_off(k);
        const bool rec_level = (top - k) > HF_SHEAR_TOP;
        float4 *recs = shear + (size_t) (hf_depth_off(k) - 1u) * 3;
        for (int node = (int) threadIdx.x; node < (1 << (2 * k)); node += (int) blockDim.x) {
            const int iy = node >> k, ix = node & ((1 << k) - 1);
            const float2 *c = c0 + ((size_t) (2 * iy) << (k + 1)) + 2 * ix;
            const float2 a = c[0], b = c[1], d = c[(size_t) 1 << (k + 1)], e = c[((size_t) 1 << (k + 1)) + 1];
            o[node] = make_float2(fminf(fminf(a.x, b.x), fminf(d.x, e.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y)));
            if (rec_level) {
                float4 *rec = recs + (size_t) node * 3;
                rec[0] = hf_minmax_plane(a, b, d, e);
                rec[1] = make_float4(a.x, a.y, b.x, b.y);
                rec[2] = make_float4(d.x, d.y, e.x, e.y);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

void hf_launch_build_mips(const hf_dev_field &f, float2 *mip, float4 *shear, hipStream_t stream) {
    const int top = f.top;
    for (int L = 1; L <= top && L <= HF_SHEAR_TOP; ++L) { // level 1: the children are the cells themselves
        const int k = top - L, n = 1 << (2 * k);
        float4 *recs = shear + (size_t) (hf_depth_off(k) - 1u) * 3;
        if (L == 1) {
            hipLaunchKernelGGL(hf_shear_level1_kernel, dim3((n + HF_BLOCK - 1) / HF_BLOCK), dim3(HF_BLOCK), 0, stream, f.h,
                               f.W, f.H, f.s, k, recs, mip + hf_depth_off(k)); // + depth top-1 of the pyramid
            continue;
        }
        hipLaunchKernelGGL(hf_shear_kernel, dim3((4 * n + HF_BLOCK - 1) / HF_BLOCK), dim3(HF_BLOCK), 0, stream, f.h, f.W,
                           f.H, f.s, L, k, recs);
    }
    (void) hipMemsetAsync(mip, 0, sizeof(float2), stream); // padding entry
    // depths <= ktop of the pyramid and their records come from one launch at the end (hf_mip_top_kernel); every level
    // it covers must be a plain min/max level (the fitted-plane levels 1..HF_SHEAR_TOP are built from the heights)
    const int ktop = max(min(HF_MIP_TOP_DEPTH, top - 1 - HF_SHEAR_TOP), -1); // -1: no such level (top <= HF_SHEAR_TOP)
    for (int k = top - 1; k > ktop; --k) {
        const int n = 1 << (2 * k), grid = (n + HF_BLOCK - 1) / HF_BLOCK;
        if (k == top - 1) {
            if (top >= 1 && HF_SHEAR_TOP >= 1) continue; // written by hf_shear_level1_kernel above
            hipLaunchKernelGGL(hf_mip_level1_kernel, dim3(grid), dim3(HF_BLOCK), 0, stream, f.h, f.W, f.H, f.s,
                               mip + hf_depth_off(k), k);
        } else
            hipLaunchKernelGGL(hf_mip_reduce_kernel, dim3(grid), dim3(HF_BLOCK), 0, stream,
                               (const float2 *) (mip + hf_depth_off(k + 1)), mip + hf_depth_off(k), k);
    }
    for (int L = HF_SHEAR_TOP + 1; L <= top; ++L) { // after the pyramid: reads depth k+1 of it
        const int k = top - L, n = 1 << (2 * k);
        if (k <= ktop) break; // the rest: hf_mip_top_kernel
        hipLaunchKernelGGL(hf_shear_minmax_kernel, dim3((n + HF_BLOCK - 1) / HF_BLOCK), dim3(HF_BLOCK), 0, stream,
                           (const float2 *) (mip + hf_depth_off(k + 1)), k, shear + (size_t) (hf_depth_off(k) - 1u) * 3);
    }
    if (ktop >= 0)
        hipLaunchKernelGGL(hf_mip_top_kernel, dim3(1), dim3(1024), 0, stream, mip, shear, top, ktop);
}

// ---------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------
#ifdef HF_WSTATS
// wave-level execution counters: WCOUNT(k) adds 1 per wave each time the enclosing code runs with any lane
__device__ __forceinline__ uint32_t *wcnt_base() {
    __shared__ uint32_t c[HF_BLOCK / 64][16];
    return c[threadIdx.x >> 6];
}
#define WCOUNT(k) do { const uint64_t e_ = __ballot(true); if ((int) (threadIdx.x & 63u) == __builtin_ctzll(e_)) wcnt_base()[k]++; } while (0)
// WLANES(k): adds the number of lanes that run the enclosing code (k = 7: low half word = visits, high = cell rounds)
#define WLANES(k, sh) do { const uint64_t e_ = __ballot(true); if ((int) (threadIdx.x & 63u) == __builtin_ctzll(e_)) wcnt_base()[k] += (uint32_t) __builtin_popcountll(e_) << (sh); } while (0)
__device__ __forceinline__ void wstats_reset() { if ((threadIdx.x & 63u) < 16u) wcnt_base()[threadIdx.x & 63u] = 0u; }
// diagnostic build (scripts/wstats.py): the counters of this batch replace the hit record
#if HF_WSTATS == 4
// ... variant 4: histogram of the lanes that run a visit (c[11..13]) / a cell round (c[8..10]): at most 8, 9..24, more
#define WHIST(base) do { const uint64_t e_ = __ballot(true); if ((int) (threadIdx.x & 63u) == __builtin_ctzll(e_)) { \
        const int n_ = __builtin_popcountll(e_); wcnt_base()[(base) + (n_ <= 8 ? 0 : n_ <= 24 ? 1 : 2)]++; } } while (0)
#define WSTATS_EXPORT(alive, best) do { if (alive) { const uint32_t *c = wcnt_base(); (best).hit = true; \
        (best).t = (float) c[8] + 1024.f * (float) c[9] + 1048576.f * (float) c[10]; \
        (best).u = (float) c[11] + 1024.f * (float) c[12] + 1048576.f * (float) c[13]; (best).v = 0.f; (best).prim = c[7]; } } while (0)
#else
#define WSTATS_EXPORT(alive, best) do { if (alive) { const uint32_t *c = wcnt_base(); (best).hit = true; \
        (best).t = (float) c[0] + 1024.f * (float) c[1] + 1048576.f * (float) c[2]; \
        (best).u = (float) c[3] + 4096.f * (float) c[4]; (best).v = (float) c[5] + 4096.f * (float) c[6]; (best).prim = c[7]; } } while (0)
#endif
#else
#define WLANES(k, sh) do { } while (0)
#define WSTATS_EXPORT(alive, best) do { } while (0)
#define WCOUNT(k) do { } while (0)
__device__ __forceinline__ void wstats_reset() { }
#endif

#ifdef HF_TSTATS
// Cycle stamps (diagnostic build, scripts/tstats.py): TSTAMP(k) adds the shader cycles since the previous stamp of the
// wave to phase k of the batch (lane 0 keeps the books in LDS; each stamp costs an s_memtime and three LDS operations);
// at the end of a batch with live lanes, lane l < 16 reports phase l in place of its hit record.
__device__ __forceinline__ uint32_t *tcnt_base() {
    __shared__ uint32_t c[HF_BLOCK / 64][16];
    return c[threadIdx.x >> 6];
}
#define TSTART() do { __builtin_amdgcn_wave_barrier(); if ((threadIdx.x & 63u) == 0u) { uint32_t *c_ = tcnt_base(); \
        const uint32_t now_ = (uint32_t) clock64(), out_ = now_ - c_[15]; /* since the last stamp of the wave's previous batch: its output phase */ \
        for (int i_ = 0; i_ < 15; ++i_) c_[i_] = 0u; c_[8] = out_; c_[15] = now_; } __builtin_amdgcn_wave_barrier(); } while (0)
#define TSTAMP(k) do { if ((threadIdx.x & 63u) == 0u) { uint32_t *c_ = tcnt_base(); const uint32_t now_ = (uint32_t) clock64(); \
        c_[k] += now_ - c_[15]; c_[15] = now_; } } while (0)
#define TSTATS_EXPORT(any_alive, valid, best) do { __builtin_amdgcn_wave_barrier(); if ((any_alive) && (valid)) { (best).hit = true; \
        (best).t = (float) tcnt_base()[threadIdx.x & 15u]; (best).u = 0.f; (best).v = 0.f; (best).prim = threadIdx.x & 63u; } } while (0)
#else
#define TSTART() do { } while (0)
#define TSTAMP(k) do { } while (0)
#endif
#ifndef HF_M0
#define HF_M0 0.00390625f // constant part of the xy margin, cells (1/64 until round 3: see setup_ray)
#endif
#ifndef HF_LINE_EPS
#define HF_LINE_EPS 1e-6f // rounding of the sheared line, per cell of the grid's side (2e-6 until round 3)
#endif
#ifndef WHIST
#define WHIST(base) do { } while (0)
#endif
// per-ray traversal constants (order space, cell units, re-based at t = tin)
struct hf_trav {
    float gxm, gxp, gym, gyp; // origin x,y  +/- the xy margin m
    float gz, dz, idx, idy, mz;
};

// four cells as (min,max) boxes in ACTUAL order j = 2*jy + jx
struct hf_quad {
    float lo[4], hi[4];
};

// Overlap mask (ACTUAL numbering j = 2*jy + jx) of the fat ray segment [0,thi] with the four
// boxes of the 2x2 block whose order-space origin is (fX,fY), box size S.
// parameter of box j.  The third coordinate of the ray is the line  gz + t dz  (+/- mz): the height
// for min/max boxes, the sheared height for sheared bounds.  Direction is >= 0 in order space, so a box's entry planes are its low
// faces and its exit planes its high faces; v_max3/v_min3 drop the NaN of 0*inf (origin of an
// axis-parallel ray exactly on a face plane).
__device__ __forceinline__ uint32_t child_mask(const hf_trav &r, bool fx, bool fy, float fX, float fY, float S,
                                               const hf_quad &q, float gz, float dz, float mz, float thi) {
    const float ex = fX - r.gxm, lx = fX + S - r.gxp; // entry / exit plane offsets of order column 0
    const float ey = fY - r.gym, ly = fY + S - r.gyp;
    // order-space offset of ACTUAL child column / row 0 and 1 (a mirrored axis swaps near and far half)
    const float sx0 = fx ? S : 0.f, sx1 = S - sx0, sy0 = fy ? S : 0.f, sy1 = S - sy0;
    const float x0lo = (ex + sx0) * r.idx, x0hi = (lx + sx0) * r.idx;
    const float x1lo = (ex + sx1) * r.idx, x1hi = (lx + sx1) * r.idx;
    const float y0lo = (ey + sy0) * r.idy, y0hi = (ly + sy0) * r.idy;
    const float y1lo = (ey + sy1) * r.idy, y1hi = (ly + sy1) * r.idy;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xlo = (j & 1) ? x1lo : x0lo, xhi = (j & 1) ? x1hi : x0hi;
        const float ylo = (j & 2) ? y1lo : y0lo, yhi = (j & 2) ? y1hi : y0hi;
        const float t0 = fmaxf(fmaxf(xlo, ylo), 0.f), t1 = fminf(fminf(xhi, yhi), thi);
        const float za = __builtin_fmaf(t0, dz, gz), zb = __builtin_fmaf(t1, dz, gz);
        const bool ok = (t0 <= t1) & (fminf(za, zb) - mz <= q.hi[j]) & (fmaxf(za, zb) + mz >= q.lo[j]);
        m |= ok ? (1u << j) : 0u;
    }
    return m;
}

// everything a lane needs to walk its ray: object-space ray (spec arithmetic input of the
// triangle test) + traversal constants in order space
struct hf_ray_state {
    v3 oo, od;
    float maxt, tin, thi;
    float gx, gy, m; // order-space entry point (cell units) and xy margin
    hf_trav r;
    bool fx, fy;
};

// transform to object space, clip against the inflated bound, build the traversal ray.
// returns false when the ray cannot hit (outside the bound, non-finite, bad maxt).
__device__ __forceinline__ bool setup_ray(const hf_dev_field &f, float2 zr, v3 o, v3 d, float maxt,
                                          hf_ray_state &rs) {
    hf_trav &r = rs.r;
    const v3 oo = xform_point(f.to_object, o), od = xform_vec(f.to_object, d);
    rs.oo = oo; rs.od = od; rs.maxt = maxt;
    // non-finite input or NaN/negative maxt: miss (also bounds the walk below)
    {
        const float chk = (oo.x + oo.y + oo.z) + (od.x + od.y + od.z);
        if (!(__builtin_fabsf(chk) < __builtin_inff()) || !(maxt >= 0.f)) return false;
    }
    const int cw = f.W - 1, ch = f.H - 1, top = f.top;
    const float hx = f.hx, hy = f.hy; // 0.5 (W - 1), 0.5 (H - 1)
    const float zspan = fmaxf(zr.y - zr.x, fmaxf(__builtin_fabsf(zr.x), __builtin_fabsf(zr.y)));
    const float mz0 = 1e-5f * zspan + 1e-30f;

    // The margins first: they depend on how far the ray travels to the grid (`reach`), measured with the entry into
    // the bound inflated by the fixed amounts only (1e-4 in xy, mz0 in z) -- also for a ray that misses that bound.
    const float oc[3] = { oo.x, oo.y, oo.z }, dc[3] = { od.x, od.y, od.z };
    float rr[3];
    float tin0 = 0.f;
    {
        const float lo[3] = { -1.f - 1e-4f, -1.f - 1e-4f, zr.x - mz0 };
        const float hi[3] = { 1.f + 1e-4f, 1.f + 1e-4f, zr.y + mz0 };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rr[k] = 1.0f / dc[k];
            if (dc[k] != 0.f) tin0 = fmaxf(tin0, fminf((lo[k] - oc[k]) * rr[k], (hi[k] - oc[k]) * rr[k]));
        }
        tin0 = tin0 - __builtin_fabsf(tin0) * 1e-6f;
        tin0 = fmaxf(tin0, 0.f);
    }
    const float reach = __builtin_fabsf(oo.x) + __builtin_fabsf(oo.y) +
                        tin0 * (__builtin_fabsf(od.x) + __builtin_fabsf(od.y)) + 2.f;
    // xy margin: covers the rounding of the walk and the NOISE OF THE TRIANGLE TEST ITSELF -- how far beside the exact
    // ray a triangle can lie and still be reported hit by the fp32 Moeller-Trumbore arithmetic -- which grows faster
    // than linearly with the distance of the origin (against float64 geometry and the brute force over all cells:
    // 0.001 cell from 3 units away, up to 1.7 cells from 50 units away on needle terrain at N = 4096).  Beyond a reach
    // of 8 units the distance term therefore grows with (reach / 8)^2 (round 3: with the linear term and a cap of 0.3
    // cell the walk missed ~1 such hit in 10^4 rays traced from 50 units away; the walk only gets slower with m);
    // up to a reach of 8 units -- most rays of the BASELINE configurations; a few reach 9.5 -- the distance term is what it was.  Capped at 8 cells so that the
    // strip a ray walks (and the time of the launch) stays bounded however far its origin.  The constant part HF_M0 is
    // pure slack on top of the distance term, which is never below 8 eps x (grid side) -- the rounding of the walk's own
    // slab arithmetic; it was 1/64 cell until round 3 and fattened every sheared slab by 3x its curvature thickness on
    // smooth terrain: at 1/256 a batch needs 4.7 instead of 5.7 cell rounds (forward -3 %, bounce rays -4 %).  Same
    // formula in the oracle's walk; the band brute force (no margins at all) and the fuzz check that it suffices.
    const float far = fmaxf(1.f, 0.125f * reach);
    const float m = HF_M0 + fminf(8.f, 4.8e-7f * reach * fmaxf(hx, hy) * (far * far));

    // clip against the object-space bound (slab test, bbox.h:302-327) inflated by what such a hit can reach: m cells in
    // xy and m x (height span) in z on top of the fixed amounts -- every node test of the walk has that needle term,
    // and without it here a ray traced from 40 units away lost a (noise) hit the brute force reports 1 % of the height
    // span above the bound (round 3, the one mismatch of 4e9 fuzz rays against the band brute force)
    float tin = 0.f, tout = maxt;
    {
        // (sx = 1 / hx: a cell.  Multiply + add, not fma and not m / hx: the two divisions cost the launch 2 %, and with
        // the fused form the fused kernel spills 6 registers instead of 2)
        const float ex = 1e-4f + m * f.sx, ey = 1e-4f + m * f.sy, ez = __builtin_fmaf(2.f * m, zspan, mz0);
        const float lo[3] = { -1.f - ex, -1.f - ey, zr.x - ez };
        const float hi[3] = { 1.f + ex, 1.f + ey, zr.y + ez };
        bool outside = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (dc[k] == 0.f) {
                outside |= (oc[k] < lo[k]) | (oc[k] > hi[k]);
            } else {
                const float t1 = (lo[k] - oc[k]) * rr[k], t2 = (hi[k] - oc[k]) * rr[k];
                tin = fmaxf(tin, fminf(t1, t2));
                tout = fminf(tout, fmaxf(t1, t2));
            }
        }
        tin = tin - __builtin_fabsf(tin) * 1e-6f;
        tin = fmaxf(tin, 0.f);
        tout = tout + __builtin_fabsf(tout) * 1e-6f;
        if (outside || !(tin <= tout)) return false;
    }

    // traversal ray in cell units, re-based at t = tin, mirrored into order space
    const bool fx = od.x < 0.f, fy = od.y < 0.f;
    const float Wp = (float) (1 << top);
    // entry point in double: keeps the walk accurate for origins far from the grid
    float gx = (float) (((double) oo.x + (double) tin * (double) od.x + 1.0) * (double) hx);
    float gy = (float) (((double) oo.y + (double) tin * (double) od.y + 1.0) * (double) hy);
    r.gz = __builtin_fmaf(tin, od.z, oo.z);
    float dx = od.x * hx, dy = od.y * hy;
    r.dz = od.z;
    if (fx) { gx = Wp - gx; dx = -dx; }
    if (fy) { gy = Wp - gy; dy = -dy; }
    // |.|: a negative-zero component (d = -up, or a -0 surviving to_object) is mirrored by neither test above and
    // would give -inf here, which flips every slab test below; canonicalised it is the +inf of an axis-parallel ray
    r.idx = 1.0f / __builtin_fabsf(dx); r.idy = 1.0f / __builtin_fabsf(dy);
    r.mz = mz0 + 4.8e-7f * (__builtin_fabsf(oo.z) + tin * __builtin_fabsf(od.z) + zspan);
    r.gxm = gx + m; r.gxp = gx - m; r.gym = gy + m; r.gyp = gy - m;
    float thi = tout - tin;
    thi = thi + thi * 1e-6f + 1e-30f;

    rs.tin = tin; rs.thi = thi; rs.gx = gx; rs.gy = gy; rs.m = m; rs.fx = fx; rs.fy = fy;
    return true;
}

// Conservative twin of setup_ray's clip: false ONLY IF setup_ray would return false for this ray (it may say true for a
// ray setup_ray rejects: that ray just takes the ordinary path).  Same formulas with v_rcp_f32 instead of the three IEEE
// divisions, every quantity that enters the decision pushed to the safe side by 1e-5 relative (the reciprocals are off
// by one ulp, the products by half an ulp each: 3e-7), a ray with a zero direction component or a non-finite one is
// "maybe".  Used by the wide path of the traversal kernels for fetches that miss the bound as a whole.
__device__ __forceinline__ bool maybe_alive(const hf_dev_field &f, float2 zr, v3 o, v3 d, float maxt) {
    const v3 oo = xform_point(f.to_object, o), od = xform_vec(f.to_object, d);
    const float chk = (oo.x + oo.y + oo.z) + (od.x + od.y + od.z);
    if (!(__builtin_fabsf(chk) < __builtin_inff()) || !(maxt >= 0.f)) return true; // (setup_ray: a miss; left to it)
    if (!(__builtin_fabsf(od.x) >= 1e-30f) || !(__builtin_fabsf(od.y) >= 1e-30f) || !(__builtin_fabsf(od.z) >= 1e-30f)) return true; // (zero / denormal: 1/d overflows)
    const float zspan = fmaxf(zr.y - zr.x, fmaxf(__builtin_fabsf(zr.x), __builtin_fabsf(zr.y)));
    const float mz0 = 1e-5f * zspan + 1e-30f;
    const float oc[3] = { oo.x, oo.y, oo.z }, dc[3] = { od.x, od.y, od.z };
    float rr[3], tin0 = 0.f;
    {
        const float lo[3] = { -1.f - 1e-4f, -1.f - 1e-4f, zr.x - mz0 };
        const float hi[3] = { 1.f + 1e-4f, 1.f + 1e-4f, zr.y + mz0 };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rr[k] = __builtin_amdgcn_rcpf(dc[k]);
            tin0 = fmaxf(tin0, fminf((lo[k] - oc[k]) * rr[k], (hi[k] - oc[k]) * rr[k]));
        }
        tin0 = fmaxf(tin0, 0.f) * (1.f + 1e-5f); // an UPPER bound of setup_ray's tin0
    }
    const float reach = (__builtin_fabsf(oo.x) + __builtin_fabsf(oo.y) + tin0 * (__builtin_fabsf(od.x) + __builtin_fabsf(od.y)) + 2.f) * (1.f + 1e-5f);
    const float far = fmaxf(1.f, 0.125f * reach);
    const float m = (HF_M0 + fminf(8.f, 4.8e-7f * reach * fmaxf(f.hx, f.hy) * (far * far))) * (1.f + 1e-5f); // >= setup_ray's m
    const float ex = 1e-4f + m * f.sx, ey = 1e-4f + m * f.sy, ez = __builtin_fmaf(2.f * m, zspan, mz0);
    const float lo[3] = { -1.f - ex, -1.f - ey, zr.x - ez }, hi[3] = { 1.f + ex, 1.f + ey, zr.y + ez }; // contain setup_ray's box
    float tin = 0.f, tout = maxt;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t1 = (lo[k] - oc[k]) * rr[k], t2 = (hi[k] - oc[k]) * rr[k];
        tin = fmaxf(tin, fminf(t1, t2));
        tout = fminf(tout, fmaxf(t1, t2));
    }
    // setup_ray keeps the ray when tin (1 - 1e-6) <= tout (1 + 1e-6); here both sides carry the reciprocals' error on top
    return tin - __builtin_fabsf(tin) * 1e-5f <= tout + __builtin_fabsf(tout) * 1e-5f;
}

// All auxiliary rays of a ray at once (hf_reparam_trace_all): false ONLY IF no von Mises-Fisher sample around d can
// pass setup_ray's clip.  Every sample lies within the angle theta_max of d (warp.h:557-566 with its clamp of the
// sample at 1e-6: cos theta >= 1 - 13.82 / kappa), so at the parameter tau of the primary ray's object-space line the
// auxiliary point is at most tau x ct away from it, ct = sqrt(2) ||to_object|| tan(theta_max) (launcher); an auxiliary
// hit lies in the bound, hence at tau <= (R + |o - c|) / (|d_obj| - ct) (R: radius of the inflated bound around its
// centre c), so the primary ray has to pass the bound inflated by r_max = that tau x ct on every axis -- tested like
// maybe_alive (v_rcp_f32, decisions pushed to the safe side).  "Maybe" whenever an assumption of the bound fails: a
// direction that is not unit length (the frame is then not orthonormal), an inflation of more than a tenth of the
// bound (tiny grids, far origins), |d_obj| <= ct.
__device__ __forceinline__ bool cone_maybe_alive(const hf_dev_field &f, float2 zr, v3 o, v3 d, float ct) {
    const v3 oo = xform_point(f.to_object, o), od = xform_vec(f.to_object, d);
    const float chk = (oo.x + oo.y + oo.z) + (od.x + od.y + od.z);
    if (!(__builtin_fabsf(chk) < __builtin_inff())) return true;
    if (!(__builtin_fabsf(dot3(d, d) - 1.f) <= 1e-3f)) return true;
    if (!(__builtin_fabsf(od.x) >= 1e-30f) || !(__builtin_fabsf(od.y) >= 1e-30f) || !(__builtin_fabsf(od.z) >= 1e-30f)) return true;
    const float zspan = fmaxf(zr.y - zr.x, fmaxf(__builtin_fabsf(zr.x), __builtin_fabsf(zr.y)));
    const float mz0 = 1e-5f * zspan + 1e-30f;
    const float zc = 0.5f * (zr.x + zr.y), zh = 0.5f * (zr.y - zr.x);
    const float zcap = 0.1f * zh + 0.1f * zspan + 1e-3f; // the z inflation R allows for (x, y: 0.1)
    const float dz0 = oo.z - zc;
    const float dist = __builtin_sqrtf(__builtin_fmaf(dz0, dz0, __builtin_fmaf(oo.y, oo.y, oo.x * oo.x))) * (1.f + 1e-5f);
    const float hzc = zh + zcap;
    const float R = __builtin_sqrtf(__builtin_fmaf(hzc, hzc, 2.42f)) * (1.f + 1e-5f); // half diagonal of (1.1, 1.1, zh + zcap)
    // no auxiliary ray travels further to the bound than this (cf. setup_ray's reach): an upper bound of every sample's m
    const float reach = (__builtin_fabsf(oo.x) + __builtin_fabsf(oo.y) + 1.4143f * (dist + R) + 2.f) * (1.f + 1e-5f);
    const float far = fmaxf(1.f, 0.125f * reach);
    const float m = (HF_M0 + fminf(8.f, 4.8e-7f * reach * fmaxf(f.hx, f.hy) * (far * far))) * (1.f + 1e-5f);
    const float ex = 1e-4f + m * f.sx, ey = 1e-4f + m * f.sy, ez = __builtin_fmaf(2.f * m, zspan, mz0);
    if (!(ex <= 0.1f) || !(ey <= 0.1f) || !(ez <= zcap)) return true;
    const float den = __builtin_sqrtf(dot3(od, od)) * (1.f - 1e-5f) - ct;
    if (!(den > 0.f)) return true;
    const float rmax = (R + dist) * __builtin_amdgcn_rcpf(den) * ct * (1.f + 1e-4f);
    const float lo[3] = { -1.f - ex - rmax, -1.f - ey - rmax, zr.x - ez - rmax }, hi[3] = { 1.f + ex + rmax, 1.f + ey + rmax, zr.y + ez + rmax };
    const float oc[3] = { oo.x, oo.y, oo.z }, dc[3] = { od.x, od.y, od.z };
    float tin = 0.f, tout = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r = __builtin_amdgcn_rcpf(dc[k]);
        const float t1 = (lo[k] - oc[k]) * r, t2 = (hi[k] - oc[k]) * r;
        tin = fmaxf(tin, fminf(t1, t2));
        tout = fminf(tout, fmaxf(t1, t2));
    }
    return tin - __builtin_fabsf(tin) * 1e-5f <= tout + __builtin_fabsf(tout) * 1e-5f;
}

// The ray in the sheared coordinate  w = z - (c + a (x - xc) + b (y - yc))  of a node whose centre is
// (xc,yc) in order space: w(t) = gz + t dz.  (a,b) are stored for actual coordinates; mirroring an
// axis into order space flips the sign of its slope.  mz grows by the slope times the xy uncertainty
// of the walk (the margin m) plus the rounding of the line itself, whose terms are as large as
// slope x grid size.
__device__ __forceinline__ void shear_line_v(const hf_trav &r, float dxo, float dyo, float eps_top, bool fx, bool fy, float a,
                                             float b, float c, float sab, float xc, float yc, float &gz, float &dz, float &mz) {
    const float ao = fx ? -a : a, bo = fy ? -b : b;
    const float ux = 0.5f * (r.gxm + r.gxp) - xc, uy = 0.5f * (r.gym + r.gyp) - yc;
    gz = __builtin_fmaf(-bo, uy, __builtin_fmaf(-ao, ux, r.gz - c));
    dz = __builtin_fmaf(-bo, dyo, __builtin_fmaf(-ao, dxo, r.dz));
    const float m = 0.5f * (r.gxm - r.gxp) + eps_top;
    mz = __builtin_fmaf(sab, m, r.mz);
}
// (dxo, dyo: the xy direction in cells per unit of t, order space; eps_top = HF_LINE_EPS x the grid's side)
__device__ __forceinline__ void shear_line(const hf_dev_field &f, const hf_ray_state &rs, bool fx, bool fy, float a,
                                           float b, float c, float sab, float xc, float yc, float &gz, float &dz,
                                           float &mz) {
    const float dxo = __builtin_fabsf(rs.od.x) * f.hx;
    const float dyo = __builtin_fabsf(rs.od.y) * f.hy;
    shear_line_v(rs.r, dxo, dyo, HF_LINE_EPS * (float) (1 << f.top), fx, fy, a, b, c, sab, xc, yc, gz, dz, mz);
}

// actual-child mask -> order-space child mask (bit k = bit (k ^ flip))
__device__ __forceinline__ uint32_t to_order(uint32_t m, bool fx, bool fy) {
    if (fx) m = ((m & 5u) << 1) | ((m >> 1) & 5u);
    if (fy) m = ((m & 3u) << 2) | ((m >> 2) & 3u);
    return m;
}

// data source of the per-lane subtree walk: heights + mips straight from global memory (L1/L2)
struct hf_src_global {
    const float2 *__restrict__ mip;
    const float4 *__restrict__ shear;
    const float *__restrict__ h;
    int top, W;
    // 32-bit byte offset from the uniform base (one scalar-base load, no 64-bit vector address maths);
    // hf_create limits the grid to 2^30 vertices
    __device__ __forceinline__ float height(int i, int j) const {
        const uint32_t off = ((uint32_t) i * (uint32_t) W + (uint32_t) j) << 2;
        return *(const float *) ((const char *) h + off);
    }
    // record of inner node (ix,iy) of level L >= 2: plane + the four child ranges (hf_device.h)
    __device__ __forceinline__ const float4 *sheared(int L, uint32_t ix, uint32_t iy) const {
        const uint32_t k = (uint32_t) (top - L);
        return shear + (size_t) (hf_depth_off((int) k) - 1u + (iy << k) + ix) * 3;
    }
};

// Per-lane depth-first walk of the subtree rooted at order-space node (X0,Y0) of level L0 >= 1
// (pending-children masks of the levels below L0 in a 4-bit-per-level register stack).
// "while-while": each lane walks until it holds a block with candidate cells (or is done); when
// every lane of the call has stopped, the candidate cells are triangle-tested together, one
// cell per lane per round -- the expensive test runs with all waiting lanes active.
// STK: the register stack of pending-children masks, 4 bits per level -- 64 bits for a walk from the root (the root of
// a 2^15-cell grid is 15 levels up), 32 bits for the walk below a hand-off node (HF_SUBTREE_LEVEL - 1 <= 8 levels)
template <typename STK>
struct hf_walk_t {
    uint32_t X, Y, cur, pend; // node (X,Y) of level L, its order-space children still to visit; candidate cells
    STK stk;                  // 4 bits per level: the ancestors' pending children
    int L;                    // level (1 while the lane holds candidate cells: they are the children of node (X,Y))
    bool fin;                 // subtree exhausted
};
typedef hf_walk_t<uint64_t> hf_walk;
__device__ __forceinline__ uint32_t stk_ctz(uint64_t v) { return (uint32_t) __builtin_ctzll(v); }
__device__ __forceinline__ uint32_t stk_ctz(uint32_t v) { return (uint32_t) __builtin_ctz(v); }
// start one level above the subtree root: a virtual parent whose only pending child is the root
template <typename STK>
__device__ __forceinline__ void walk_init(hf_walk_t<STK> &w, uint32_t X0, uint32_t Y0, int L0) {
    w.X = X0 >> 1; w.Y = Y0 >> 1; w.cur = 1u << ((X0 & 1u) | ((Y0 & 1u) << 1)); w.pend = 0u;
    w.stk = 0; w.L = L0 + 1; w.fin = false;
}
// One round of the walk for every lane of the call: walk until parked or done, then the parked blocks, then
// the candidate cells.  Lanes with w.fin set do nothing.  Returns whether this lane recorded a hit.
template <bool ANY, typename Src, typename STK>
__device__ __forceinline__ bool walk_round(const hf_dev_field &f, const Src &src, const hf_ray_state &rs,
                                           const hf_trav &r, bool fx, bool fy, uint32_t fxm, uint32_t fym,
                                           float &thi, hf_hit &best, hf_walk_t<STK> &w) {
    auto loadh = [&src](int i, int j) { return src.height(i, j); };
    bool hit_any = false;
    // ---- walk until this lane holds candidate cells or has exhausted the subtree ----
    while (!w.fin && w.pend == 0u) {
        WCOUNT(3);
        if (w.cur == 0u) {
            // Node exhausted: pop.  Every level between here and the nearest ancestor with pending children is
            // skipped at once (the masks are 4-bit fields of one register: count the empty ones); the virtual
            // parent of the subtree root holds no children, so an all-zero stack means the subtree is done.
            if (w.stk == 0) { w.fin = true; break; }
            const uint32_t z = stk_ctz(w.stk) >> 2;
            w.stk >>= 4u * z;
            w.cur = (uint32_t) w.stk & 15u; w.stk >>= 4;
            w.X >>= z + 1u; w.Y >>= z + 1u; w.L += (int) z + 1;
        }
        const uint32_t k = (uint32_t) __builtin_ctz(w.cur);
        w.cur &= w.cur - 1u;
        const uint32_t cx = 2u * w.X + (k & 1u), cy = 2u * w.Y + (k >> 1);
        const float S = (float) (1u << (w.L - 1));
        // the mask may predate a hit: re-check the child's entry against the current t_hi
        const float te = fmaxf(((float) cx * S - r.gxm) * r.idx, ((float) cy * S - r.gym) * r.idy);
        if (te > thi) continue;
        WCOUNT(5); WLANES(7, 0); WHIST(11);
        w.stk = (w.stk << 4) | (STK) w.cur;
        w.X = cx; w.Y = cy; --w.L;
        // the four children of (X,Y,L): sheared bounds on the fine levels, min/max boxes above
        const float Sc = 0.5f * S;
        const uint32_t ix = w.X ^ (fxm >> w.L), iy = w.Y ^ (fym >> w.L);
        hf_quad q;
        float gz, dz, mz;
        {   // A wave-level gather returns when its slowest lane does (an L2 / Infinity Cache round trip, not an
            // L1 hit), so the plane and the child ranges of the record are requested together: one memory round
            // trip per visit.  Levels above HF_SHEAR_TOP carry the zero plane (hf_shear_minmax_kernel).
            const float4 *rec = src.sheared(w.L, ix, iy);
            const float4 pl = rec[0], q01 = rec[1], q23 = rec[2];
            shear_line(f, rs, fx, fy, pl.x, pl.y, pl.z, pl.w, __builtin_fmaf((float) w.X, S, Sc),
                       __builtin_fmaf((float) w.Y, S, Sc), gz, dz, mz);
            q.lo[0] = q01.x; q.hi[0] = q01.y; q.lo[1] = q01.z; q.hi[1] = q01.w;
            q.lo[2] = q23.x; q.hi[2] = q23.y; q.lo[3] = q23.z; q.hi[3] = q23.w;
        }
        const uint32_t m = child_mask(r, fx, fy, (float) w.X * S, (float) w.Y * S, Sc, q, gz, dz, mz, thi);
        if (w.L == 1) {
            // the children of a level-1 node are cells: they become this lane's candidates (ACTUAL numbering
            // j = 2 jy + jx, tested in any order -- the result is order independent) and the lane stops walking
            // until the converged triangle rounds below have run
            w.pend = m;
            w.cur = 0u;
        } else {
            w.cur = to_order(m, fx, fy);
        }
    }
    if (__ballot(w.pend != 0u) == 0ull) return false; // nobody holds candidate cells: every lane of the call is done
    // ---- candidate cells, one per lane per round ----
    while (__ballot(w.pend != 0u) != 0ull) {
        WCOUNT(6);
        if (w.pend != 0u) {
            WLANES(7, 16); WHIST(8);
            const int j = __builtin_ctz(w.pend);
            w.pend &= w.pend - 1u;
            // (the lane's node (X,Y) is the level-1 node whose children the candidate cells are: actual cell = 2 * actual node + j)
            const int cxx = (int) (2u * (w.X ^ (fxm >> 1))) + (j & 1), cyy = (int) (2u * (w.Y ^ (fym >> 1))) + (j >> 1);
            const float z00 = loadh(cyy, cxx) * f.s, z10 = loadh(cyy, cxx + 1) * f.s;
            const float z01 = loadh(cyy + 1, cxx) * f.s, z11 = loadh(cyy + 1, cxx + 1) * f.s;
            if (test_cell(f, cxx, cyy, z00, z10, z01, z11, rs.oo, rs.od, rs.maxt, best)) {
                hit_any = true;
                float tb = best.t - rs.tin;
                tb = tb + __builtin_fabsf(tb) * 1e-6f + 1e-30f;
                thi = fminf(thi, tb);
                if (ANY) { w.pend = 0u; w.fin = true; }
            }
        }
    }
    return hit_any;
}

template <bool ANY, typename Src>
__device__ __forceinline__ bool walk_subtree(const hf_dev_field &f, const Src &src, const hf_ray_state &rs,
                                             const hf_trav &r, bool fx, bool fy, uint32_t fxm, uint32_t fym,
                                             uint32_t X0, uint32_t Y0, int L0, float &thi, hf_hit &best) {
    hf_walk w;
    walk_init(w, X0, Y0, L0);
    bool hit_any = false;
    do {
        hit_any |= walk_round<ANY>(f, src, rs, r, fx, fy, fxm, fym, thi, best, w);
    } while (__ballot(!w.fin) != 0ull);
    return hit_any;
}

// wave-wide minimum / maximum of an unsigned value, result scalar (DPP within rows of 16, readlane across rows);
// every lane of the wave must call these
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t x = v;
    x = min(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    x = min(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    x = min(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x141, 0xF, 0xF, true)); // row_half_mirror
    x = min(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x140, 0xF, 0xF, true)); // row_mirror
    const uint32_t a = (uint32_t) __builtin_amdgcn_readlane((int) x, 0), b = (uint32_t) __builtin_amdgcn_readlane((int) x, 16);
    const uint32_t c = (uint32_t) __builtin_amdgcn_readlane((int) x, 32), d = (uint32_t) __builtin_amdgcn_readlane((int) x, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

// wave-wide minimum / maximum of a float, result uniform (DPP within rows of 16, readlane across rows); every lane calls
__device__ __forceinline__ float row_min_f32(float v) { // minimum over the lane's row of 16, in every lane of the row
    float x = v;
    x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)));  // quad_perm [1,0,3,2]
    x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)));  // quad_perm [2,3,0,1]
    x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true))); // row_half_mirror
    x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true))); // row_mirror
    return x;
}
__device__ __forceinline__ float wave_min_f32(float v) {
    const int xi = __builtin_bit_cast(int, row_min_f32(v));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_max_f32(float v) { return -wave_min_f32(-v); }
// Four row minima at once, one v_min_f32 with a DPP source per step (the compiler's own code for fminf(x, dpp(x)) is
// a v_mov_dpp, a canonicalising v_max and the v_min: 216 instructions for the sixteen reductions instead of 64).
// The four chains are interleaved, so a DPP read never follows the write of its register by less than the two
// wait states the hardware asks for; the leading s_nop covers the writes before the block.
__device__ __forceinline__ void row_min4_f32(float &a, float &b, float &c, float &d) {
    asm volatile("s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_min_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// Sixteen wave-wide minima for the price of sixteen row reductions (DPP) and one pass through LDS: the first lane
// of each row of 16 parks its row's sixteen minima (four 16-byte writes), lane q < 16 then combines the four rows of
// quantity q.  `red` = 64 floats of the wave's own LDS.  Result: quantity q in lane q (other lanes: undefined).
__device__ __forceinline__ float wave_min16(const float (&h)[16], uint32_t lane, float *red) {
    float rm[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) rm[q] = h[q];
#pragma unroll
    for (int q = 0; q < 16; q += 4) row_min4_f32(rm[q], rm[q + 1], rm[q + 2], rm[q + 3]);
    if ((lane & 15u) == 0u) {
        float4 *dst = (float4 *) (red + (lane & 48u)); // row r -> red[16 r ..]
        dst[0] = make_float4(rm[0], rm[1], rm[2], rm[3]); dst[1] = make_float4(rm[4], rm[5], rm[6], rm[7]);
        dst[2] = make_float4(rm[8], rm[9], rm[10], rm[11]); dst[3] = make_float4(rm[12], rm[13], rm[14], rm[15]);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t q = lane & 15u;
    const float v = fminf(fminf(red[q], red[16 + q]), fminf(red[32 + q], red[48 + q]));
    __builtin_amdgcn_wave_barrier();
    return v;
}

// ---------------------------------------------------------------------------------
// ITEM WALK (round 4): the subtrees below the beam sweep's hand-off nodes as a wave-wide work list instead of 64 private
// depth-first walks.  The per-lane walk left most of the wave idle -- a visit ran with 19 of 64 lanes, a cell round with
// 13 -- because every converged round lasts until its slowest lane has made its step, and a lane's steps are a
// dependent chain (rays that skim the surface crawl from 2x2-cell node to node).  Here a unit of work is an ITEM,
// (ray, node) or (ray, cell), on one of two stacks in the wave's LDS: the lanes that take a hand-off node push the
// children they are to visit (ballot + mbcnt prefix per child slot), and every round the wave pops up to 64 node items
// -- of any level: a visit is the same code at every level -- or up to 64 cell items.  Any lane takes any item: it
// pulls the ray's traversal constants from the owning lane's registers (ds_bpermute, no copy of the ray in LDS),
// evaluates the node's record exactly as a per-lane visit did (same shear_line / child_mask arithmetic) and pushes the
// children that pass; a cell item runs the two-triangle test and merges its hit into the ray's entry of an LDS table
// with one 64-bit minimum -- key = (t bits, ~prim_index): the closest hit, the higher primitive on a tie, what the
// brute-force loop produces (kdtree.h:2424-2448).  A crawling ray's chain is thereby spread over the idle lanes, and
// the hand-off nodes of a pass are worked off TOGETHER: items are pushed node after node and the list runs when it
// holds a wave's worth (or the pass ends), so a node that only a handful of lanes take no longer costs a walk of its
// own.  What is lost is front-to-back pruning between the children of a node and between the nodes of one run (t_hi
// takes effect at the next pop, after a cell round) -- harmless for the result (a minimum over a superset of the
// cells) and cheap while the extra items ride in lanes that would have idled.
// Round selection: cell tests when a full round of them is waiting or no node item is left, visits otherwise (so the
// cell stack never holds more than 63 + 4 x 64 items); a visit round takes fewer than 64 items when the children
// they may push (four each) would not fit the node stack -- down to one item per round, a plain depth-first walk whose
// stack grows by at most three per level -- so neither capacity is ever exceeded whatever the rays do.
#define HF_NODE_CAP (HF_ITEM_FLUSH + 4 * 64 + 12 + 20) // a candidate of the sweep pushes up to 4 x 64 items onto fewer than HF_ITEM_FLUSH waiting
                                                      // ones, and items_run needs 12 spare entries on top (one-item rounds grow the stack by up to 6)
#define HF_CELL_CAP 320
#ifndef HF_ITEM_FLUSH
#define HF_ITEM_FLUSH 32 // the list runs when it holds this many node items (or the pass ends)
#endif
struct hf_items_lds {
    unsigned long long best[64]; // per ray of the batch: (t bits << 32) | ~prim_index of its closest hit so far; all ones = none
    float2 uv[64];               // barycentrics of that hit
    // items: x | y << 4 (node / cell relative to its hand-off node, order space) | level << 8 | ray << 10 | slot << 16
    // (slot: the hand-off node's lane in the pass of the beam sweep)
    uint32_t nodes[HF_NODE_CAP + HF_CELL_CAP]; // the node stack, then the cell stack
};
__device__ __forceinline__ float bperm_f(int src4, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}
__device__ __forceinline__ void items_clear(hf_items_lds *q, uint32_t lane) {
    // (opaque: the pair of all-ones registers is otherwise hoisted out of the kernel's persistent loop and spilled)
    uint32_t ones = 0xFFFFFFFFu;
    asm volatile("" : "+v"(ones));
    q->best[lane] = ((unsigned long long) ones << 32) | (unsigned long long) ones;
}
// the lane's own ray: closest hit of the item walks so far -> `best` (minimum t, higher prim on a tie)
__device__ __forceinline__ void items_fold(hf_items_lds *q, uint32_t lane, hf_hit &best) {
    __builtin_amdgcn_wave_barrier();
    const unsigned long long mk = q->best[lane];
    if (mk != ~0ull) {
        const float2 uv = q->uv[lane];
        best_update(best, __builtin_bit_cast(float, (uint32_t) (mk >> 32)), uv.x, uv.y, ~(uint32_t) mk);
    }
}
// wave-wide inclusive prefix sum (DPP: shifts within rows of 16, then the row totals broadcast into the rows above)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    int x = (int) v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return (uint32_t) x;
}
// Every lane pushes the children (order-space mask m4; 0: none) of its node as items of level lvc, far child first:
// onto the cell stack where cell_item, the node stack otherwise.  One prefix sum places everybody (node counts in the low
// half word, cell counts in the high one): ballot-free, 4 predicated LDS writes.  tag = ray << 10 | slot << 16.
__device__ __forceinline__ void push_children(hf_items_lds *q, uint32_t &nn, uint32_t &ncell, uint32_t m4, bool cell_item,
                                              uint32_t tag, uint32_t lvc, uint32_t x2, uint32_t y2) {
    const uint32_t c = (uint32_t) __builtin_popcount(m4);
    const uint32_t cp = cell_item ? c << 16 : c;
    const uint32_t inc = wave_scan_add(cp);
    const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
    const uint32_t ex = inc - cp;
    // (one array: the cell stack follows the node stack, see hf_items_lds)
    uint32_t at = cell_item ? (uint32_t) HF_NODE_CAP + ncell + (ex >> 16) : nn + (ex & 0xFFFFu);
    const uint32_t base = (lvc << 8) | tag;
#pragma unroll
    for (int k = 3; k >= 0; --k)
        if ((m4 >> k) & 1u) q->nodes[at++] = (x2 + (uint32_t) (k & 1)) | ((y2 + (uint32_t) (k >> 1)) << 4) | base;
    nn += tot & 0xFFFFu; ncell += tot >> 16;
}
// Works the node stack (nn items) off.  All 64 lanes call this (wave-uniform control flow).  The hand-off node of an
// item is the node of lane `slot` of the current pass of the beam sweep: slab sb0 + slot / 4, cross coordinate nc of
// that lane (xm: slabs run along x); fx, fy: the wave's mirror flags.
static_assert(HF_ITEM_FLUSH - 1 + 4 * 64 <= HF_NODE_CAP - 12, "node stack: a candidate's children on top of the waiting items");
static_assert(63 + 4 * 64 <= HF_CELL_CAP, "cell stack: a round of level-1 visits on top of less than a round of cells");
template <bool ANY>
__device__ __forceinline__ void items_run(const hf_dev_field &f, const hf_ray_state &rs, float dxo, float dyo, bool fx, bool fy,
                                          bool xm, uint32_t sb0, uint32_t nc, uint32_t nn, float &thi, hf_items_lds *q, uint32_t lane) {
    const hf_trav &r = rs.r;
    const int top = f.top;
    const uint32_t fxm = fx ? ((1u << top) - 1u) : 0u, fym = fy ? ((1u << top) - 1u) : 0u;
    uint32_t nc_ = 0u; // cell items
    while ((nn | nc_) != 0u) { // wave-uniform
        WCOUNT(3);
        __builtin_amdgcn_wave_barrier(); // (same wave: LDS operations stay in order; this keeps the compiler from reordering them)
        if (nn != 0u && nc_ < 64u) {
            // ---- a round of visits: up to 64 node items from the top of the node stack ----
            uint32_t n = min(nn, 64u);
            {   // room for the (at most three per popped item, net) node items the round may add
                const int32_t room = (int32_t) HF_NODE_CAP - 12 - (int32_t) nn;
                n = min(n, max(room > 0 ? (uint32_t) room / 3u : 0u, 1u));
            }
            const bool mine = lane < n;
            const uint32_t it = mine ? q->nodes[nn - 1u - lane] : (lane << 10);
            nn -= n;
            const int src4 = (int) (((it >> 10) & 63u) << 2);
            const uint32_t slot = (it >> 16) & 63u;
            hf_trav rr;
            rr.gxm = bperm_f(src4, r.gxm); rr.gxp = bperm_f(src4, r.gxp); rr.gym = bperm_f(src4, r.gym); rr.gyp = bperm_f(src4, r.gyp);
            rr.gz = bperm_f(src4, r.gz); rr.dz = bperm_f(src4, r.dz); rr.idx = bperm_f(src4, r.idx); rr.idy = bperm_f(src4, r.idy);
            rr.mz = bperm_f(src4, r.mz);
            const float rdxo = bperm_f(src4, dxo), rdyo = bperm_f(src4, dyo), rthi = bperm_f(src4, thi);
            const uint32_t ck = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (slot << 2), (int) nc), sk = sb0 + (slot >> 2);
            uint32_t m4 = 0u;
            const uint32_t lv = (it >> 8) & 3u, x = it & 15u, y = (it >> 4) & 15u; // lv: 1 .. HF_SUBTREE_LEVEL - 1
            if (mine) {
                const uint32_t X0 = xm ? sk : ck, Y0 = xm ? ck : sk;
                const uint32_t X = (X0 << ((uint32_t) HF_SUBTREE_LEVEL - lv)) + x, Y = (Y0 << ((uint32_t) HF_SUBTREE_LEVEL - lv)) + y;
                const float S = (float) (1u << lv), Sc = 0.5f * S;
                // the item may predate a hit: its entry against the ray's current t_hi (any hit: a ray that has hit holds -1)
                const float te = fmaxf(((float) X * S - rr.gxm) * rr.idx, ((float) Y * S - rr.gym) * rr.idy);
                if (te <= rthi && (!ANY || rthi >= 0.f)) {
                    WCOUNT(5); WLANES(7, 0); WHIST(11);
                    const uint32_t ix = X ^ (fxm >> lv), iy = Y ^ (fym >> lv);
                    const uint32_t k = (uint32_t) top - lv;
                    const float4 *rec = f.shear + (size_t) (hf_depth_off((int) k) - 1u + (iy << k) + ix) * 3;
                    const float4 pl4 = rec[0], q01 = rec[1], q23 = rec[2]; // one round trip: the three parts are requested together
                    float gz, dz, mz;
                    shear_line_v(rr, rdxo, rdyo, HF_LINE_EPS * (float) (1 << top), fx, fy, pl4.x, pl4.y, pl4.z, pl4.w,
                                 __builtin_fmaf((float) X, S, Sc), __builtin_fmaf((float) Y, S, Sc), gz, dz, mz);
                    hf_quad qd;
                    qd.lo[0] = q01.x; qd.hi[0] = q01.y; qd.lo[1] = q01.z; qd.hi[1] = q01.w;
                    qd.lo[2] = q23.x; qd.hi[2] = q23.y; qd.lo[3] = q23.z; qd.hi[3] = q23.w;
                    m4 = to_order(child_mask(rr, fx, fy, (float) X * S, (float) Y * S, Sc, qd, gz, dz, mz, rthi), fx, fy);
                }
            }
            const uint32_t tag = it & 0x3FFC00u; // ray and slot
            push_children(q, nn, nc_, m4, lv == 1u, tag, lv - 1u, 2u * x, 2u * y); // (m4 = 0 in the lanes without an item)
            TSTAMP(4); // visit rounds
        } else {
            // ---- a round of cell tests: up to 64 cell items from the top of the cell stack ----
            const uint32_t n = min(nc_, 64u);
            const bool mine = lane < n;
            const uint32_t it = mine ? q->nodes[(uint32_t) HF_NODE_CAP + nc_ - 1u - lane] : (lane << 10);
            nc_ -= n;
            const uint32_t ray = (it >> 10) & 63u, slot = (it >> 16) & 63u;
            const int src4 = (int) (ray << 2);
            const v3 oo = mk3(bperm_f(src4, rs.oo.x), bperm_f(src4, rs.oo.y), bperm_f(src4, rs.oo.z));
            const v3 od = mk3(bperm_f(src4, rs.od.x), bperm_f(src4, rs.od.y), bperm_f(src4, rs.od.z));
            const float rmaxt = bperm_f(src4, rs.maxt);
            const float rthi = ANY ? bperm_f(src4, thi) : 0.f;
            const uint32_t ck = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (slot << 2), (int) nc), sk = sb0 + (slot >> 2);
            hf_hit b;
            b.hit = false; b.t = __builtin_inff(); b.u = 0.f; b.v = 0.f; b.prim = 0u;
            unsigned long long key = ~0ull;
            if (mine && (!ANY || rthi >= 0.f)) {
                WCOUNT(6); WLANES(7, 16); WHIST(8);
                const uint32_t X0 = xm ? sk : ck, Y0 = xm ? ck : sk, x = it & 15u, y = (it >> 4) & 15u;
                const int cxx = (int) (((X0 << HF_SUBTREE_LEVEL) + x) ^ fxm), cyy = (int) (((Y0 << HF_SUBTREE_LEVEL) + y) ^ fym);
                // (32-bit byte offsets from the uniform base: hf_create limits the grid to 2^30 vertices)
                const uint32_t off = ((uint32_t) cyy * (uint32_t) f.W + (uint32_t) cxx) << 2, pitch = (uint32_t) f.W << 2;
                const char *hb = (const char *) f.h;
                const float z00 = *(const float *) (hb + off) * f.s, z10 = *(const float *) (hb + off + 4u) * f.s;
                const float z01 = *(const float *) (hb + off + pitch) * f.s, z11 = *(const float *) (hb + off + pitch + 4u) * f.s;
                if (test_cell(f, cxx, cyy, z00, z10, z01, z11, oo, od, rmaxt, b)) {
                    key = ((unsigned long long) __builtin_bit_cast(uint32_t, b.t) << 32) | (unsigned long long) (~b.prim);
                    atomicMin(&q->best[ray], key);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (b.hit && q->best[ray] == key) q->uv[ray] = make_float2(b.u, b.v); // the (unique) winner's barycentrics
            __builtin_amdgcn_wave_barrier();
            // every lane: its own ray's closest hit so far bounds what is left of its segment
            const unsigned long long mk = q->best[lane];
            if (mk != ~0ull) {
                float tb = __builtin_bit_cast(float, (uint32_t) (mk >> 32)) - rs.tin;
                tb = tb + __builtin_fabsf(tb) * 1e-6f + 1e-30f;
                thi = ANY ? -1.f : fminf(thi, tb);
            }
            TSTAMP(5); // cell rounds
        }
    }
}

// Coherent wave, upper levels by BEAM SWEEP.  The row sweep above pays per enumerated node: a uniform (scalar) load of
// its box, whose latency nothing hides, a per-lane box test and a ballot -- 16.8 nodes per batch of which 4.2 are
// entered, and 7.8 rows with two wave reductions each: a fifth of a traversing batch's instructions.  Here the
// lanes of the wave act as NODE testers first: the wave's rays are bounded by a beam (wave-wide extrema of the
// per-lane slab coefficients: any lane's parameter interval in a node lies inside the beam's, any lane's height
// inside the beam's height range over it), 16 rows x 4 node slots of the hand-off level are assigned to the 64
// lanes front to back, every lane loads ITS node's box and record in one round trip (addresses do not depend on
// loaded data), tests the beam against the box and parks box + record in the wave's LDS table.  The nodes that
// survive (a ballot: bit order = front-to-back order) are then taken one by one: box and record come back from
// LDS at a uniform address, every lane tests its own fat ray against the box with the row sweep's arithmetic, the
// node's record is evaluated for all lanes at once (the first visit of the hand-off, hoisted: a node the rays only
// skim is dropped here), and the lanes with children to visit walk them.  The beam test only removes nodes that no
// lane's own test would accept, so the visited set -- and the result -- is the row sweep's.
#ifndef HF_BEAM_ROWS
#define HF_BEAM_ROWS 16 // slabs per pass (x 4 node slots per slab = the 64 lanes; fewer: the upper lanes sit the enumeration out)
#endif
// The slabs are cut across the beam's MAJOR direction (rows of the level when the rays advance faster in y, columns
// when faster in x), so that a slab holds few nodes whatever the view: at most four, or the sweep declines.
// Front-to-back either way: a monotone ray that visits slab s before s' has s' > s, and inside a slab the cross
// coordinate only grows.
#define HF_BEAM_CAND (4 * HF_BEAM_ROWS)
struct hf_beam_lds {
    struct { float4 pl, q01, q23; } e[HF_BEAM_CAND]; // record of the pass's nodes ...
    float2 box[HF_BEAM_CAND];                          // ... and their (min z, max z)
    float hdr[20];               // the beam (wave-uniform), parked here between passes instead of in registers; [16] = largest xy margin
};
enum { BM_ISN, BM_ISX, BM_ICN, BM_ICX, BM_AS, BM_BS, BM_AC, BM_BC, BM_GCLO, BM_GCHI, BM_DCN, BM_DCX, BM_ZLO, BM_ZHI, BM_DZN, BM_DZX };
// Returns false when it gives up -- an axis-parallel ray in the wave, or a beam wider than four nodes (rays that fan
// out over a long path) -- at the start or between two passes: the caller then lets every live lane walk from the
// root with the t_hi and the best hit reached so far (re-testing a cell is harmless: the result is a minimum).
template <bool ANY>
__device__ __forceinline__ bool walk_beam_impl(const hf_dev_field &f, const hf_ray_state &rs, bool alive, bool fx, bool fy,
                                               float &thi, hf_beam_lds *lds, hf_items_lds *items) {
    const hf_trav &r = rs.r;
    // (opaque copies: everything derived from the lane number or the level count is loop-invariant for the kernel's
    // persistent loop, and the compiler would compute it all at kernel entry and then spill it -- 30 registers)
    int top = f.top;
    asm volatile("" : "+s"(top));
    uint32_t lane = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane));
    const uint32_t kd = (uint32_t) (top - HF_SUBTREE_LEVEL), nn = 1u << kd;
    const float S = (float) (1u << HF_SUBTREE_LEVEL), iS = 1.0f / S, lim = (float) nn - 0.5f;
    const float inf = __builtin_inff();
    const float dxo = __builtin_fabsf(rs.od.x) * f.hx, dyo = __builtin_fabsf(rs.od.y) * f.hy;
    // slab axis s / cross axis c (wave-uniform choice from the first live lane; any choice is correct)
    const int first = __builtin_ctzll(__ballot(alive));
    const bool xm = __builtin_amdgcn_readlane((int) (dxo > dyo), first) != 0;
    uint32_t smin, smax;
    {
        // ---- the beam: wave-wide extrema, all as minima (a maximum is the negated minimum of the negated value; dead
        // lanes contribute +inf) ----
        const float gsm = xm ? r.gxm : r.gym, gsp = xm ? r.gxp : r.gyp, ids = xm ? r.idx : r.idy, dso = xm ? dxo : dyo;
        const float gcm = xm ? r.gym : r.gxm, gcp = xm ? r.gyp : r.gxp, idc = xm ? r.idy : r.idx, dco = xm ? dyo : dxo;
        float h[16];
        h[BM_ISN] = ids; h[BM_ISX] = -ids; h[BM_ICN] = idc; h[BM_ICX] = -idc;
        h[BM_AS] = -(gsm * ids); h[BM_BS] = gsp * ids; h[BM_AC] = -(gcm * idc); h[BM_BC] = gcp * idc;
        h[BM_GCLO] = gcp; h[BM_GCHI] = -gcm; h[BM_DCN] = dco; h[BM_DCX] = -dco;
        h[BM_ZLO] = r.gz - r.mz; h[BM_ZHI] = -(r.gz + r.mz); h[BM_DZN] = r.dz; h[BM_DZX] = -r.dz;
#pragma unroll
        for (int q = 0; q < 16; ++q) h[q] = alive ? h[q] : inf;
        const float hv = wave_min16(h, lane, (float *) &lds->e[0]); // (the node table is not in use yet)
        // an axis-parallel ray in the wave (1/d = inf): decline
        const float isx = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), BM_ISX));
        const float icx = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), BM_ICX));
        if (!(isx < inf) || !(icx < inf)) return false;
        if (lane < 16u) lds->hdr[lane] = hv;
        // slabs the wave's rays can touch: [g - m, g + m + t_hi d] per lane, widened by the slack of this estimate
        const float sa = gsp - 1e-3f, sb = __builtin_fmaf(thi, dso, gsm);
        const float sb2 = sb + 1e-3f + 1e-6f * sb;
        {   // largest xy margin of the wave's rays: the needle term of the beam's box test is m x (height range of the box)
            const float mmax = wave_max_f32(alive ? rs.m : 0.f);
            if (lane == 0u) lds->hdr[16] = mmax;
        }
        smin = wave_min_u32(alive ? (uint32_t) fminf(fmaxf(sa * iS, 0.f), lim) : 0xFFFFFFFFu);
        smax = wave_max_u32(alive ? (uint32_t) fminf(fmaxf(sb2 * iS, 0.f), lim) : 0u);
    }
    TSTAMP(1); // beam set-up: sixteen wave-wide extrema, slab range
    for (uint32_t sb0 = smin; sb0 <= smax; sb0 += HF_BEAM_ROWS) { // wave-uniform
        WCOUNT(0);
        __builtin_amdgcn_wave_barrier();
        // ---- this lane's node of the pass: slab sb0 + lane / 4, slot lane % 4 of the beam's node range in that slab ----
        uint32_t nc; // the node's cross coordinate (its slab number follows from the lane)
        bool cand = false;
        {
            const float4 h0 = *(const float4 *) &lds->hdr[0], h1 = *(const float4 *) &lds->hdr[4];
            const float4 h2 = *(const float4 *) &lds->hdr[8], h3 = *(const float4 *) &lds->hdr[12];
            const float isn = h0.x, isx = -h0.y, icn = h0.z, icx = -h0.w, as = -h1.x, bs = h1.y, ac = -h1.z, bc = h1.w;
            const float gclo = h2.x, gchi = -h2.y, dcn = h2.z, dcx = -h2.w, zlo = h3.x, zhi = -h3.y, dzn = h3.z, dzx = -h3.w;
            const float T = wave_max_f32(thi);
            const uint32_t sl = sb0 + (lane >> 2);
            const float fS = (float) sl * S;
            const float t0 = fmaxf(fS * isn - as, 0.f), t1 = fminf((fS + S) * isx - bs, T);
            const float t0s = fmaxf(t0 - (1e-5f * t0 + 1e-5f), 0.f), t1s = t1 + (1e-5f * t1 + 1e-5f); // rounding of the beam's own arithmetic
            const bool in_slab = (HF_BEAM_CAND >= 64 || lane < (uint32_t) HF_BEAM_CAND) & (sl <= smax) & (t0s <= t1s);
            const float ca = __builtin_fmaf(t0s, dcn, gclo) - 1e-3f, cb = __builtin_fmaf(t1s, dcx, gchi);
            const float cb2 = cb + 1e-3f + 1e-6f * cb;
            const uint32_t c0 = (uint32_t) fminf(fmaxf(ca * iS, 0.f), lim), c1 = (uint32_t) fminf(fmaxf(cb2 * iS, 0.f), lim);
            if (__ballot(in_slab && c1 - c0 > 3u) != 0ull) return false; // a beam wider than four nodes
            const uint32_t cc = c0 + (lane & 3u);
            const bool valid = in_slab & (cc <= c1);
            const uint32_t i = xm ? sl : cc, j = xm ? cc : sl;
            nc = cc;
            if (valid) {
                const uint32_t ai = fx ? nn - 1u - i : i, aj = fy ? nn - 1u - j : j;
                const uint32_t node = (aj << kd) + ai;
                const float2 box = (f.mip + hf_depth_off((int) kd))[node];
                const float4 *rec = f.shear + ((size_t) (hf_depth_off((int) kd) - 1u) + node) * 3;
                const float4 pl = rec[0], q01 = rec[1], q23 = rec[2];
                const float fC = (float) cc * S;
                float u0 = fmaxf(fmaxf(fS * isn - as, fC * icn - ac), 0.f), u1 = fminf(fminf((fS + S) * isx - bs, (fC + S) * icx - bc), T);
                u0 = fmaxf(u0 - (1e-5f * u0 + 1e-5f), 0.f); u1 = u1 + (1e-5f * u1 + 1e-5f);
                const float za = zlo + fminf(u0 * dzn, u1 * dzn), zb = zhi + fmaxf(u0 * dzx, u1 * dzx);
                const float zs = __builtin_fmaf(2.f * lds->hdr[16], box.y - box.x, 1e-5f * (__builtin_fabsf(za) + __builtin_fabsf(zb)) + 1e-30f);
                cand = (u0 <= u1) & (za - zs <= box.y) & (zb + zs >= box.x);
                if (cand) {
                    lds->box[lane] = box;
                    lds->e[lane].pl = pl; lds->e[lane].q01 = q01; lds->e[lane].q23 = q23;
                }
            }
        }
        uint64_t cm = __ballot(cand);
        TSTAMP(2); // a pass: node assignment, box + record loads, beam test, LDS table
        __builtin_amdgcn_wave_barrier(); // the table is read below by every lane (same wave: LDS operations stay in order)
        // One loop with ONE walk site at its end: a lane that takes a node notes it (node + children packed in one
        // register: hf_create allows at most 2^11 nodes per side at this level) and the walk below runs at once.  The
        // shape matters more than it should: the same sweep with the walk inside the candidate's own branch spills 13
        // registers in the fused kernel instead of 2 (2.90-3.00 vs 2.71 ms, profiles/r03_ab/r03_v2).  Holding the nodes
        // of a pass and walking them together when a holder wants a second one was measured too: a lane that will hit in
        // its node still passes the box and record tests of the nodes behind it, so nearly every node flushes -- 2.1
        // instead of 2.2 walks, 9.2 instead of 6.8 box tests, 2.80 ms.
        // The candidates front to back: box test and record per lane, the children a lane is to visit go onto the wave's
        // work list; the list runs (items_run) when it holds a wave's worth of items or the pass ends.  (Until round 4 a
        // node was walked at once by the lanes that took it, each lane on its own: 2.2 walks per batch with half of the
        // lanes in each, every one as long as its slowest lane.)
        uint32_t nn = 0u; // node items waiting
        bool done = false;
        while (cm != 0ull) { // wave-uniform
            const uint32_t k = (uint32_t) __builtin_ctzll(cm);
            WCOUNT(1);
            const uint32_t ck = (uint32_t) __builtin_amdgcn_readlane((int) nc, (int) k), sk = sb0 + (k >> 2);
            const uint32_t ci = xm ? sk : ck, nodej = xm ? ck : sk;
            const float2 cbox = lds->box[k]; // uniform address: broadcast
            // ---- per-lane test of the node's box with the row sweep's arithmetic ----
            const float fXc = (float) ci * S, fYc = (float) nodej * S;
            const float xlo = (fXc - r.gxm) * r.idx, xhi = (fXc + S - r.gxp) * r.idx;
            const float ylo = (fYc - r.gym) * r.idy, yhi = (fYc + S - r.gyp) * r.idy;
            const float u0 = fmaxf(fmaxf(xlo, ylo), 0.f), u1 = fminf(fminf(xhi, yhi), thi);
            const float za = __builtin_fmaf(u0, r.dz, r.gz), zb = __builtin_fmaf(u1, r.dz, r.gz);
            const float bz = __builtin_fmaf(2.f * rs.m, cbox.y - cbox.x, r.mz); // needle term: 2 m x (range), see hf_shear_kernel
            const bool mine = (u0 <= u1) & (fminf(za, zb) - bz <= cbox.y) & (fmaxf(za, zb) + bz >= cbox.x);
            cm &= cm - 1ull;
            if (__ballot(mine) == 0ull) {
                // nobody overlaps this node; done when nobody can reach its slab -- or any later one -- before its t_hi
                // (a t_hi that waiting items have yet to shorten only delays this)
                const float tsk = ((float) sk * S - (xm ? r.gxm : r.gym)) * (xm ? r.idx : r.idy);
                if (__ballot(tsk <= thi) == 0ull) { done = true; cm = 0ull; }
            } else {
                WCOUNT(2);
                // the node's own record, for all lanes at once (the first visit of the hand-off, hoisted)
                const float4 pl = lds->e[k].pl, q01 = lds->e[k].q01, q23 = lds->e[k].q23;
                const float Sc = 0.5f * S;
                float gz, dz, mz;
                shear_line_v(r, dxo, dyo, HF_LINE_EPS * (float) (1 << top), fx, fy, pl.x, pl.y, pl.z, pl.w, fXc + Sc, fYc + Sc, gz, dz, mz);
                hf_quad q;
                q.lo[0] = q01.x; q.hi[0] = q01.y; q.lo[1] = q01.z; q.hi[1] = q01.w;
                q.lo[2] = q23.x; q.hi[2] = q23.y; q.lo[3] = q23.z; q.hi[3] = q23.w;
                const uint32_t m4 = child_mask(r, fx, fy, fXc, fYc, Sc, q, gz, dz, mz, thi);
                const uint32_t cur0 = mine ? to_order(m4, fx, fy) : 0u;
                if (__ballot(cur0 != 0u) != 0ull) WCOUNT(4);
                {
                    uint32_t no_cells = 0u;
                    push_children(items, nn, no_cells, cur0, false, (lane << 10) | (k << 16), (uint32_t) (HF_SUBTREE_LEVEL - 1), 0u, 0u);
                }
            }
            // ---- the work list: when it holds a wave's worth, and before the pass ends ----
            if (nn >= (uint32_t) HF_ITEM_FLUSH || (cm == 0ull && nn != 0u)) {
                TSTAMP(3); // candidates: per-lane box tests, hoisted records, pushes
                items_run<ANY>(f, rs, dxo, dyo, fx, fy, xm, sb0, nc, nn, thi, items, lane);
                nn = 0u;
                if (ANY && __ballot(thi >= 0.f) == 0ull) return true;
            }
        }
        if (done) return true;
        // done when nobody can reach the first slab of the next pass before its t_hi
        const float gsm = xm ? r.gxm : r.gym, ids = xm ? r.idx : r.idy;
        const float tsn = ((float) (sb0 + HF_BEAM_ROWS) * S - gsm) * ids;
        if (__ballot(tsn <= thi) == 0ull) break;
    }
    return true;
}

// the beam sweep with the item walk's hit table around it: cleared before, folded into `best` after (every exit)
template <bool ANY>
__device__ __forceinline__ bool walk_beam(const hf_dev_field &f, const hf_ray_state &rs, bool alive, bool fx, bool fy,
                                          hf_hit &best, float &thi, hf_beam_lds *lds, hf_items_lds *items) {
    const uint32_t lane = threadIdx.x & 63u;
    items_clear(items, lane);
    const bool done = walk_beam_impl<ANY>(f, rs, alive, fx, fy, thi, lds, items);
    items_fold(items, lane, best);
    return done;
}

// ---------------------------------------------------------------------------------
// Warped-area reparameterisation (include/hf.h; reparam.py:10-123,224-333): per-sample kernels
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void tea32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1) { // random.h:76-91
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}

struct hf_aux_sample {
    v3 omega;     // square_to_von_mises_fisher(sample, kappa), xy negated on the flipped half of a pair
    float sy;     // sample.y
    v3 fs, ft;    // Frame3f(d): s, t (n = d)
};
__device__ __forceinline__ void aux_sample(const hf_reparam_args &a, size_t i, v3 d, hf_aux_sample &q) {
    const uint32_t pair = a.antithetic ? (a.k >> 1) : a.k;
    // The stream of a sample: (seed, pair) hashed TOGETHER (added, the streams of seed s, pair p and seed s + 1,
    // pair p - 1 were the same one), keyed by the ray's id -- its index in the launch, or ray_id[i] when the caller
    // says which ray of a larger wavefront this is (a rank's tiles of a partitioned wavefront: the sharded launch then
    // draws the samples of the unsharded one).
    uint32_t key, unused, r0, r1;
    tea32(a.seed, pair, key, unused);
    tea32(key, a.ray_id ? a.ray_id[i] : (uint32_t) i, r0, r1);
    const float sx = (float) (r0 >> 9) * (1.0f / 8388608.0f), sy = (float) (r1 >> 9) * (1.0f / 8388608.0f);
    // warp.h:557-566
    const float syc = fmaxf(1.f - sy, 1e-6f);
    const float cos_theta = 1.f + logf(__builtin_fmaf(1.f - syc, expf(-2.f * a.kappa), syc)) / a.kappa;
    float sn, cs;
    sincospif(2.f * sx, &sn, &cs); // sin / cos of 2 pi sx without the range reduction of a radian argument
    const float sin_theta = __builtin_sqrtf(fmaxf(1.f - cos_theta * cos_theta, 0.f));
    const bool flip = a.antithetic && ((a.k & 1u) == 0u); // reparam.py:83-85,189
    q.omega = mk3(flip ? -(cs * sin_theta) : cs * sin_theta, flip ? -(sn * sin_theta) : sn * sin_theta, cos_theta);
    q.sy = sy;
    coordinate_system(d, q.fs, q.ft);
}
// Frame3f::to_world (frame.h:39-41)
__device__ __forceinline__ v3 frame_to_world(const hf_aux_sample &q, v3 n, v3 v) {
    return mk3(__builtin_fmaf(n.x, v.z, __builtin_fmaf(q.ft.x, v.y, q.fs.x * v.x)),
               __builtin_fmaf(n.y, v.z, __builtin_fmaf(q.ft.y, v.y, q.fs.y * v.x)),
               __builtin_fmaf(n.z, v.z, __builtin_fmaf(q.ft.z, v.y, q.fs.z * v.x)));
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

// record stores are write-once streams (72 B/ray): non-temporal, so that they do not push the mip
// and height lines of concurrently traversing waves out of L2.  Address = (row + ub) + lo: uniform base, lane offset.
__device__ __forceinline__ void st(float *p, size_t ub, uint32_t lo, float v) { if (p) __builtin_nontemporal_store(v, &(p + ub)[lo]); }
typedef float f4 __attribute__((ext_vector_type(4)));
// four consecutive entries of a row per lane (the wide path): (row + ub) as 16-byte elements, lane offset l4
__device__ __forceinline__ void st4(float *p, size_t ub, uint32_t l4, f4 v) { if (p) __builtin_nontemporal_store(v, &((f4 *) (p + ub))[l4]); }
#define st3(p, ub, lo, v) do { st((p)[0], ub, lo, (v).x); st((p)[1], ub, lo, (v).y); st((p)[2], ub, lo, (v).z); } while (0)

#ifndef HF_GRAB
#define HF_GRAB 256 // most rays a wave takes from the work counter per fetch (hf_grab_for); 512 before the per-XCD counters
#endif
// Scratch block of one trace launch (zeroed by hf_launch_trace): the per-XCD work counters.
#define HF_SCR_BYTES 1024
#ifndef HF_PRIO
#define HF_PRIO 1     // s_setprio of a batch's traversal (0: off) ...
#endif
#ifndef HF_PRIO_OUT
#define HF_PRIO_OUT 0 // ... and of its output phase; the wide path runs at 0
#endif
#ifndef HF_COH_WINDOW
#define HF_COH_WINDOW 32.f // a wave is coherent when its entry points are within this many cells of the first live lane's
#endif
#ifndef HF_COH_DIR
#define HF_COH_DIR 0.05f   // ... and the xy direction ratios within this relative distance
#endif
#ifndef HF_TRACE_WAVES_FUSED
#define HF_TRACE_WAVES_FUSED 5 // ... of the fused mode: its surface-interaction tail needs ~100 registers (6 waves: 30 spills, slower)
#endif
#ifndef HF_TRACE_WAVES
#define HF_TRACE_WAVES 5 // resident waves per SIMD = workgroups per CU of the traversal kernel (96 VGPRs; 6 waves = 80 VGPRs spill 40 of them since the beam sweep: closest hit 2.76 vs 2.29 ms)
#endif
#define HF_NUM_XCD 8u // work counters per launch, one per XCD (power of two)
#define HF_COUNTER_STRIDE 16u // in counters: 128 bytes apart

// the one kernel argument (kernarg segment offset 0)
struct hf_trace_args {
    hf_dev_field f;
    size_t n;
    hf_rays_dev rays;
    const uint8_t *active;
    hf_pi_dev pi;
    uint8_t *hit_out;
    hf_si_dev sio;
    uint32_t flags;
    uint32_t grab; // rays per fetch, a multiple of 64
    uint32_t wide; // every ray / record row is 16-byte aligned and there is no `active` mask: fetches that miss the bound as a whole take the wide path
    unsigned long long *counter;
    unsigned long long n_grabs; // ceil(n / grab)
    // fused mode only: trace auxiliary ray aux_k of every ray instead of the ray itself (hf_reparam_trace): the
    // direction is replaced by hf_reparam_aux_kernel's, maxt by infinity
    uint32_t aux_on, aux_k, aux_seed;
    float aux_kappa;
    int aux_antithetic;
    const uint32_t *aux_ray_id;
    // ... samples aux_k .. aux_k + aux_n - 1 in ONE launch (hf_reparam_trace_all): sample j of ray i is written at
    // [j * aux_stride + i] of every output row; aux_cull > 0: sqrt(2) ||to_object|| tan(theta_max) of cone_maybe_alive
    uint32_t aux_n;
    size_t aux_stride;
    float aux_cull;
};

// member-wise copy out of the kernarg segment (constant address space)
__device__ __forceinline__ hf_dev_field load_field(const __attribute__((address_space(4))) hf_dev_field *p) {
    hf_dev_field f;
    f.h = p->h; f.mip = p->mip; f.shear = p->shear;
    f.W = p->W; f.H = p->H; f.top = p->top;
    f.s = p->s; f.sx = p->sx; f.sy = p->sy; f.iu = p->iu; f.iv = p->iv; f.hx = p->hx; f.hy = p->hy; f.flip = p->flip;
#pragma unroll
    for (int k = 0; k < 12; ++k) { f.to_world[k] = p->to_world[k]; f.to_object[k] = p->to_object[k]; }
    return f;
}

typedef const __attribute__((address_space(4))) hf_trace_args *hf_kargs_ptr;
__device__ __forceinline__ hf_rays_dev load_rays(hf_kargs_ptr ka) {
    hf_rays_dev r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.o[k] = ka->rays.o[k]; r.d[k] = ka->rays.d[k]; }
    r.maxt = ka->rays.maxt;
    return r;
}
__device__ __forceinline__ hf_pi_dev load_pi(hf_kargs_ptr ka) {
    hf_pi_dev p;
    p.t = ka->pi.t; p.u = ka->pi.u; p.v = ka->pi.v; p.prim = ka->pi.prim;
    return p;
}
__device__ __forceinline__ hf_si_dev load_si(hf_kargs_ptr ka) {
    hf_si_dev d;
    d.t = ka->sio.t; d.bt = ka->sio.bt;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d.p[k] = ka->sio.p[k]; d.n[k] = ka->sio.n[k]; d.sh_n[k] = ka->sio.sh_n[k]; d.dp_du[k] = ka->sio.dp_du[k];
        d.dp_dv[k] = ka->sio.dp_dv[k]; d.sh_s[k] = ka->sio.sh_s[k]; d.sh_t[k] = ka->sio.sh_t[k]; d.wi[k] = ka->sio.wi[k];
    }
    d.uv[0] = ka->sio.uv[0]; d.uv[1] = ka->sio.uv[1];
    return d;
}

// compute_si_to sink of the fused mode: a field goes to memory when it is final; the row pointers are read from the
// kernarg segment at the store
struct hf_si_store_sink {
    hf_kargs_ptr ka;
    size_t ub;
    uint32_t lo, flags;
    __device__ __forceinline__ void s1(float *p, float v) { st(p, ub, lo, v); }
    __device__ __forceinline__ void s3(float *p0, float *p1, float *p2, v3 v) { st(p0, ub, lo, v.x); st(p1, ub, lo, v.y); st(p2, ub, lo, v.z); }
    __device__ __forceinline__ void t(float v) { s1(ka->sio.t, v); }
    __device__ __forceinline__ void p(v3 v) { s3(ka->sio.p[0], ka->sio.p[1], ka->sio.p[2], v); }
    __device__ __forceinline__ void boundary_test(float v) { if (flags & 0x40u) s1(ka->sio.bt, v); }
    __device__ __forceinline__ void uv(float a, float b) { s1(ka->sio.uv[0], a); s1(ka->sio.uv[1], b); }
    __device__ __forceinline__ void dp_du(v3 v) { s3(ka->sio.dp_du[0], ka->sio.dp_du[1], ka->sio.dp_du[2], v); }
    __device__ __forceinline__ void dp_dv(v3 v) { s3(ka->sio.dp_dv[0], ka->sio.dp_dv[1], ka->sio.dp_dv[2], v); }
    __device__ __forceinline__ void n(v3 v) {
        s3(ka->sio.n[0], ka->sio.n[1], ka->sio.n[2], v);
        s3(ka->sio.sh_n[0], ka->sio.sh_n[1], ka->sio.sh_n[2], v);
    }
    __device__ __forceinline__ void sh_s(v3 v) { s3(ka->sio.sh_s[0], ka->sio.sh_s[1], ka->sio.sh_s[2], v); }
    __device__ __forceinline__ void sh_t(v3 v) { s3(ka->sio.sh_t[0], ka->sio.sh_t[1], ka->sio.sh_t[2], v); }
    __device__ __forceinline__ void wi(v3 v) { s3(ka->sio.wi[0], ka->sio.wi[1], ka->sio.wi[2], v); }
};

// Persistent waves: every wave pulls `grab` consecutive rays at a time from a global
// counter (zeroed on the stream before the launch), so expensive image regions are
// spread over all CUs whatever their position in the wavefront.
// AUX (fused mode only): auxiliary ray a.aux_k of every ray is traced instead of the ray itself (hf_reparam_trace) -- an
// instantiation of its own, so that the sampling code costs the ordinary fused launch nothing (inline it was +1.5 %).
// LEAN (round 4): the instantiations for launches the caller declares INCOHERENT -- the `coherent = false` of
// Scene::ray_intersect / ray_test / ray_intersect_preliminary (scene.h:117-146, 188-207, 237-259: "a hint that can improve
// performance in the first step of finding the PreliminaryInteraction"; reparam.py:95 traces its auxiliary rays with it).
// No wave tries the beam sweep, so the sweep, the item walk and their 29 KB of LDS are not in the kernel: 71 VGPRs (79 with
// the sampling code) instead of 95, no spilled register, and 6 (7) waves per SIMD instead of 5.  The results are the same
// (the per-lane walk is what a wave falls back to anyway); bounce rays 3.46 -> 3.18 ms, the reparameterisation backward
// 20.4 -> 19.0 ms; a launch of coherent rays is up to 25 % slower with it (profiles/r04_ab/r04_lean).
#ifndef HF_LEAN_WAVES
#define HF_LEAN_WAVES 6
#endif
#ifndef HF_LEAN_WAVES_AUX
#define HF_LEAN_WAVES_AUX 7
#endif
template <int MODE, bool AUX = false, bool LEAN = false>
__global__ __launch_bounds__(HF_BLOCK, (LEAN ? (AUX ? HF_LEAN_WAVES_AUX : HF_LEAN_WAVES) : MODE == 2 ? HF_TRACE_WAVES_FUSED : HF_TRACE_WAVES))
void hf_trace_kernel(hf_trace_args a) {
    const hf_dev_field &f = a.f;
    const unsigned lane = threadIdx.x & 63u;
    __shared__ hf_beam_lds s_beam[HF_BLOCK / 64]; // per wave: the beam and box + record of the pass's nodes (walk_beam)
    __shared__ hf_items_lds s_items[HF_BLOCK / 64]; // per wave: work list and hit table of the item walk (walk_items)
    // Work distribution: the wavefront is cut into grabs of `grab` consecutive rays; XCD x owns grabs x, x + 8, ...
    // and hands them out through its own counter (a single counter serves ~80 fetches/us, which capped the rays
    // that only stream at half the memory bandwidth; eight addresses are served in parallel).  A wave pulls from the
    // counter of the XCD it runs on and moves on to the next XCD's when its own is exhausted, so all XCDs finish
    // together.  Grab numbers map to rays from the middle of the wavefront outwards (even: towards the end, odd:
    // towards the front): stream-only and traversal regions of a rendered wavefront then overlap in time.
    // (launch constants are read from the kernarg segment where they are needed -- see below -- so that the only
    // scalar state alive across a walk is the band, the current grab and the position in it)
    unsigned xc, tried = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
    xc &= HF_NUM_XCD - 1u;
    for (;;) {
        const __attribute__((address_space(4))) hf_trace_args *kg =
            (const __attribute__((address_space(4))) hf_trace_args *) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kg));
        unsigned long long *counter = kg->counter;
        const unsigned grab = kg->grab;
        const unsigned long long n_grabs = kg->n_grabs;
        unsigned long long g = 0;
#ifdef HF_TSTATS
        const long long tw0 = clock64(); // (wide-path stamps: before the grab number is asked for)
#endif
        if (lane == 0) g = atomicAdd(counter + (size_t) xc * HF_COUNTER_STRIDE, 1ull);
        g = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (g >> 32)) << 32) |
            (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (g & 0xffffffffull));
#ifdef HF_TSTATS
        const long long tw1 = clock64(); // ... the grab number is there
#endif
        const unsigned long long gg = g * HF_NUM_XCD + xc;
        // FOUR FRONTS (round 4): grab numbers map to rays along four fronts -- from the middle of the wavefront towards its
        // end and towards its beginning, and from both ends inwards; an XCD takes its grabs from the fronts in turn.  A
        // rendered wavefront is expensive in the middle rows of the image (the terrain) and all-miss at its top and bottom:
        // with the two fronts from the middle that the kernel had until round 4, the waves traversed first and the launch
        // ended in a phase in which all of them streamed miss records.  With fronts from the ends as well, half of the
        // grabs in flight are cheap and half expensive for the whole launch: fused -4 %, closest hit -2.7 %, any hit
        // -2.9 %; uniform launches (bounce rays) +1.6 % (four regions of the terrain in the caches instead of two) -- so the
        // LEAN instantiations, whose launches are uniform by declaration, keep the two fronts from the middle.
#ifdef HF_TWO_FRONTS
        const bool two_fronts = true;
#else
        const bool two_fronts = LEAN && !AUX; // (the auxiliary rays of a rendered wavefront are as uneven as the wavefront itself)
#endif
        unsigned long long base;
        if (!two_fronts) {
            const unsigned long long quarter = (n_grabs + 3ull) >> 2; // grabs per front (the last ones of two fronts may not exist)
            if (gg >= 4ull * quarter) {
                if (++tried == HF_NUM_XCD) break;
                xc = (xc + 1u) & (HF_NUM_XCD - 1u);
                continue;
            }
            const unsigned long long k = gg >> 2;
            const unsigned front = (unsigned) ((gg + (gg >> 5)) & 3ull); // (a permutation of the four fronts within every four grab numbers)
            const unsigned long long pos = front == 0u ? 2ull * quarter + k : front == 1u ? 2ull * quarter - 1ull - k :
                                           front == 2u ? 4ull * quarter - 1ull - k : k;
            if (pos >= n_grabs) continue; // (4 x quarter rounds n_grabs up: up to three positions beyond the wavefront)
            base = pos * grab;
        } else {
            if (gg >= n_grabs) {
                if (++tried == HF_NUM_XCD) break;
                xc = (xc + 1u) & (HF_NUM_XCD - 1u);
                continue;
            }
            base = ((gg & 1ull) ? (n_grabs >> 1) - 1ull - (gg >> 1) : (n_grabs >> 1) + (gg >> 1)) * grab;
        }
        // the ray of the batch in flight: requested one batch ahead (see below)
        v3 o = mk3(0.f, 0.f, 0.f), d = o;
        float maxt = 0.f;
#pragma unroll 1
        for (unsigned sub = 0; sub < grab; sub += 64) {
            if (!AUX && (sub & 255u) == 0u) {
                // ---- WIDE PATH: 256 rays that miss the bound as a whole (70 % of the bench wavefront) ----
                // Four rays per lane through 16-byte loads, the conservative clip (maybe_alive: false only where setup_ray
                // says false), and if nobody may enter the bound the 256 miss records go out as 16-byte stores: 37 memory
                // instructions and ~360 others per 256 rays where four ordinary batches take ~1900 -- and a quarter of the
                // memory round trips, i.e. of the time the wave holds its slot.  A fetch with any "maybe" ray takes the
                // ordinary path below (its rays are re-read: L2 hits).
                const __attribute__((address_space(4))) hf_trace_args *kw =
                    (const __attribute__((address_space(4))) hf_trace_args *) __builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kw));
                const size_t ubw = base + sub;
                if (kw->wide != 0u && sub + 256u <= grab && ubw + 256u <= kw->n) { // wave-uniform
#if HF_PRIO_OUT != 0
                    __builtin_amdgcn_s_setprio(0);
#endif
                    const uint32_t l4 = lane; // lane l: rays ubw + 4 l .. 4 l + 3
                    const hf_rays_dev rp = load_rays(kw);
                    const f4 ox = ((const f4 *) (rp.o[0] + ubw))[l4], oy = ((const f4 *) (rp.o[1] + ubw))[l4], oz = ((const f4 *) (rp.o[2] + ubw))[l4];
                    const f4 dx = ((const f4 *) (rp.d[0] + ubw))[l4], dy = ((const f4 *) (rp.d[1] + ubw))[l4], dz = ((const f4 *) (rp.d[2] + ubw))[l4];
                    const f4 mt = ((const f4 *) (rp.maxt + ubw))[l4];
                    bool maybe = false;
                    {
                        const hf_dev_field f0 = load_field(&kw->f);
                        const float2 zr = f0.mip[1];
#pragma unroll
                        for (int j = 0; j < 4; ++j) maybe |= maybe_alive(f0, zr, mk3(ox[j], oy[j], oz[j]), mk3(dx[j], dy[j], dz[j]), mt[j]);
                    }
#ifdef HF_TSTATS
                    asm volatile("s_waitcnt vmcnt(0)");
                    const long long tw2 = clock64(); // ... the rays are there and tested
#endif
                    if (__ballot(maybe) == 0ull) {
                        // (opaque: constant register quadruples are otherwise hoisted out of the kernel's persistent loop,
                        // spilled, and re-loaded before every store)
                        float z0 = 0.f, i0 = __builtin_inff();
                        asm volatile("" : "+v"(z0), "+v"(i0));
                        const f4 zero = { z0, z0, z0, z0 };
                        if (MODE == 1) {
                            ((uint32_t *) (kw->hit_out + ubw))[l4] = 0u;
                        } else {
                            const f4 inf4 = { i0, i0, i0, i0 };
                            const hf_pi_dev pi = load_pi(kw);
                            if (pi.t) st4(pi.t, ubw, l4, inf4);
                            if (pi.u) st4(pi.u, ubw, l4, zero);
                            if (pi.v) st4(pi.v, ubw, l4, zero);
                            if (pi.prim) st4((float *) pi.prim, ubw, l4, zero);
                            if (MODE == 2) { // zero-initialised record (interaction.h:479-499, 667-673), wi = -d
                                const uint32_t flags = kw->flags;
                                const hf_si_dev sd = load_si(kw);
                                st4(sd.t, ubw, l4, inf4);
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    st4(sd.p[c], ubw, l4, zero); st4(sd.n[c], ubw, l4, zero); st4(sd.sh_n[c], ubw, l4, zero);
                                    st4(sd.dp_du[c], ubw, l4, zero); st4(sd.dp_dv[c], ubw, l4, zero);
                                    st4(sd.sh_s[c], ubw, l4, zero); st4(sd.sh_t[c], ubw, l4, zero);
                                }
                                st4(sd.uv[0], ubw, l4, zero); st4(sd.uv[1], ubw, l4, zero);
                                if (flags & 0x40u) { const float b0 = z0 + 1e8f; const f4 big = { b0, b0, b0, b0 }; st4(sd.bt, ubw, l4, big); }
                                st4(sd.wi[0], ubw, l4, -dx); st4(sd.wi[1], ubw, l4, -dy); st4(sd.wi[2], ubw, l4, -dz);
                            }
                        }
#ifdef HF_TSTATS
                        if (MODE != 1 && sub == 0u) { // lane 0's four prim_uv[0] entries: cycles waiting for the grab number, for the rays + test, issuing the stores
                            const long long tw3 = clock64();
                            const hf_pi_dev pq = load_pi(kw);
                            if (lane == 0u && pq.u) { (pq.u + ubw)[0] = (float) (tw1 - tw0); (pq.u + ubw)[1] = (float) (tw2 - tw1); (pq.u + ubw)[2] = (float) (tw3 - tw2); (pq.u + ubw)[3] = 1.f; }
                        }
#endif
                        sub += 192u; // (+ 64 by the loop: the next fetch of the grab)
                        continue;
                    }
                }
            }
            // all 64 lanes stay in the loop body (the shared walk relies on whole-wave ballots);
            // lanes past the end of the wavefront re-read the last ray and store nothing.
            // Pointers and constants that are only needed before or after the walk are read from the kernarg
            // segment where they are used: held in scalar registers across the walk they get spilled into
            // vector-register lanes, and fetching them back costs vector instructions.
            const __attribute__((address_space(4))) hf_trace_args *ka =
                (const __attribute__((address_space(4))) hf_trace_args *) __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka)); // opaque: keeps these loads inside the loop body
            const size_t n = ka->n;
            // Addressing: every array is read / written at  (pointer + ub) + lo  with the wave-uniform element
            // offset ub and a 32-bit lane offset lo -- scalar base + vector offset, no 64-bit vector arithmetic.
            const size_t ub = base + sub;
            if (ub >= n) break; // wave-uniform
            const size_t left = n - ub; // >= 1
            const bool valid = lane < left;
            const uint32_t lo = valid ? lane : (uint32_t) (left - 1);
            if ((MODE == 2 && AUX) || (sub & 255u) == 0u) { // first batch of a fetch: nothing was requested ahead (see the end of the body; AUX never requests ahead)
                const hf_rays_dev rp = load_rays(ka);
                o = mk3((rp.o[0] + ub)[lo], (rp.o[1] + ub)[lo], (rp.o[2] + ub)[lo]);
                d = mk3((rp.d[0] + ub)[lo], (rp.d[1] + ub)[lo], (rp.d[2] + ub)[lo]);
                maxt = (rp.maxt + ub)[lo];
            }
            if (MODE == 2 && AUX) {
                // ---- all samples of all 64 rays at once: when no auxiliary ray of the batch can enter the bound (60 % of
                // the bench wavefront), the aux_n miss records go out without a single sample being drawn ----
                const float ct = ka->aux_cull;
                if (ct > 0.f) { // wave-uniform
                    bool maybe;
                    {
                        const hf_dev_field f0 = load_field(&ka->f);
                        maybe = valid && cone_maybe_alive(f0, f0.mip[1], o, d, ct);
                    }
                    if (__ballot(maybe) == 0ull) {
                        {
                            const uint32_t aux_n = ka->aux_n;
#pragma unroll 1
                            for (uint32_t ak = 0; ak < aux_n; ++ak) { // wave-uniform loop, the stores predicated inside
                                asm volatile("" : "+s"(ka)); // opaque per sample: the ~30 pointer tests stay in the loop instead of being hoisted into (spilled) scalar pairs
                                if (valid) {
                                    const uint32_t flags = ka->flags;
                                    const size_t ubo = ub + (size_t) ak * ka->aux_stride;
                                    const hf_pi_dev pi = load_pi(ka);
                                    if (pi.t) (pi.t + ubo)[lo] = __builtin_inff();
                                    if (pi.u) (pi.u + ubo)[lo] = 0.f;
                                    if (pi.v) (pi.v + ubo)[lo] = 0.f;
                                    if (pi.prim) (pi.prim + ubo)[lo] = 0u;
                                    hf_si_store_sink out = { ka, ubo, lo, flags };
                                    const v3 z = mk3(0.f, 0.f, 0.f); // (the launcher culls only when si.wi -- minus the sample's direction -- is not asked for)
                                    out.t(__builtin_inff()); out.p(z); out.boundary_test((flags & 0x40u) ? 1e8f : 0.f);
                                    out.uv(0.f, 0.f); out.dp_dv(z); out.n(z); out.dp_du(z); out.sh_s(z); out.sh_t(z);
                                }
                            }
                        }
                        continue;
                    }
                }
            }
#pragma unroll 1
            for (uint32_t ak = 0;; ++ak) { // (one trip unless AUX; the trip count is read from the kernarg segment at the end of the body)
            if (MODE == 2 && AUX) {
                asm volatile("" : "+s"(ka)); // opaque per sample: kernarg loads stay inside the loop instead of being hoisted (and spilled)
                // the ray again (an L1 / L2 hit from the second sample on): nothing of it is held across the walk
                const hf_rays_dev rp = load_rays(ka);
                o = mk3((rp.o[0] + ub)[lo], (rp.o[1] + ub)[lo], (rp.o[2] + ub)[lo]);
                const v3 dr = mk3((rp.d[0] + ub)[lo], (rp.d[1] + ub)[lo], (rp.d[2] + ub)[lo]);
                hf_reparam_args sa = {};
                sa.k = ka->aux_k + ak; sa.seed = ka->aux_seed; sa.kappa = ka->aux_kappa; sa.antithetic = ka->aux_antithetic;
                sa.ray_id = ka->aux_ray_id;
                hf_aux_sample q;
                aux_sample(sa, ub + lo, dr, q);
                d = frame_to_world(q, dr, q.omega);
                maxt = __builtin_inff();
            }
            hf_hit best;
            best.hit = false; best.t = __builtin_inff(); best.u = 0.f; best.v = 0.f; best.prim = 0u;
            const uint8_t *active = ka->active;
            const bool act = valid && (active ? ((active + ub)[lo] != 0) : true);
            TSTART();
            hf_ray_state rs = {}; // fully defined on every path: undefined fields become loop-carried registers
            bool alive;
            {
                const hf_dev_field f0 = load_field(&ka->f); // to_object etc.
                alive = act && setup_ray(f0, f0.mip[1], o, d, maxt, rs); // mip[1]: the global height range
            }
            const v3 dw = d; // world-space direction: wi of the record
            const uint64_t am = __ballot(alive);
            if (am != 0ull) {
                // coherent wave?  equal direction signs, entry points and directions close to the first live lane's
                const int src = __builtin_ctzll(am);
                // (v_readlane: the result is a scalar, so everything the shared walk derives from it stays scalar)
                const bool fx0 = __builtin_amdgcn_readlane((int) rs.fx, src) != 0, fy0 = __builtin_amdgcn_readlane((int) rs.fy, src) != 0;
                const float gx0 = __shfl(rs.gx, src), gy0 = __shfl(rs.gy, src);
                const float ux = rs.r.idy, uy = rs.r.idx; // direction ratio proxy: compare idx/idy cross products
                const float ux0 = __shfl(ux, src), uy0 = __shfl(uy, src);
                const bool near = (rs.fx == fx0) & (rs.fy == fy0) & (__builtin_fabsf(rs.gx - gx0) <= HF_COH_WINDOW) &
                                  (__builtin_fabsf(rs.gy - gy0) <= HF_COH_WINDOW) &
                                  (__builtin_fabsf(ux * uy0 - uy * ux0) <= HF_COH_DIR * __builtin_fabsf(ux * uy0));
                const bool coherent = !LEAN && __ballot(alive && !near) == 0ull; // (LEAN: every wave walks per lane)
                // incoherent wave: the shared walk degenerates to handing the root to every live lane
                TSTAMP(0); // ray set-up, clip and coherence test
                // Instruction-issue priority (round 4): a wave that traverses issues ahead of the waves of its SIMD that store
                // records or stream misses -- those wait for memory most of the time, the traversal is a chain of dependent
                // loads and ~100-instruction visits.  Closest hit -2.5 %, any hit -1.3 %, fused -1 %, bounce rays -1.5 %
                // (profiles/r04_ab/r04_prio; levels 1 / 2 / 3 and a middle level for the output phase measure the same).
                __builtin_amdgcn_s_setprio(HF_PRIO);
                // (one root-walk site for both the incoherent wave and a coherent wave whose beam sweep gives up)
                float thi = alive ? rs.thi : -1.f; // dead lanes overlap nothing
                bool root = alive;
                wstats_reset();
                if (coherent && f.top > HF_SUBTREE_LEVEL)
                    root = !walk_beam<MODE == 1>(f, rs, alive, fx0, fy0, best, thi, &s_beam[threadIdx.x >> 6], &s_items[threadIdx.x >> 6]) && alive && thi >= 0.f;
                if (root) {
                    hf_src_global src;
                    src.mip = f.mip; src.shear = f.shear; src.h = f.h; src.top = f.top; src.W = f.W;
                    const uint32_t lfxm = rs.fx ? ((1u << f.top) - 1u) : 0u, lfym = rs.fy ? ((1u << f.top) - 1u) : 0u;
                    (void) walk_subtree<MODE == 1>(f, src, rs, rs.r, rs.fx, rs.fy, lfxm, lfym, 0u, 0u, f.top, thi, best);
                }
#ifdef HF_WSTATS_ROOT // (diagnostic: the counters of batches that took the per-lane walk from the root as well)
                WSTATS_EXPORT(alive, best);
#else
                if (coherent && f.top > HF_SUBTREE_LEVEL) WSTATS_EXPORT(alive, best);
#endif
                TSTAMP(7); // fold of the item walks' hit table (and whatever the stamps above do not cover)
            }
#ifdef HF_TSTATS
            TSTATS_EXPORT(am != 0ull, valid, best);
#endif
            __builtin_amdgcn_s_setprio(HF_PRIO_OUT);
            asm volatile("" : "+s"(ka)); // the ~30 output pointers: loaded here, not before the walk
            // Request the next batch's rays BEFORE this batch's records are stored: vector-memory operations
            // complete in order, so a wave that loads after its stores waits for the store acknowledgements
            // (HBM write latency) on top of its own load latency -- that serial chain, not bandwidth, bounded
            // the rays that only stream.
            const bool more = (MODE == 2 && AUX) && ak + 1u < ka->aux_n; // further samples of this batch (AUX re-reads its rays: no request ahead)
            const size_t ubo = (MODE == 2 && AUX) ? ub + (size_t) ak * ka->aux_stride : ub; // where this sample's records go
            if (MODE == 2 && AUX) {
                o = mk3(0.f, 0.f, 0.f); d = o; maxt = 0.f;
            } else if (((sub + 64u) & 255u) != 0u && sub + 64 < grab && ub + 64 < n) { // (the next fetch decides about its own rays)
                const size_t ub2 = ub + 64, left2 = n - ub2;
                const uint32_t lo2 = lane < left2 ? lane : (uint32_t) (left2 - 1);
                const hf_rays_dev rp = load_rays(ka);
                o = mk3((rp.o[0] + ub2)[lo2], (rp.o[1] + ub2)[lo2], (rp.o[2] + ub2)[lo2]);
                d = mk3((rp.d[0] + ub2)[lo2], (rp.d[1] + ub2)[lo2], (rp.d[2] + ub2)[lo2]);
                maxt = (rp.maxt + ub2)[lo2];
            } else { // (defined on this path too, or the old values stay live across the walk)
                o = mk3(0.f, 0.f, 0.f); d = o; maxt = 0.f;
            }
            if (valid) { // (no per-lane exit from the sample loop: its control flow stays wave-uniform)
            if (MODE == 1) {
                uint8_t *hit_out = ka->hit_out;
                (hit_out + ub)[lo] = best.hit ? 1 : 0;
            } else {
                const hf_pi_dev pi = load_pi(ka);
                if (pi.t) (pi.t + ubo)[lo] = best.hit ? best.t : __builtin_inff();
                if (pi.u) (pi.u + ubo)[lo] = best.hit ? best.u : 0.f;
                if (pi.v) (pi.v + ubo)[lo] = best.hit ? best.v : 0.f;
                if (pi.prim) (pi.prim + ubo)[lo] = best.hit ? best.prim : 0u;
                if (MODE == 2) {
                    const uint32_t flags = ka->flags;
                    hf_si_store_sink out = { ka, ubo, lo, flags };
                    if (best.hit) {
                        // the origin is only needed by a hit: read again (an L2 hit) rather than held across the walk
                        const hf_rays_dev rp = load_rays(ka);
                        const v3 ow = mk3((rp.o[0] + ub)[lo], (rp.o[1] + ub)[lo], (rp.o[2] + ub)[lo]);
                        const hf_dev_field fl = load_field(&ka->f); // to_world etc.: not held across the walk
                        // every field is stored as soon as it is final: the record is never whole in registers
                        compute_si_to(fl, ow, dw, best.t, best.u, best.v, best.prim, flags, out);
                    } else { // zero-initialised record (interaction.h:479-499, 667-673)
                        const v3 z = mk3(0.f, 0.f, 0.f);
                        out.t(__builtin_inff()); out.p(z); out.boundary_test((flags & 0x40u) ? 1e8f : 0.f);
                        out.uv(0.f, 0.f); out.dp_dv(z); out.n(z); out.dp_du(z); out.sh_s(z); out.sh_t(z);
                        out.wi(neg3(dw));
                    }
                }
            }
            } // valid
            if (!more) break;
            } // samples of the batch
        }
    }
}

#ifndef HF_FLAT_GRID_CAP
#define HF_FLAT_GRID_CAP (256 * 64)
#endif
static int grid_for(size_t n, size_t cap = HF_FLAT_GRID_CAP) {
    size_t blocks = (n + HF_BLOCK - 1) / HF_BLOCK;
    // grid-stride kernels: many short blocks balance better than one resident set
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int) blocks;
}
// hf_si_kernel streams 148 bytes per ray and keeps nothing per block: one block per 256 rays (no grid stride up to
// 67 M rays) instead of 16384 blocks: 1.44 -> 1.24 ms on the bench wavefront (profiles/r03_ab/r03_sicap).  The adjoint
// (an LDS tile to clear per wave) is best at the default cap (r03_cap).
#ifndef HF_SI_GRID_CAP
#define HF_SI_GRID_CAP (256 * 1024)
#endif

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

// Rays per fetch: HF_GRAB for big wavefronts (a single counter serves ~80 fetches/us, which caps the rate of rays
// that only stream), fewer for smaller ones so that every resident wave gets about HF_FETCHES_PER_WAVE fetches: the
// launch ends when its slowest wave does, and a fetch of 256 incoherent rays is 50 us of work.  Measured with 4 / 12 /
// 24 / 48 fetches per wave (profiles/r03_ab/r03_fpw): a 4.19 M-ray wavefront (configs[2]) 0.435 / 0.393 / 0.394 /
// 0.393 ms fused, the bench's 16.5 M bounce rays 3.25 / 3.26 / 3.17 / 3.16 ms; the 67.1 M-ray wavefront takes HF_GRAB
// either way.
#ifndef HF_GRAB01
#define HF_GRAB01 512 // cap of the closest-hit / any-hit launches, which store 16 / 1 bytes per ray: 67.1 M rays in fetches of
                      // 512 instead of 256: closest hit 2.13 -> 2.10 ms, any hit 1.94 -> 1.88 ms (profiles/r03_ab/r03_g01); the fused
                      // launch loses 1 % with 512
#endif
static uint32_t hf_grab_for(size_t n, int mode) {
    // TEST HOOK (include/hf.h): HF_FORCE_GRAB=<multiple of 64> fixes the fetch size, so that small launches reach the paths
    // that depend on it -- the wide path needs fetches of 256 rays, which the rule below gives to launches of > 27 M rays
    if (const char *e = getenv("HF_FORCE_GRAB")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 64 && v <= 4096 && v % 64 == 0) return (uint32_t) v;
    }
    const size_t resident = 256 * 4 * HF_TRACE_WAVES; // waves the launch keeps on the chip (about)
#ifndef HF_FETCHES_PER_WAVE
#define HF_FETCHES_PER_WAVE 24
#endif
    size_t g = (n / (resident * HF_FETCHES_PER_WAVE) + 32) / 64 * 64; // about that many fetches per wave (rounded to whole batches)
    if (g < 64) g = 64;
    const size_t cap = mode == 2 ? (size_t) HF_GRAB : (size_t) HF_GRAB01;
    if (g > cap) g = cap;
    return (uint32_t) g;
}

size_t hf_trace_scratch_bytes(size_t n) {
    (void) n;
    return HF_SCR_BYTES;
}

void hf_launch_trace(int mode, const hf_dev_field &f, size_t n, const hf_rays_t *rays, const uint8_t *active,
                     const hf_pi_t *pi, uint8_t *hit, const hf_si_t *si, uint32_t flags, void *scratch,
                     hipStream_t stream, const hf_reparam_args *aux, bool lean) {
    if (n == 0) return;
    (void) hipMemsetAsync(scratch, 0, HF_SCR_BYTES, stream);
    hf_pi_dev p = { nullptr, nullptr, nullptr, nullptr };
    if (pi) { p.t = pi->t; p.u = pi->prim_uv[0]; p.v = pi->prim_uv[1]; p.prim = pi->prim_index; }
    hf_si_dev sd;
    memset(&sd, 0, sizeof(sd));
    if (si) sd = to_dev(si);
    const hf_rays_dev r = to_dev(rays);
    const uint32_t grab = hf_grab_for(n, mode);
    size_t waves = (n + grab - 1) / grab, blocks = (waves + 3) / 4;
    const size_t per_cu = lean ? ((aux && mode == 2) ? HF_LEAN_WAVES_AUX : HF_LEAN_WAVES) : mode == 2 ? HF_TRACE_WAVES_FUSED : HF_TRACE_WAVES;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu; // the resident set: that many workgroups per CU
    const dim3 grid((unsigned) blocks), block(HF_BLOCK);
    hf_trace_args a;
    a.f = f; a.n = n; a.rays = r; a.active = active; a.pi = p; a.hit_out = hit; a.sio = sd; a.flags = flags;
    a.counter = (unsigned long long *) scratch; a.grab = grab;
    {   // the wide path reads and writes 16 bytes per lane: every row it touches must be 16-byte aligned
        uintptr_t bits = 0;
        for (int k = 0; k < 3; ++k) bits |= (uintptr_t) r.o[k] | (uintptr_t) r.d[k];
        bits |= (uintptr_t) r.maxt | (uintptr_t) p.t | (uintptr_t) p.u | (uintptr_t) p.v | (uintptr_t) p.prim;
        if (mode == 1) bits |= (uintptr_t) hit << 2; // (4 bytes per lane)
        const float *const *rows = (const float *const *) &sd;
        for (size_t k = 0; k < sizeof(sd) / sizeof(float *); ++k) bits |= (uintptr_t) rows[k];
        a.wide = (active == nullptr && (bits & 15u) == 0u && grab % 256u == 0u) ? 1u : 0u;
#ifdef HF_NO_WIDE
        a.wide = 0u;
#endif
    }
    a.n_grabs = waves;
    a.aux_on = 0u; a.aux_k = 0u; a.aux_seed = 0u; a.aux_kappa = 1.f; a.aux_antithetic = 0; a.aux_ray_id = nullptr;
    a.aux_n = 1u; a.aux_stride = 0; a.aux_cull = 0.f;
    if (aux && mode == 2) {
        a.aux_on = 1u; a.aux_k = aux->k; a.aux_seed = aux->seed; a.aux_kappa = aux->kappa; a.aux_antithetic = aux->antithetic;
        a.aux_ray_id = aux->ray_id;
        if (aux->num > 1u) { a.aux_n = aux->num; a.aux_stride = aux->stride; }
        // cone_maybe_alive's constant: sqrt(2) ||to_object||_2 tan(theta_max), the norm bounded by sqrt(||.||_1 ||.||_inf);
        // cos(theta_max) = 1 - 13.83 / kappa (warp.h:557-566 with the sample clamped at 1e-6: log(1e-6) = -13.8155).
        // Not when si.wi -- minus the sample's direction -- is asked for: a culled batch draws no sample.
        const double cmin = 1.0 - 13.83 / (double) aux->kappa - 1e-6;
        if (cmin > 0.7 && !sd.wi[0] && !sd.wi[1] && !sd.wi[2]) {
            double n1 = 0.0, ninf = 0.0;
            for (int c = 0; c < 3; ++c) {
                double col = 0.0, row = 0.0;
                for (int rr = 0; rr < 3; ++rr) { col += fabs((double) f.to_object[4 * rr + c]); row += fabs((double) f.to_object[4 * c + rr]); }
                n1 = col > n1 ? col : n1; ninf = row > ninf ? row : ninf;
            }
            a.aux_cull = (float) (1.41422 * sqrt(n1 * ninf) * sqrt(1.0 - cmin * cmin) / cmin * 1.001);
        }
#ifdef HF_NO_AUX_CULL
        a.aux_cull = 0.f;
#endif
    }
    if (lean) {
        if (mode == 0)
            hipLaunchKernelGGL((hf_trace_kernel<0, false, true>), grid, block, 0, stream, a);
        else if (mode == 1)
            hipLaunchKernelGGL((hf_trace_kernel<1, false, true>), grid, block, 0, stream, a);
        else if (a.aux_on)
            hipLaunchKernelGGL((hf_trace_kernel<2, true, true>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((hf_trace_kernel<2, false, true>), grid, block, 0, stream, a);
    } else if (mode == 0)
        hipLaunchKernelGGL(hf_trace_kernel<0>, grid, block, 0, stream, a);
    else if (mode == 1)
        hipLaunchKernelGGL(hf_trace_kernel<1>, grid, block, 0, stream, a);
    else if (a.aux_on)
        hipLaunchKernelGGL((hf_trace_kernel<2, true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL(hf_trace_kernel<2>, grid, block, 0, stream, a);
}

// ---------------------------------------------------------------------------------
// surface interaction from (ray, pi)
// ---------------------------------------------------------------------------------
struct hf_pi_cdev {
    const float *t, *u, *v;
    const uint32_t *prim;
};

// The one kernel argument, read from the kernarg segment where it is used (as in the traversal kernel: ~45 pointers and
// the field by value held across the loop body were spilled into vector-register lanes: 134 SGPR spills).
struct hf_si_args {
    hf_dev_field f;
    size_t n;
    hf_rays_dev rays;
    hf_pi_cdev pi;
    const uint8_t *active;
    hf_si_dev sio;
    uint32_t flags;
};
// compute_si_to sink of hf_si_kernel: a field goes to memory when it is final, row pointers from the kernarg segment
struct hf_si_kernel_sink {
    const __attribute__((address_space(4))) hf_si_args *ka;
    size_t ub;
    uint32_t lo, flags;
    __device__ __forceinline__ void s1(float *p, float v) { st(p, ub, lo, v); }
    __device__ __forceinline__ void s3(float *p0, float *p1, float *p2, v3 v) { st(p0, ub, lo, v.x); st(p1, ub, lo, v.y); st(p2, ub, lo, v.z); }
    __device__ __forceinline__ void t(float v) { s1(ka->sio.t, v); }
    __device__ __forceinline__ void p(v3 v) { s3(ka->sio.p[0], ka->sio.p[1], ka->sio.p[2], v); }
    __device__ __forceinline__ void boundary_test(float v) { if (flags & 0x40u) s1(ka->sio.bt, v); }
    __device__ __forceinline__ void uv(float a, float b) { s1(ka->sio.uv[0], a); s1(ka->sio.uv[1], b); }
    __device__ __forceinline__ void dp_du(v3 v) { s3(ka->sio.dp_du[0], ka->sio.dp_du[1], ka->sio.dp_du[2], v); }
    __device__ __forceinline__ void dp_dv(v3 v) { s3(ka->sio.dp_dv[0], ka->sio.dp_dv[1], ka->sio.dp_dv[2], v); }
    __device__ __forceinline__ void n(v3 v) {
        s3(ka->sio.n[0], ka->sio.n[1], ka->sio.n[2], v);
        s3(ka->sio.sh_n[0], ka->sio.sh_n[1], ka->sio.sh_n[2], v);
    }
    __device__ __forceinline__ void sh_s(v3 v) { s3(ka->sio.sh_s[0], ka->sio.sh_s[1], ka->sio.sh_s[2], v); }
    __device__ __forceinline__ void sh_t(v3 v) { s3(ka->sio.sh_t[0], ka->sio.sh_t[1], ka->sio.sh_t[2], v); }
    __device__ __forceinline__ void wi(v3 v) { s3(ka->sio.wi[0], ka->sio.wi[1], ka->sio.wi[2], v); }
};

__global__ __launch_bounds__(HF_BLOCK) void hf_si_kernel(hf_si_args a_) {
    (void) a_;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t ub = (size_t) blockIdx.x * HF_BLOCK + (threadIdx.x & ~63u);; ub += stride) {
        const __attribute__((address_space(4))) hf_si_args *ka =
            (const __attribute__((address_space(4))) hf_si_args *) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka)); // opaque: keeps the loads that follow where they are written
        const size_t n = ka->n;
        if (ub >= n) break; // wave-uniform
        if (lane >= n - ub) continue;
        const uint32_t lo = lane;
        const v3 o = mk3((ka->rays.o[0] + ub)[lo], (ka->rays.o[1] + ub)[lo], (ka->rays.o[2] + ub)[lo]);
        const v3 d = mk3((ka->rays.d[0] + ub)[lo], (ka->rays.d[1] + ub)[lo], (ka->rays.d[2] + ub)[lo]);
        const float t = (ka->pi.t + ub)[lo];
        const uint8_t *active = ka->active;
        const bool act = (active ? ((active + ub)[lo] != 0) : true) && (t != __builtin_inff());
        const uint32_t flags = ka->flags;
        hf_si_kernel_sink out = { ka, ub, lo, flags };
        if (act) {
            const float b1 = (ka->pi.u + ub)[lo], b2 = (ka->pi.v + ub)[lo];
            const uint32_t prim = (ka->pi.prim + ub)[lo];
            const hf_dev_field f = load_field(&ka->f);
            compute_si_to(f, o, d, t, b1, b2, prim, flags, out);
        } else { // zero-initialised record (interaction.h:479-499, 667-673)
            const v3 z = mk3(0.f, 0.f, 0.f);
            out.t(__builtin_inff()); out.p(z); out.boundary_test((flags & 0x40u) ? 1e8f : 0.f);
            out.uv(0.f, 0.f); out.dp_dv(z); out.n(z); out.dp_du(z); out.sh_s(z); out.sh_t(z);
            out.wi(neg3(d));
        }
    }
}

void hf_launch_si(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                  const uint8_t *active, const hf_si_t *si, uint32_t flags, hipStream_t stream) {
    if (n == 0) return;
    hf_si_args a;
    a.f = f; a.n = n; a.rays = to_dev(rays);
    a.pi.t = pi->t; a.pi.u = pi->prim_uv[0]; a.pi.v = pi->prim_uv[1]; a.pi.prim = pi->prim_index;
    a.active = active; a.sio = to_dev(si); a.flags = flags;
    hipLaunchKernelGGL(hf_si_kernel, dim3(grid_for(n, HF_SI_GRID_CAP)), dim3(HF_BLOCK), 0, stream, a);
}

// ---------------------------------------------------------------------------------
// adjoint: reverse mode of compute_si, atomic scatter of dL/dheight
// ---------------------------------------------------------------------------------
#define HF_ADJ_TILE 32 // texels per side of the per-wave LDS accumulation tile

// bitwise OR over the 64 lanes of a wave (DPP within rows of 16, readlane across the four rows)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    int x = (int) v;
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); // row_half_mirror
    x |= __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); // row_mirror
    return (uint32_t) (__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) |
                       __builtin_amdgcn_readlane(x, 32) | __builtin_amdgcn_readlane(x, 48));
}

// harmonic weight of one auxiliary sample and its gradient w.r.t. the sampled direction (reparam.py:103-121)
__device__ __forceinline__ void reparam_weight(const hf_reparam_args &a, const hf_aux_sample &q, v3 d, float B,
                                               float &w, v3 &dw) {
    const float inv_vmf = 1.0f / __builtin_fmaf(q.sy, expf(-2.f * a.kappa), 1.f - q.sy);
    const float w_denom = inv_vmf - 1.f + B;
    const float w_rcp = (w_denom > 1e-4f) ? 1.0f / w_denom : 0.f;
    // (the exponent of the reference's callers is 3: three multiplications instead of exp2(y log2 x))
    w = (a.exponent == 3.f ? (w_rcp * w_rcp) * w_rcp : powf(w_rcp, a.exponent)) * inv_vmf;
    const float tmp1 = fminf(fmaxf(inv_vmf * w * w_rcp * a.kappa * a.exponent, -1e10f), 1e10f);
    const v3 tmp2 = frame_to_world(q, d, mk3(q.omega.x, q.omega.y, 0.f));
    dw = mk3(tmp1 * tmp2.x, tmp1 * tmp2.y, tmp1 * tmp2.z);
}
// gradient w.r.t. this sample's V_direct: backward of direction = normalize(d + V/Z),
// divergence = (div - <V/Z, dZ>) / Z at V = 0 (reparam.py:262-281); V_i = w V_direct, div_i = <d_w_omega, V_direct>
__device__ __forceinline__ v3 reparam_grad_vdirect(const hf_reparam_args &a, size_t i, v3 d, float w, v3 dw) {
    const float Z = fmaxf(a.Z[i], 1e-8f), iZ = 1.0f / Z;
    const v3 gd = mk3(a.g_dir[0][i], a.g_dir[1][i], a.g_dir[2][i]);
    const float gdiv = a.g_div[i];
    const float dd = dot3(d, d), idn = 1.0f / __builtin_sqrtf(dd);
    const float pr = dot3(d, gd) / dd;
    const v3 dZ = mk3(a.dZ[0][i], a.dZ[1][i], a.dZ[2][i]);
    const float c = gdiv * iZ * iZ;
    const v3 gV = mk3((gd.x - d.x * pr) * idn * iZ - c * dZ.x, (gd.y - d.y * pr) * idn * iZ - c * dZ.y,
                      (gd.z - d.z * pr) * idn * iZ - c * dZ.z);
    const float gdivV = gdiv * iZ;
    return mk3(__builtin_fmaf(w, gV.x, gdivV * dw.x), __builtin_fmaf(w, gV.y, gdivV * dw.y),
               __builtin_fmaf(w, gV.z, gdivV * dw.z));
}

struct hf_grad_dev {
    const float *t, *p[3], *n[3], *uv[2], *sh_n[3], *dp_du[3], *dp_dv[3];
};
__device__ __forceinline__ float ld(const float *p, size_t i) { return p ? p[i] : 0.f; }
__device__ __forceinline__ v3 ld3(const float *const p[3], size_t i) { return mk3(ld(p[0], i), ld(p[1], i), ld(p[2], i)); }

// The one kernel argument.  ~45 pointers and the field by value do not fit the scalar register file: held across the
// loop body they were spilled into vector-register lanes (108 SGPR spills, 342 v_readlane / v_writelane).  As in the
// traversal kernel, everything is read from the kernarg segment where it is used (scalar loads that hit the constant
// cache), and every array is addressed as (pointer + wave-uniform element offset)[lane].
struct hf_adjoint_args {
    hf_dev_field f;
    size_t n;
    hf_rays_dev rays;
    hf_pi_cdev pi;
    const uint8_t *active;
    hf_grad_dev g;
    uint32_t flags;
    float *grad_h;
    float *go[3], *gd[3];
    uint32_t *row_band; // optional: {lowest texture row that received a contribution, highest + 1}, atomicMin / atomicMax
};
typedef const __attribute__((address_space(4))) hf_adjoint_args *hf_adj_kargs;
__device__ __forceinline__ hf_adj_kargs adj_kargs() {
    hf_adj_kargs ka = (hf_adj_kargs) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka)); // opaque: keeps the loads that follow where they are written
    return ka;
}
// (row + ub)[lo], 0 for an absent row: scalar base + 32-bit lane offset
__device__ __forceinline__ float ldu(const float *p, size_t ub, uint32_t lo) { return p ? (p + ub)[lo] : 0.f; }

// RAYGRAD: dL/do and dL/dd are wanted (hf_adjoint's grad_o / grad_d); without them their accumulation is dead code
template <bool RAYGRAD>
__global__ __launch_bounds__(HF_BLOCK, 5) void hf_adjoint_kernel(hf_adjoint_args a_) {
    (void) a_;
    // Wave-level pre-reduction of the scatter: the hits of one wave (one pixel's samples for
    // primary rays) fall on a few dozen vertices, so their three contributions each are first
    // summed into a 32x32-texel LDS tile anchored near the wave's first hit (ds_add_f32) and the
    // tile is then flushed row by row -- contiguous segments, one global atomic per touched texel
    // instead of three per ray.  Contributions outside the tile go straight to global memory.
    __shared__ float s_acc[HF_BLOCK / 64][HF_ADJ_TILE * HF_ADJ_TILE];
    float *acc = s_acc[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = lane; k < HF_ADJ_TILE * HF_ADJ_TILE; k += 64) acc[k] = 0.f;
    uint32_t row_lo = 0xFFFFFFFFu, row_hi = 0u; // rows this lane scattered to (hf_adjoint_rows)
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t ub = (size_t) blockIdx.x * HF_BLOCK + (threadIdx.x & ~63u);; ub += stride) { // whole waves stay in the loop (ballots below)
        hf_adj_kargs ka = adj_kargs();
        const size_t n = ka->n;
        if (ub >= n) break; // wave-uniform
        const size_t left = n - ub;
        const bool valid = lane < left;
        const uint32_t lo = valid ? lane : (uint32_t) (left - 1);
        const float t_in = (ka->pi.t + ub)[lo];
        const uint8_t *active = ka->active;
        const bool act = valid && (active ? ((active + ub)[lo] != 0) : true) && (t_in != __builtin_inff());
        const uint32_t flags = ka->flags;
        const bool follow = (flags & 0x80u) != 0, detach = (flags & 0x100u) != 0;
        const bool tex = (flags & (0x2u | 0x4u)) != 0;
        v3 go = mk3(0.f, 0.f, 0.f), gd = mk3(0.f, 0.f, 0.f);
        float gh[3] = { 0.f, 0.f, 0.f };
        int vr[3] = { 0, 0, 0 }, vc[3] = { 0, 0, 0 };
        bool scatter = false;
        if (act) {
            // ONE batch of requests for everything a hit needs that does not depend on other loads -- the ray, the
            // rest of pi, the 18 upstream rows -- then the three heights behind prim_index: three dependent round
            // trips per iteration (pi.t, this batch, the heights) where the loads used to trail the arithmetic (six).
            const v3 o = mk3((ka->rays.o[0] + ub)[lo], (ka->rays.o[1] + ub)[lo], (ka->rays.o[2] + ub)[lo]);
            const v3 d = mk3((ka->rays.d[0] + ub)[lo], (ka->rays.d[1] + ub)[lo], (ka->rays.d[2] + ub)[lo]);
            const float b1 = (ka->pi.u + ub)[lo], b2 = (ka->pi.v + ub)[lo], b0 = 1.f - b1 - b2;
            const uint32_t prim = (ka->pi.prim + ub)[lo];
            const float gt = ldu(ka->g.t, ub, lo);
            v3 gp = mk3(ldu(ka->g.p[0], ub, lo), ldu(ka->g.p[1], ub, lo), ldu(ka->g.p[2], ub, lo));
            const v3 gn_a = mk3(ldu(ka->g.n[0], ub, lo), ldu(ka->g.n[1], ub, lo), ldu(ka->g.n[2], ub, lo));
            const v3 gn_b = mk3(ldu(ka->g.sh_n[0], ub, lo), ldu(ka->g.sh_n[1], ub, lo), ldu(ka->g.sh_n[2], ub, lo));
            const float guv0 = ldu(ka->g.uv[0], ub, lo), guv1 = ldu(ka->g.uv[1], ub, lo);
            v3 gu_ = mk3(0.f, 0.f, 0.f), gv_ = gu_;
            if (flags & 0x4u) {
                gu_ = mk3(ldu(ka->g.dp_du[0], ub, lo), ldu(ka->g.dp_du[1], ub, lo), ldu(ka->g.dp_du[2], ub, lo));
                gv_ = mk3(ldu(ka->g.dp_dv[0], ub, lo), ldu(ka->g.dp_dv[1], ub, lo), ldu(ka->g.dp_dv[2], ub, lo));
            }
            const hf_dev_field f = load_field(&ka->f);
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

            // dp_du / dp_dv from the (constant) texcoord differences
            if (flags & 0x4u) {
                const float du0 = U[1] - U[0], dv0 = V[1] - V[0], du1 = U[2] - U[0], dv1 = V[2] - V[0];
                const float det = __builtin_fmaf(du0, dv1, -(dv0 * du1));
                const float inv_det = rcp_ieee(det);
                if (det != 0.f) {
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
                const v3 gn = mk3(sgn * (gn_a.x + gn_b.x), sgn * (gn_a.y + gn_b.y), sgn * (gn_a.z + gn_b.z));
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

            if (!detach && ka->grad_h) { // dP_k/dh_k = s * (third column of to_world)
                const v3 ez = mk3(f.to_world[2], f.to_world[6], f.to_world[10]);
                gh[0] = f.s * dot3(ez, gP0); gh[1] = f.s * dot3(ez, gP1); gh[2] = f.s * dot3(ez, gP2);
                vr[0] = vi[0]; vr[1] = vi[1]; vr[2] = vi[2];
                vc[0] = vj[0]; vc[1] = vj[1]; vc[2] = vj[2];
                scatter = true;
                row_lo = min(row_lo, (uint32_t) min(vi[0], min(vi[1], vi[2])));
                row_hi = max(row_hi, (uint32_t) max(vi[0], max(vi[1], vi[2])) + 1u);
            }
        }
        const uint64_t sm = __ballot(scatter);
        if (sm != 0ull) {
            float *grad_h = adj_kargs()->grad_h;
            const int W = adj_kargs()->f.W;
            // tile anchor from the first scattering lane (wave-uniform)
            const int src = __builtin_ctzll(sm);
            const int ar = __shfl(vr[0], src) - HF_ADJ_TILE / 4, ac = __shfl(vc[0], src) - HF_ADJ_TILE / 4;
            uint32_t rows = 0u; // tile rows this lane added to
            if (scatter) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int rr = vr[k] - ar, cc = vc[k] - ac;
                    if ((unsigned) rr < (unsigned) HF_ADJ_TILE && (unsigned) cc < (unsigned) HF_ADJ_TILE) {
                        atomicAdd(acc + rr * HF_ADJ_TILE + cc, gh[k]);
                        rows |= 1u << rr;
                    } else {
                        atomicAdd(grad_h + (size_t) vr[k] * W + vc[k], gh[k]);
                    }
                }
            }
            rows = wave_or(rows);
            // flush the touched rows: 64 consecutive tile entries (two 32-texel row segments) per wave-instruction
#pragma unroll 1
            while (rows != 0u) { // wave-uniform
                const int k2 = __builtin_ctz(rows) >> 1;
                rows &= ~(3u << (2 * k2));
                const int k = k2 * 64 + (int) lane;
                const float v = acc[k];
                if (v != 0.f) {
                    const int rr = ar + k / HF_ADJ_TILE, cc = ac + k % HF_ADJ_TILE;
                    atomicAdd(grad_h + (size_t) rr * W + cc, v);
                    acc[k] = 0.f;
                }
            }
        }
        if (RAYGRAD && valid) {
            hf_adj_kargs kb = adj_kargs();
            if (kb->go[0]) { (kb->go[0] + ub)[lo] = go.x; (kb->go[1] + ub)[lo] = go.y; (kb->go[2] + ub)[lo] = go.z; }
            if (kb->gd[0]) { (kb->gd[0] + ub)[lo] = gd.x; (kb->gd[1] + ub)[lo] = gd.y; (kb->gd[2] + ub)[lo] = gd.z; }
        }
    }
    uint32_t *band = adj_kargs()->row_band;
    if (band) {
        // Per wave that scattered at all -- and only when it WIDENS the band: 65 k waves doing two atomics each on the
        // same two words cost the launch 0.9 ms (same-address atomics serialise); a relaxed read first (a stale value
        // only costs a redundant atomic) leaves the handful of waves that see the band grow.
        const uint32_t wl = wave_min_u32(row_lo), wh = wave_max_u32(row_hi);
        if (lane == 0u && wl < wh) {
            if (wl < __hip_atomic_load(band, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(band, wl);
            if (wh > __hip_atomic_load(band + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(band + 1, wh);
        }
    }
}

void hf_launch_adjoint(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                       const uint8_t *active, const hf_si_grad_t *gs, uint32_t flags, float *grad_h,
                       float *const grad_o[3], float *const grad_d[3], uint32_t *row_band, hipStream_t stream) {
    if (n == 0) return;
    const hf_pi_cdev p = { pi->t, pi->prim_uv[0], pi->prim_uv[1], pi->prim_index };
    hf_grad_dev g;
    g.t = gs->t;
    for (int k = 0; k < 3; ++k) {
        g.p[k] = gs->p[k]; g.n[k] = gs->n[k]; g.sh_n[k] = gs->sh_n[k];
        g.dp_du[k] = gs->dp_du[k]; g.dp_dv[k] = gs->dp_dv[k];
    }
    g.uv[0] = gs->uv[0]; g.uv[1] = gs->uv[1];
    hf_adjoint_args a;
    a.f = f; a.n = n; a.rays = to_dev(rays); a.pi = p; a.active = active; a.g = g; a.flags = flags; a.grad_h = grad_h;
    a.row_band = row_band;
    for (int k = 0; k < 3; ++k) { a.go[k] = grad_o ? grad_o[k] : nullptr; a.gd[k] = grad_d ? grad_d[k] : nullptr; }
    if (grad_o || grad_d) hipLaunchKernelGGL(hf_adjoint_kernel<true>, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, a);
    else                  hipLaunchKernelGGL(hf_adjoint_kernel<false>, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, a);
}

// ---------------------------------------------------------------------------------
// Backward of the warped-area reparameterisation with respect to the heights, all auxiliary samples of a ray in ONE
// pass (reparam.py:224-333 for the shape parameter): the weights of the samples and their sums Z, dZ (first loop,
// :236-256), then for every auxiliary HIT the gradient of its V_direct = (si.p - o) / si.t through the FollowShape
// surface interaction to the three heights of the hit triangle (third loop, :296-325).  Nothing but the auxiliary
// hits (pi + si.boundary_test per sample) is read: the auxiliary direction is regenerated from (d, k, seed), si.p and
// the FollowShape si.t are re-derived from pi with compute_si's expressions.  A ray none of whose samples hit does
// not reach the heights and is skipped after its num_rays reads of pi.t.
// Sample k of ray i sits at [k * stride + i] of every per-sample array.
// ---------------------------------------------------------------------------------
struct hf_reparam_bwd_args {
    size_t n, stride;
    const float *o[3], *d[3];
    const uint8_t *active;
    uint32_t num_rays, seed;
    float kappa, exponent;
    int antithetic;
    const uint32_t *ray_id;
    const float *pi_t, *pi_u, *pi_v, *si_bt;
    const uint32_t *pi_prim;
    const float *g_dir[3], *g_div;
    float *grad_h;
};

#ifndef HF_RB_KEEP
#define HF_RB_KEEP 4u // samples whose direction and weight the first loop of hf_reparam_backward_kernel keeps for the third
#endif
#ifndef HF_RB_ANCHOR
#define HF_RB_ANCHOR (HF_RB_TILE * 7 / 16) // the tile starts this many texels before the first hit of the batch (rows and columns): 8 / 16 /
                                          // 24 / 32 of 64 measured 21.4 / 20.8 / 20.5 / 20.5 ms for the backward with 4 samples (profiles/r04_ab/r04_rb)
#endif
#ifndef HF_RB_TILE
#define HF_RB_TILE 64 // the auxiliary hits of a pixel spread over tens of cells: a larger tile than hf_adjoint_kernel's
#endif
// (one argument struct, read from the kernarg segment where it is used: see hf_adjoint_kernel)
struct hf_reparam_bwd_kargs {
    hf_dev_field f;
    hf_reparam_bwd_args a;
};
__global__ __launch_bounds__(HF_BLOCK, (HF_RB_TILE > 32 ? 2 : 5)) void hf_reparam_backward_kernel(hf_reparam_bwd_kargs k_) {
    (void) k_;
    typedef const __attribute__((address_space(4))) hf_reparam_bwd_kargs *kargs_t;
    kargs_t kc = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kc));
    const size_t a_n = kc->a.n;
    __shared__ float s_acc[HF_BLOCK / 64][HF_RB_TILE * HF_RB_TILE]; // per-wave accumulation tile, as in hf_adjoint_kernel
    float *acc = s_acc[threadIdx.x >> 6];
    const int lane = (int) (threadIdx.x & 63u);
    for (int k = lane; k < HF_RB_TILE * HF_RB_TILE; k += 64) acc[k] = 0.f;
    hf_reparam_args sa = {}; // what the sampling helpers read
    sa.seed = kc->a.seed; sa.kappa = kc->a.kappa; sa.exponent = kc->a.exponent; sa.antithetic = kc->a.antithetic;
    sa.ray_id = kc->a.ray_id;
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    const size_t n_round = (a_n + HF_BLOCK - 1) / HF_BLOCK * HF_BLOCK; // whole waves stay in the loop (ballots below)
    for (size_t i_raw = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i_raw < n_round; i_raw += stride) {
        kargs_t ka = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka)); // opaque: keeps the loads that follow inside the loop body
        const bool valid = i_raw < a_n;
        const size_t i = valid ? i_raw : a_n - 1;
        const bool act = valid && (ka->a.active ? (ka->a.active[i] != 0) : true);
        uint32_t hm = 0u; // samples that hit
        for (uint32_t k = 0; k < ka->a.num_rays; ++k)
            hm |= (act && ka->a.pi_t[k * ka->a.stride + i] != __builtin_inff()) ? (1u << k) : 0u;
        if (__ballot(hm != 0u) == 0ull) continue; // wave-uniform
        v3 o = mk3(0.f, 0.f, 0.f), d = o, gV = o;
        float gdivV = 0.f;
        v3 kept_om[HF_RB_KEEP], kept_dw[HF_RB_KEEP];
        float kept_w[HF_RB_KEEP];
#pragma unroll
        for (uint32_t c = 0; c < HF_RB_KEEP; ++c) { kept_om[c] = o; kept_dw[c] = o; kept_w[c] = 0.f; }
        if (hm != 0u) {
            o = mk3(ka->a.o[0][i], ka->a.o[1][i], ka->a.o[2][i]);
            d = mk3(ka->a.d[0][i], ka->a.d[1][i], ka->a.d[2][i]);
            // first loop: Z = sum_k w_k, dZ = sum_k d_w_omega_k, in sample order from zero.  The first HF_RB_KEEP samples'
            // direction, weight and weight gradient are kept for the third loop (registers are free here: the tile's LDS
            // limits the kernel to two waves per SIMD), the others are drawn again there
            float Zs = 0.f;
            v3 dZ = mk3(0.f, 0.f, 0.f);
            for (uint32_t k = 0; k < ka->a.num_rays; ++k) {
                sa.k = k;
                hf_aux_sample q;
                aux_sample(sa, i, d, q);
                const float B = ((hm >> k) & 1u) ? ka->a.si_bt[k * ka->a.stride + i] : 1.0f;
                float w;
                v3 dw;
                reparam_weight(sa, q, d, B, w, dw);
                Zs += w;
                dZ.x += dw.x; dZ.y += dw.y; dZ.z += dw.z;
#pragma unroll
                for (uint32_t c = 0; c < HF_RB_KEEP; ++c)
                    if (k == c) { kept_om[c] = q.omega; kept_w[c] = w; kept_dw[c] = dw; } // (k is wave-uniform)
            }
            // the part of reparam_grad_vdirect that is common to the samples of a ray
            const float Z = fmaxf(Zs, 1e-8f), iZ = 1.0f / Z;
            const v3 gd = mk3(ka->a.g_dir[0][i], ka->a.g_dir[1][i], ka->a.g_dir[2][i]);
            const float gdiv = ka->a.g_div[i];
            const float dd = dot3(d, d), idn = 1.0f / __builtin_sqrtf(dd);
            const float pr = dot3(d, gd) / dd;
            const float c = gdiv * iZ * iZ;
            gV = mk3((gd.x - d.x * pr) * idn * iZ - c * dZ.x, (gd.y - d.y * pr) * idn * iZ - c * dZ.y,
                     (gd.z - d.z * pr) * idn * iZ - c * dZ.z);
            gdivV = gdiv * iZ;
        }
        // third loop: the auxiliary hits, sample by sample (wave-uniform loop, lanes with a hit take part); all
        // samples of the batch go into the tile before it is flushed
        int ar = 0, ac = 0;   // tile anchor (wave-uniform), set at the first sample with a hit
        bool anchored = false;
        uint64_t rows = 0ull;
        const uint32_t hm_any = wave_or(hm);
        for (uint32_t k = 0; k < ka->a.num_rays; ++k) {
            if (((hm_any >> k) & 1u) == 0u) continue; // wave-uniform
            const bool hit = ((hm >> k) & 1u) != 0u;
            float gh[3] = { 0.f, 0.f, 0.f };
            int vr[3] = { 0, 0, 0 }, vc[3] = { 0, 0, 0 };
            if (hit) {
                sa.k = k;
                hf_aux_sample q;
                float w;
                v3 dw;
                if (k < HF_RB_KEEP) { // wave-uniform: the values of the first loop (the same expressions on the same inputs)
                    q.omega = kept_om[0]; w = kept_w[0]; dw = kept_dw[0];
#pragma unroll
                    for (uint32_t c = 1; c < HF_RB_KEEP; ++c)
                        if (k == c) { q.omega = kept_om[c]; w = kept_w[c]; dw = kept_dw[c]; }
                    q.sy = 0.f;
                    coordinate_system(d, q.fs, q.ft);
                } else {
                    aux_sample(sa, i, d, q);
                    reparam_weight(sa, q, d, ka->a.si_bt[k * ka->a.stride + i], w, dw);
                }
                const v3 gVd = mk3(__builtin_fmaf(w, gV.x, gdivV * dw.x), __builtin_fmaf(w, gV.y, gdivV * dw.y),
                                   __builtin_fmaf(w, gV.z, gdivV * dw.z));
                const v3 da = frame_to_world(q, d, q.omega); // the auxiliary direction (= hf_reparam_aux_kernel's)
                const float b1 = ka->a.pi_u[k * ka->a.stride + i], b2 = ka->a.pi_v[k * ka->a.stride + i], b0 = 1.f - b1 - b2;
                v3 P[3];
                float U[3], V[3];
                int vi[3], vj[3];
                prim_world(load_field(&ka->f), ka->a.pi_prim[k * ka->a.stride + i], P, U, V, vi, vj);
                const v3 p = mk3(__builtin_fmaf(P[0].x, b0, __builtin_fmaf(P[1].x, b1, P[2].x * b2)),
                                 __builtin_fmaf(P[0].y, b0, __builtin_fmaf(P[1].y, b1, P[2].y * b2)),
                                 __builtin_fmaf(P[0].z, b0, __builtin_fmaf(P[1].z, b1, P[2].z * b2)));
                // V_direct = (p - o) / t with the FollowShape t = sqrt(|p - o|^2 / |d_aux|^2) of compute_si
                const v3 po = p - o;
                const float dda = dot3(da, da), tt = __builtin_sqrtf(dot3(po, po) / dda), it = 1.0f / tt;
                v3 gp = mk3(gVd.x * it, gVd.y * it, gVd.z * it);
                const float gt = -dot3(gVd, po) * it * it;
                axpy3(gt / (tt * dda), po, gp); // t's dependence on p (hf_adjoint_kernel, FollowShape branch)
                // p = sum b_k P_k with detached barycentrics; dP_k/dh_k = s * (third column of to_world)
                const v3 ez = mk3(ka->f.to_world[2], ka->f.to_world[6], ka->f.to_world[10]);
                const v3 z3 = mk3(0.f, 0.f, 0.f);
                v3 gP0 = z3, gP1 = z3, gP2 = z3;
                axpy3(b0, gp, gP0); axpy3(b1, gp, gP1); axpy3(b2, gp, gP2);
                const float fs = ka->f.s;
                gh[0] = fs * dot3(ez, gP0); gh[1] = fs * dot3(ez, gP1); gh[2] = fs * dot3(ez, gP2);
                vr[0] = vi[0]; vr[1] = vi[1]; vr[2] = vi[2];
                vc[0] = vj[0]; vc[1] = vj[1]; vc[2] = vj[2];
            }
            if (!anchored) { // wave-uniform: the first sample with a hit anchors the tile
                const int src = __builtin_ctzll(__ballot(hit));
                ar = __shfl(vr[0], src) - HF_RB_ANCHOR; ac = __shfl(vc[0], src) - HF_RB_ANCHOR;
                anchored = true;
            }
            if (hit) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int rr = vr[c] - ar, cc = vc[c] - ac;
                    if ((unsigned) rr < (unsigned) HF_RB_TILE && (unsigned) cc < (unsigned) HF_RB_TILE) {
                        atomicAdd(acc + rr * HF_RB_TILE + cc, gh[c]);
                        rows |= 1ull << rr;
                    } else {
                        atomicAdd(ka->a.grad_h + (size_t) vr[c] * ka->f.W + vc[c], gh[c]);
                    }
                }
            }
        }
        rows = (uint64_t) wave_or((uint32_t) rows) | ((uint64_t) wave_or((uint32_t) (rows >> 32)) << 32);
#pragma unroll 1
        while (rows != 0ull) { // flush the touched rows: 64 consecutive tile entries per wave-instruction
            const int rr = __builtin_ctzll(rows);
            rows &= rows - 1ull;
            for (int c0 = 0; c0 < HF_RB_TILE; c0 += 64) {
                const int cc = c0 + lane;
                if (cc < HF_RB_TILE) {
                    const float v = acc[rr * HF_RB_TILE + cc];
                    if (v != 0.f) {
                        atomicAdd(ka->a.grad_h + (size_t) (ar + rr) * ka->f.W + (ac + cc), v);
                        acc[rr * HF_RB_TILE + cc] = 0.f;
                    }
                }
            }
        }
    }
}

void hf_launch_reparam_backward(const hf_dev_field &f, const hf_reparam_args &ra, uint32_t num_rays, size_t stride,
                                const hf_pi_const_t *pi, float *grad_h, hipStream_t stream) {
    if (ra.n == 0) return;
    hf_reparam_bwd_args a = {};
    a.n = ra.n; a.stride = stride; a.active = ra.active; a.num_rays = num_rays; a.seed = ra.seed; a.kappa = ra.kappa;
    a.exponent = ra.exponent; a.antithetic = ra.antithetic; a.ray_id = ra.ray_id;
    for (int c = 0; c < 3; ++c) { a.o[c] = ra.o[c]; a.d[c] = ra.d[c]; a.g_dir[c] = ra.g_dir[c]; }
    a.g_div = ra.g_div; a.si_bt = ra.si_bt;
    a.pi_t = pi->t; a.pi_u = pi->prim_uv[0]; a.pi_v = pi->prim_uv[1]; a.pi_prim = pi->prim_index;
    a.grad_h = grad_h;
    hf_reparam_bwd_kargs k;
    k.f = f; k.a = a;
    hipLaunchKernelGGL(hf_reparam_backward_kernel, dim3(grid_for(ra.n)), dim3(HF_BLOCK), 0, stream, k);
}

// ---------------------------------------------------------------------------------
// Adam step on the height texture (optimizers.py:263-300), explicit operation order (no contraction)
// ---------------------------------------------------------------------------------
// (sched, ctr): hf_adam_step_scheduled -- the step size is sched[*ctr] (a host-filled table of the bias-corrected step
// sizes, so that a captured step can be replayed: the step number is not baked into the launch); else lr_t
__global__ __launch_bounds__(HF_BLOCK) void hf_adam_kernel(size_t n, float *__restrict__ h, const float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v, float lr_t,
                                                          float beta1, float beta2, float c1, float c2, float eps,
                                                          int mask_updates, const float *__restrict__ sched,
                                                          const uint32_t *__restrict__ ctr) {
    if (sched) lr_t = sched[*ctr];
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        if (mask_updates && gi == 0.f) continue;
        const float mt = beta1 * m[i] + c1 * gi;
        const float vt = beta2 * v[i] + c2 * (gi * gi);
        m[i] = mt; v[i] = vt;
        h[i] = h[i] - (lr_t * mt) / (__builtin_sqrtf(vt) + eps);
    }
}

// 'UniformAdam' (optimizers.py:259, 290-291: the update divides by the square root of the MAXIMUM second moment of the
// step instead of the per-element one): pass 1 updates the moments and reduces max(v) into *vmax (float bits as an
// unsigned: v >= 0), pass 2 applies  h -= lr_t m / (sqrt(max v) + eps).
__global__ __launch_bounds__(HF_BLOCK) void hf_adam_moments_kernel(size_t n, const float *__restrict__ g, float *__restrict__ m,
                                                                  float *__restrict__ v, float beta1, float beta2, float c1,
                                                                  float c2, int mask_updates, uint32_t *vmax) {
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    float mx = 0.f;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        float vt = v[i];
        if (!(mask_updates && gi == 0.f)) {
            m[i] = beta1 * m[i] + c1 * gi;
            vt = beta2 * vt + c2 * (gi * gi);
            v[i] = vt;
        }
        mx = (vt != vt) ? vt : fmaxf(mx, vt); // dr.max(v_t) runs over every entry, masked ones with their old value (:282-285, 290); a NaN moment is kept (fmaxf would drop it and mask a diverged run)
    }
    // (as unsigned bits: v >= 0, and a NaN -- 0x7fc00000 and up -- is larger than every finite value, so it reaches *vmax)
    const uint32_t wm = wave_max_u32(__builtin_bit_cast(uint32_t, mx) & 0x7FFFFFFFu);
    if ((threadIdx.x & 63u) == 0u && wm > __hip_atomic_load(vmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(vmax, wm);
}
__global__ __launch_bounds__(HF_BLOCK) void hf_adam_apply_uniform_kernel(size_t n, float *__restrict__ h, const float *__restrict__ g,
                                                                        const float *__restrict__ m, float lr_t, float eps,
                                                                        int mask_updates, const uint32_t *vmax,
                                                                        const float *__restrict__ sched, const uint32_t *__restrict__ ctr) {
    if (sched) lr_t = sched[*ctr];
    const float den = __builtin_sqrtf(__builtin_bit_cast(float, *vmax)) + eps;
    const size_t stride = (size_t) gridDim.x * HF_BLOCK;
    for (size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x; i < n; i += stride) {
        if (mask_updates && g[i] == 0.f) continue;
        h[i] = h[i] - (lr_t * m[i]) / den;
    }
}

__global__ void hf_counter_increment_kernel(uint32_t *ctr) { *ctr += 1u; }

hipError_t hf_launch_adam(size_t n, float *h, const float *g, float *m, float *v, float lr_t, float beta1, float beta2,
                          float c1, float c2, float eps, int mask_updates, hipStream_t stream, uint32_t *uniform_scratch,
                          const float *sched, uint32_t *ctr) {
    if (n == 0) return hipSuccess;
    if (uniform_scratch) {
        const hipError_t e = hipMemsetAsync(uniform_scratch, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e; // (the reduction below would start from a stale maximum)
        hipLaunchKernelGGL(hf_adam_moments_kernel, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, n, g, m, v, beta1, beta2, c1, c2,
                           mask_updates, uniform_scratch);
        hipLaunchKernelGGL(hf_adam_apply_uniform_kernel, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, n, h, g, m, lr_t, eps,
                           mask_updates, uniform_scratch, sched, (const uint32_t *) ctr);
    } else {
        hipLaunchKernelGGL(hf_adam_kernel, dim3(grid_for(n)), dim3(HF_BLOCK), 0, stream, n, h, g, m, v, lr_t, beta1, beta2,
                           c1, c2, eps, mask_updates, sched, (const uint32_t *) ctr);
    }
    if (sched) hipLaunchKernelGGL(hf_counter_increment_kernel, dim3(1), dim3(1), 0, stream, ctr); // the next replay's step
    return hipSuccess;
}

// ---------------------------------------------------------------------------------
// Minimal direct lighting on the wavefront (include/hf.h): one lane per sample, coalesced SoA loads,
// the box-filter film as a shuffle tree over the samples of a pixel.  Pure streaming: 28 B/sample in.
// ---------------------------------------------------------------------------------
#define HF_DPP_ADD(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
struct hf_f3ptr { const float *p[3]; };
struct hf_f3out { float *p[3]; };

// POINT: L.l[k] is the light's position and L.w[k] = albedo/pi * intensity; the direction towards the light and the
// inverse-square falloff are per sample (src/emitters/point.cpp: d = pos - it.p, spec = intensity / |d|^2)
template <bool POINT>
__global__ __launch_bounds__(HF_BLOCK) void hf_direct_kernel(size_t n, uint32_t spp, hf_f3ptr sn, hf_f3ptr dd,
                                                            const float *__restrict__ t, hf_f3ptr pp, hf_lights_dev L,
                                                            float *__restrict__ image) {
    const size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x;
    const size_t npix = n / spp;
    const bool in = i < n;
    const size_t ii = in ? i : n - 1;
    const v3 nn = mk3(sn.p[0][ii], sn.p[1][ii], sn.p[2][ii]);
    const v3 d = mk3(dd.p[0][ii], dd.p[1][ii], dd.p[2][ii]);
    const bool lit = in && (t[ii] != __builtin_inff()) && (-dot3(nn, d) > 0.f); // hit, seen from the front (diffuse.cpp:137)
    const bool pow2 = (spp & (spp - 1u)) == 0u;
    const uint32_t g = spp < 64u ? spp : 64u; // lanes per pixel within a wave (tree path)
    const float inv_spp = 1.0f / (float) spp;
    v3 p = mk3(0.f, 0.f, 0.f);
    if (POINT) p = mk3(pp.p[0][ii], pp.p[1][ii], pp.p[2][ii]);
    for (uint32_t k = 0; k < L.n; ++k) {
        v3 l = mk3(L.l[k][0], L.l[k][1], L.l[k][2]);
        float wk = L.w[k];
        if (POINT) {
            const v3 v = l - p;
            const float ir = 1.0f / __builtin_sqrtf(dot3(v, v));
            l = v * ir;
            wk = wk * (ir * ir);
        }
        const float co = dot3(nn, l);
        float c = (lit && co > 0.f && (L.vis[k] ? L.vis[k][ii] != 0 : true)) ? wk * co : 0.f;
        if (L.weight) c *= L.weight[ii];
        if (pow2) {
            // butterfly over the g = min(spp, 64) lanes of a pixel: DPP within rows of 16, cross-lane beyond
            if (g > 1u) c += HF_DPP_ADD(c, 0xB1);   // quad_perm [1,0,3,2]
            if (g > 2u) c += HF_DPP_ADD(c, 0x4E);   // quad_perm [2,3,0,1]
            if (g > 4u) c += HF_DPP_ADD(c, 0x141);  // row_half_mirror: the other quad pair
            if (g > 8u) c += HF_DPP_ADD(c, 0x140);  // row_mirror: the other half row
            if (g > 16u) c += __shfl_xor(c, 16);
            if (g > 32u) c += __shfl_xor(c, 32);
            if (in && (threadIdx.x & (g - 1u)) == 0u) {
                if (spp <= 64u) image[k * npix + i / spp] = c * inv_spp;          // the wave holds whole pixels
                else            atomicAdd(&image[k * npix + i / spp], c * inv_spp); // several waves per pixel
            }
        } else if (in) {
            atomicAdd(&image[k * npix + i / spp], c * inv_spp);
        }
    }
}

// POINT: f = w (n.v) |v|^-3 with v = pos - p:  df/dn = w |v|^-2 l,  df/dp = w |v|^-3 (3 (n.l) l - n)
template <bool POINT>
__global__ __launch_bounds__(HF_BLOCK) void hf_direct_adjoint_kernel(size_t n, uint32_t spp, hf_f3ptr sn, hf_f3ptr dd,
                                                                    const float *__restrict__ t, hf_f3ptr pp, hf_lights_dev L,
                                                                    const float *__restrict__ gimg, hf_f3out gn, hf_f3out gp) {
    const size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const size_t npix = n / spp, pix = i / spp;
    const v3 nn = mk3(sn.p[0][i], sn.p[1][i], sn.p[2][i]);
    const v3 d = mk3(dd.p[0][i], dd.p[1][i], dd.p[2][i]);
    const bool lit = (t[i] != __builtin_inff()) && (-dot3(nn, d) > 0.f);
    const float inv_spp = 1.0f / (float) spp;
    v3 g = mk3(0.f, 0.f, 0.f), gq = mk3(0.f, 0.f, 0.f);
    v3 p = mk3(0.f, 0.f, 0.f);
    if (POINT) p = mk3(pp.p[0][i], pp.p[1][i], pp.p[2][i]);
    const float wgt = L.weight ? L.weight[i] : 1.f;
    float gw = 0.f; // dL/dweight: the unweighted sample values against the image gradient
    for (uint32_t k = 0; k < L.n; ++k) {
        v3 l = mk3(L.l[k][0], L.l[k][1], L.l[k][2]);
        float ir = 1.f;
        if (POINT) {
            const v3 v = l - p;
            ir = 1.0f / __builtin_sqrtf(dot3(v, v));
            l = v * ir;
        }
        const float co = dot3(nn, l);
        if (lit && co > 0.f && (L.vis[k] ? L.vis[k][i] != 0 : true)) {
            float w = (L.w[k] * inv_spp) * gimg[k * npix + pix];
            if (!POINT) gw = __builtin_fmaf(w, co, gw);
            w *= wgt;
            if (POINT) {
                w = w * (ir * ir);
                const float wp = w * ir, c3 = 3.f * co;
                gq.x = __builtin_fmaf(wp, __builtin_fmaf(c3, l.x, -nn.x), gq.x);
                gq.y = __builtin_fmaf(wp, __builtin_fmaf(c3, l.y, -nn.y), gq.y);
                gq.z = __builtin_fmaf(wp, __builtin_fmaf(c3, l.z, -nn.z), gq.z);
            }
            g.x = __builtin_fmaf(w, l.x, g.x); g.y = __builtin_fmaf(w, l.y, g.y); g.z = __builtin_fmaf(w, l.z, g.z);
        }
    }
    gn.p[0][i] = g.x; gn.p[1][i] = g.y; gn.p[2][i] = g.z;
    if (POINT) { gp.p[0][i] = gq.x; gp.p[1][i] = gq.y; gp.p[2][i] = gq.z; }
    if (!POINT && L.grad_weight) L.grad_weight[i] = gw;
}

// ---------------------------------------------------------------------------------
// Gaussian reconstruction filter, splatted the way ImageBlock::put does (src/render/imageblock.cpp:258-330 with
// src/rfilters/gaussian.cpp:48-101): a sample at film position pos adds  w_x(px) w_y(py) value  to every pixel
// whose index lies in [ceil(pos - 0.5 - r), floor(pos - 0.5 + r)] (clamped to the film), with the windowed Gaussian
// w(x) = max(0, exp(-x^2 / (2 stddev^2)) - exp(-r^2 / (2 stddev^2))), r = 4 stddev, x = pixel index - (pos - 0.5);
// the accumulated weight goes to its own plane and the film divides by it.
// ---------------------------------------------------------------------------------

template <bool ADJOINT>
__global__ __launch_bounds__(HF_BLOCK) void hf_film_splat_kernel(hf_splat_args a) {
    const size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const float fx = a.pos_x[i] - 0.5f, fy = a.pos_y[i] - 0.5f;
    const int x0 = max((int) ceilf(fx - a.radius), 0), x1 = min((int) floorf(fx + a.radius), (int) a.width - 1);
    const int y0 = max((int) ceilf(fy - a.radius), 0), y1 = min((int) floorf(fy + a.radius), (int) a.height - 1);
    float acc[HF_MAX_LIGHTS];
    float val[HF_MAX_LIGHTS];
#pragma unroll
    for (uint32_t k = 0; k < HF_MAX_LIGHTS; ++k) {
        acc[k] = 0.f;
        val[k] = (!ADJOINT && k < a.channels) ? a.values[k][i] : 0.f;
    }
    const size_t plane = (size_t) a.width * a.height;
    for (int y = y0; y <= y1; ++y) {
        const float dy = (float) y - fy;
        const float wy = fmaxf(expf(a.alpha * (dy * dy)) - a.bias, 0.f);
        for (int x = x0; x <= x1; ++x) {
            const float dx = (float) x - fx;
            const float w = fmaxf(expf(a.alpha * (dx * dx)) - a.bias, 0.f) * wy;
            if (w == 0.f) continue;
            const size_t pix = (size_t) y * a.width + x;
            if (ADJOINT) {
#pragma unroll
                for (uint32_t k = 0; k < HF_MAX_LIGHTS; ++k)
                    if (k < a.channels) acc[k] = __builtin_fmaf(w, a.grad_image[k * plane + pix], acc[k]);
            } else {
                atomicAdd(a.weight + pix, w);
#pragma unroll
                for (uint32_t k = 0; k < HF_MAX_LIGHTS; ++k)
                    if (k < a.channels) atomicAdd(a.image + k * plane + pix, w * val[k]);
            }
        }
    }
    if (ADJOINT) {
#pragma unroll
        for (uint32_t k = 0; k < HF_MAX_LIGHTS; ++k)
            if (k < a.channels) a.grad_values[k][i] = acc[k];
    }
}

void hf_launch_film_splat(const hf_splat_args &a, bool adjoint, hipStream_t stream) {
    if (a.n == 0) return;
    const dim3 grid((unsigned) ((a.n + HF_BLOCK - 1) / HF_BLOCK)), block(HF_BLOCK);
    if (adjoint) hipLaunchKernelGGL(hf_film_splat_kernel<true>, grid, block, 0, stream, a);
    else         hipLaunchKernelGGL(hf_film_splat_kernel<false>, grid, block, 0, stream, a);
}

void hf_launch_direct(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3], const float *t,
                      const float *const p[3], const hf_lights_dev &lights, float *image, hipStream_t stream) {
    if (n == 0) return;
    const bool pow2 = (spp & (spp - 1u)) == 0u;
    if (!pow2 || spp > 64u) (void) hipMemsetAsync(image, 0, sizeof(float) * lights.n * (n / spp), stream); // atomic paths
    hf_f3ptr sn = { { sh_n[0], sh_n[1], sh_n[2] } }, dd = { { d[0], d[1], d[2] } }, pp = { { nullptr, nullptr, nullptr } };
    const size_t blocks = (n + HF_BLOCK - 1) / HF_BLOCK;
    if (p) { // point lights
        pp = { { p[0], p[1], p[2] } };
        hipLaunchKernelGGL(hf_direct_kernel<true>, dim3((unsigned) blocks), dim3(HF_BLOCK), 0, stream, n, spp, sn, dd, t, pp, lights, image);
    } else {
        hipLaunchKernelGGL(hf_direct_kernel<false>, dim3((unsigned) blocks), dim3(HF_BLOCK), 0, stream, n, spp, sn, dd, t, pp, lights, image);
    }
}

void hf_launch_direct_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                              const float *t, const float *const p[3], const hf_lights_dev &lights,
                              const float *grad_image, float *const grad_sh_n[3], float *const grad_p[3],
                              hipStream_t stream) {
    if (n == 0) return;
    hf_f3ptr sn = { { sh_n[0], sh_n[1], sh_n[2] } }, dd = { { d[0], d[1], d[2] } }, pp = { { nullptr, nullptr, nullptr } };
    hf_f3out gn = { { grad_sh_n[0], grad_sh_n[1], grad_sh_n[2] } }, gp = { { nullptr, nullptr, nullptr } };
    const size_t blocks = (n + HF_BLOCK - 1) / HF_BLOCK;
    if (p) {
        pp = { { p[0], p[1], p[2] } };
        gp = { { grad_p[0], grad_p[1], grad_p[2] } };
        hipLaunchKernelGGL(hf_direct_adjoint_kernel<true>, dim3((unsigned) blocks), dim3(HF_BLOCK), 0, stream, n, spp, sn, dd, t,
                           pp, lights, grad_image, gn, gp);
    } else {
        hipLaunchKernelGGL(hf_direct_adjoint_kernel<false>, dim3((unsigned) blocks), dim3(HF_BLOCK), 0, stream, n, spp, sn, dd, t,
                           pp, lights, grad_image, gn, gp);
    }
}

// ---- warped-area reparameterisation: per-sample kernels (helpers: above hf_adjoint_kernel) ----
__global__ __launch_bounds__(HF_BLOCK) void hf_reparam_aux_kernel(hf_reparam_args a) {
    const size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const v3 d = mk3(a.d[0][i], a.d[1][i], a.d[2][i]);
    hf_aux_sample q;
    aux_sample(a, i, d, q);
    const v3 ad = frame_to_world(q, d, q.omega);
    a.aux_d[0][i] = ad.x; a.aux_d[1][i] = ad.y; a.aux_d[2][i] = ad.z;
    a.aux_maxt[i] = (a.active && a.active[i] == 0) ? -1.f : __builtin_inff();
}

__global__ __launch_bounds__(HF_BLOCK) void hf_reparam_weight_kernel(hf_reparam_args a) {
    const size_t i = (size_t) blockIdx.x * HF_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const bool act = !(a.active && a.active[i] == 0);
    const v3 o = mk3(a.o[0][i], a.o[1][i], a.o[2][i]), d = mk3(a.d[0][i], a.d[1][i], a.d[2][i]);
    hf_aux_sample q;
    aux_sample(a, i, d, q);
    const float t = a.si_t[i];
    const bool hit = act && (t != __builtin_inff());
    const float B = hit ? a.si_bt[i] : 1.0f;
    float w;
    v3 dw;
    reparam_weight(a, q, d, B, w, dw);
    if (a.mode == 0) {
        if (!act) return;
        a.Z[i] += w;
        a.dZ[0][i] += dw.x; a.dZ[1][i] += dw.y; a.dZ[2][i] += dw.z;
        return;
    }
    v3 gp = mk3(0.f, 0.f, 0.f), gvd = mk3(0.f, 0.f, 0.f);
    float gt = 0.f;
    if (act) {
        const v3 gVd = reparam_grad_vdirect(a, i, d, w, dw);
        gvd = gVd;
        if (hit) { // V_direct = (p - o) / t
            const v3 po = mk3(a.si_p[0][i] - o.x, a.si_p[1][i] - o.y, a.si_p[2][i] - o.z);
            const float it = 1.0f / t;
            gp = mk3(gVd.x * it, gVd.y * it, gVd.z * it);
            gt = -dot3(gVd, po) * it * it;
        }
    }
    a.g_p[0][i] = gp.x; a.g_p[1][i] = gp.y; a.g_p[2][i] = gp.z;
    a.g_t[i] = gt;
    if (a.g_vd[0]) { a.g_vd[0][i] = gvd.x; a.g_vd[1][i] = gvd.y; a.g_vd[2][i] = gvd.z; }
}

void hf_launch_reparam_aux(const hf_reparam_args &a, hipStream_t stream) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(hf_reparam_aux_kernel, dim3((unsigned) ((a.n + HF_BLOCK - 1) / HF_BLOCK)), dim3(HF_BLOCK), 0, stream, a);
}
void hf_launch_reparam_weights(const hf_reparam_args &a, hipStream_t stream) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(hf_reparam_weight_kernel, dim3((unsigned) ((a.n + HF_BLOCK - 1) / HF_BLOCK)), dim3(HF_BLOCK), 0, stream, a);
}
