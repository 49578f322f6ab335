/*
 * hf.h -- C ABI of libhf: MI355X (gfx950) differentiable heightfield intersector.
 *
 * This is the drop-in boundary for the reference's Shape plugin interface on
 * the heightfield hot path.  A Mitsuba 3.3 `Shape`
 * subclass ("heightfield" plugin) forwards whole ray
 * wavefronts through these entry points instead of running per-lane Dr.Jit
 * code / vcalls.  Plain C types only: device pointers + sizes, no torch,
 * no Dr.Jit, no C++ types.
 *
 * Conventions
 *   - All array arguments are DEVICE pointers (HIP) to SoA component arrays
 *     of `n` elements each (Dr.Jit arrays are SoA: one array per scalar
 *     component, cf. the RayHitT offsets in src/render/scene_native.inl:106-113).
 *   - Every call is asynchronous and ordered on the caller's `stream`
 *     (a hipStream_t passed as void*; NULL = default stream).  No hidden
 *     synchronisation except where stated (hf_bbox, hf_get_mip).
 *   - `active` (uint8 per lane, NULL = all lanes active) mirrors the `Mask active`
 *     argument of the reference methods; inactive lanes produce a miss /
 *     zero-initialised record (include/mitsuba/render/interaction.h:479-499, 667-673).
 *   - Return value: HF_OK or an error code; hf_last_error_string() describes
 *     the last failure on the calling thread.  No exceptions cross the ABI;
 *     the adapter turns codes into Throw(...) (src/render/mesh.cpp:711 style).
 *   - Query functions are re-entrant on a const handle (the reference calls
 *     them concurrently from worker threads, src/render/integrator.cpp:161-200);
 *     hf_set_heights* requires that no query on the same handle is in flight
 *     on another stream (reference: dr::sync_thread() in parameters_changed,
 *     src/shapes/rectangle.cpp:131-142).
 *   - HIP graphs: the wavefront entry points (device-pointer forms), hf_set_heights and hf_adam_step may be issued
 *     on a stream that is being captured; they then allocate nothing and record / wait for no event, so one
 *     optimisation step can be captured once and replayed.  Limits, all the caller's to honour:
 *       * every captured trace launch (hf_ray_intersect*, hf_ray_test, hf_reparam_trace) reserves one of 32 scratch
 *         blocks of the handle for as long as its graph may be replayed; the 33rd is refused with HF_EINVAL.
 *         hf_capture_reset() hands the blocks back once the graphs captured so far have been destroyed;
 *       * two graphs of one handle must not be replayed concurrently with each other unless they were captured
 *         without a reset in between (distinct blocks); replays that run concurrently with other work of the same
 *         handle on other streams are ordered by the caller, as for any buffer the graph writes;
 *       * a replayed hf_set_heights / hf_adam_step does not record the handle's "built" event: after such a replay
 *         synchronise the replay's stream before hf_bbox, hf_get_mip, the packet entry points or hf_destroy;
 *       * a captured launch snapshots the transform (to_world / to_object) by value: hf_set_transform after the
 *         capture does not reach the replays -- re-capture.
 *     Not capturable: hf_create / hf_destroy, hf_set_heights_host, hf_bbox, hf_get_mip and the host-pointer packet
 *     entry (they synchronise).
 */
#ifndef HF_H
#define HF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_VERSION 4 /* 2: hf_reparam_* take ray_id; hf_adjoint_rows, weighted lighting, hf_capture_reset; 3: hf_adam_step_scheduled;
                        4: hf_set_ray_coherence */

/* status codes */
enum {
    HF_OK      = 0,
    HF_EINVAL  = 1,  /* bad argument (NULL pointer, width/height < 2: src/textures/bitmap.cpp:280-283) */
    HF_EDEVICE = 2,  /* HIP runtime error */
    HF_ENOMEM  = 3,  /* device allocation failed */
    HF_EFLAGS  = 4   /* DetachShape | FollowShape (src/render/mesh.cpp:709-711) */
};

/* RayFlags -- identical values to include/mitsuba/render/interaction.h:19-69 */
enum {
    HF_RAY_EMPTY         = 0x0,
    HF_RAY_MINIMAL       = 0x1,
    HF_RAY_UV            = 0x2,
    HF_RAY_DPDUV         = 0x4,
    HF_RAY_SHADINGFRAME  = 0x8,
    HF_RAY_DNGDUV        = 0x10,
    HF_RAY_DNSDUV        = 0x20,
    HF_RAY_BOUNDARYTEST  = 0x40,
    HF_RAY_FOLLOWSHAPE   = 0x80,
    HF_RAY_DETACHSHAPE   = 0x100,
    HF_RAY_ALL           = 0x2 | 0x4 | 0x8,
    HF_RAY_ALL_NONDIFF   = 0x2 | 0x4 | 0x8 | 0x100,
    /* libhf extension (not a Mitsuba RayFlags value; bits above the reference's): with HF_RAY_BOUNDARYTEST,
     * boundary_test is the reference Mesh's per-triangle SDF over ALL three edges of the hit triangle
     * (src/render/mesh.cpp:845-890, 0 on an edge .. 1 at the incentre) instead of this shape's default, the same
     * SDF restricted to SILHOUETTE edges (a deliberate extension for heightfields) */
    HF_RAY_BOUNDARY_ALL_EDGES = 0x10000
};

typedef struct hf_field hf_field_t; /* opaque handle: owns heights copy, min/max mips */
typedef void *hf_stream_t;          /* hipStream_t */

/* Plugin properties (build decision, SURVEY.md section 8a: {to_world, max_height,
 * heightfield tensor, flip_normals}); replaces the Properties-driven constructor +
 * update() of a shape plugin (src/shapes/rectangle.cpp:83-112). */
typedef struct hf_desc {
    uint32_t width;        /* vertices along object x (tensor columns), >= 2 */
    uint32_t height;       /* vertices along object y (tensor rows),    >= 2 */
    float    max_height;   /* object z = height value * max_height */
    float    to_world[12]; /* row-major 3x4 affine */
    float    to_object[12];/* inverse of to_world; used when has_to_object != 0
                              (the adapter passes Mitsuba's own m_to_object,
                              rectangle.cpp:102), else computed by hf_invert_affine */
    int32_t  has_to_object;
    int32_t  flip_normals;
    int32_t  device;       /* HIP device ordinal */
} hf_desc_t;

/* Ray3f (include/mitsuba/core/ray.h:24-82): o, d, maxt.  time/wavelengths are
 * not used by a static shape and stay on the caller's side.
 * ALIGNMENT: any float alignment works.  When every ray row and every output row of a launch is 16-byte aligned and
 * no `active` mask is given (what a device allocator hands out), the traversal kernels answer fetches of 256 rays
 * that miss the bound as a whole -- the part of an image beside the terrain -- through 16-byte loads and stores
 * (DESIGN 4.1 "wide path": -4 % on the BASELINE wavefront); results do not depend on it.
 * TEST HOOK: the environment variable HF_FORCE_GRAB=<64 .. 4096, a multiple of 64>, read at every trace launch, fixes
 * the number of rays a wave fetches at a time (otherwise chosen from the size of the launch); 256 lets launches of any
 * size take the wide path (tests/test_gpu_parity.py, tests/tools/fuzz_parity.py).  Results do not depend on it. */
typedef struct hf_rays {
    const float *o[3];
    const float *d[3];
    const float *maxt;
} hf_rays_t;

/* PreliminaryIntersection3f (interaction.h:587-691): t (+inf = miss), prim_uv,
 * prim_index = 2*(cell_y*(width-1)+cell_x)+tri.  shape_index is always
 * (uint32_t)-1 for a non-instanced shape (rectangle.cpp:222) and is not stored. */
typedef struct hf_pi {
    float    *t;
    float    *prim_uv[2];
    uint32_t *prim_index;
} hf_pi_t;

typedef struct hf_pi_const {
    const float    *t;
    const float    *prim_uv[2];
    const uint32_t *prim_index;
} hf_pi_const_t;

/* SurfaceInteraction3f fields filled by Shape::compute_surface_interaction +
 * finalize_surface_interaction (interaction.h:175-507).  Any pointer may be NULL
 * (field not wanted).  dn_du/dn_dv are identically zero (flat shading) and
 * duv_dx/duv_dy are zeroed by finalize; neither is stored. */
typedef struct hf_si {
    float *t;
    float *p[3];
    float *n[3];
    float *uv[2];
    float *sh_n[3];      /* sh_frame.n */
    float *dp_du[3];
    float *dp_dv[3];
    float *boundary_test;/* written only with HF_RAY_BOUNDARYTEST */
    float *sh_s[3];      /* sh_frame.s, sh_frame.t: HF_RAY_SHADINGFRAME (interaction.h:257-267) */
    float *sh_t[3];
    float *wi[3];
} hf_si_t;

/* Upstream gradient dL/d(si field); NULL pointer = zero gradient. */
typedef struct hf_si_grad {
    const float *t;
    const float *p[3];
    const float *n[3];
    const float *uv[2];
    const float *sh_n[3];
    const float *dp_du[3];
    const float *dp_dv[3];
} hf_si_grad_t;

/* ---- lifetime / parameters ------------------------------------------------ */

/* Replaces: plugin construction + update() (src/shapes/rectangle.cpp:83-112) and the
 * OptiX blob upload optix_prepare_geometry (rectangle.cpp:328-338).  Heights start
 * as all zero; call hf_set_heights* before tracing.  Limits: width, height >= 2
 * (bitmap.cpp:280-283), at most 32768 cells per side and 2^30 vertices (HF_EINVAL beyond). */
int hf_create(const hf_desc_t *desc, hf_field_t **out);
int hf_destroy(hf_field_t *hf);
/* Returns the scratch blocks reserved by captured trace launches (HIP graphs, above) to the handle.  Call it only
 * when every graph captured from this handle so far has been destroyed or will not be replayed again.  (No
 * reference counterpart: Dr.Jit owns its kernel-launch scratch; cf. jit_free, src/shapes/rectangle.cpp:330-336.) */
int hf_capture_reset(hf_field_t *hf);

/* Replaces: parameters_changed({"heightfield"}) (pattern rectangle.cpp:131-142,
 * tensor form src/textures/bitmap.cpp:272-286) + the accel rebuild it triggers
 * (src/render/scene.cpp:343-385): copies width*height floats (row-major, row 0 at
 * object y=-1) from DEVICE memory and rebuilds the acceleration data (min/max mip pyramid and the
 * sheared bounds of its fine levels), all on `stream`. */
int hf_set_heights(hf_field_t *hf, const float *d_heights, hf_stream_t stream);
/* same from HOST memory (synchronous copy) */
int hf_set_heights_host(hf_field_t *hf, const float *h_heights, hf_stream_t stream);
/* Replaces: mitsuba.ad.Adam.step() for this parameter (src/python/python/ad/optimizers.py:263-300) followed
 * by params.update() -> parameters_changed({"heightfield"}) (src/python/python/util.py:185-232): one
 * Adam step on the caller's parameter buffer d_heights[width*height] (DEVICE, updated in place) with the
 * gradient d_grad and the moment buffers d_m, d_v (zero them before step 1; `step` counts from 1), then
 * hf_set_heights(hf, d_heights).  lr_t = lr * sqrt(1 - beta2^step) / (1 - beta1^step);
 * m = beta1 m + (1-beta1) g;  v = beta2 v + (1-beta2) g^2;  h -= lr_t m / (sqrt(v) + eps);
 * mask_updates: bit 0 (HF_ADAM_MASK_UPDATES) leaves h, m, v untouched where g == 0 (optimizers.py:282-285,
 * 293-294); bit 1 (HF_ADAM_UNIFORM) selects the 'UniformAdam' variant (optimizers.py:259, 290-291): the update
 * divides by sqrt(max over the texture of v) + eps instead of the per-texel sqrt(v) + eps (two launches).
 * The hyper-parameters are host doubles like the reference's Python scalars: the bias-correction scale is
 * evaluated in double and rounded once (optimizers.py:267-268), everything else runs in float32.
 * SERIAL PER HANDLE: the uniform variant keeps its running maximum in one device word owned by the handle, and the
 * rebuild writes the handle's acceleration data -- issue the steps of one handle on one stream (or order them yourself).
 * A NaN second moment makes the uniform step NaN everywhere (the maximum keeps it), as dr.max over a NaN does in the
 * reference; the reference holds no fixture for that case (parity unpinned).
 */
#define HF_ADAM_MASK_UPDATES 1
#define HF_ADAM_UNIFORM 2
int hf_adam_step(hf_field_t *hf, float *d_heights, const float *d_grad, float *d_m, float *d_v, double lr,
                 double beta1, double beta2, double eps, uint32_t step, int mask_updates, hf_stream_t stream);
/* hf_adam_step for CAPTURED steps (HIP graphs).  hf_adam_step bakes the step number into its launch (the bias-corrected
 * step size lr_t is a kernel argument computed on the host), so a captured step would replay step 1 for ever.  Here
 * the step sizes come from DEVICE memory: d_lr_t[k] = hf_adam_lr_t(lr, beta1, beta2, k + 1) for the steps the caller
 * intends to run (filled once, on the host, with hf_adam_step's own arithmetic: bit-identical updates), and *d_step
 * (device, zero before the first step) selects the entry and is incremented on the stream after the update -- every
 * replay of the captured step is the next Adam step.  Same mask_updates bits, same rebuild, serial per handle. */
float hf_adam_lr_t(double lr, double beta1, double beta2, uint32_t step);
int hf_adam_step_scheduled(hf_field_t *hf, float *d_heights, const float *d_grad, float *d_m, float *d_v,
                           const float *d_lr_t, uint32_t *d_step, double beta1, double beta2, double eps,
                           int mask_updates, hf_stream_t stream);

/* Replaces: m_to_world update + update() (rectangle.cpp:101-112, 131-142). */
int hf_set_transform(hf_field_t *hf, const float to_world[12], const float *to_object_or_null);

/* Replaces: Shape::bbox() (include/mitsuba/render/shape.h:253; analog rectangle.cpp:114-124).
 * World-space {min xyz, max xyz}.  Synchronises `stream`-ordered height updates. */
int hf_bbox(hf_field_t *hf, float out[6]);

/* device pointer to the handle's own copy of the heights (width*height floats) */
int hf_heights_device(hf_field_t *hf, const float **out);
int hf_dims(const hf_field_t *hf, uint32_t *width, uint32_t *height);

/* ---- the hot path ------------------------------------------------------------ */
/* All of these launch on `stream` and return immediately; the calling thread's current HIP device must be the
 * handle's (HF_EDEVICE otherwise), all arrays are device memory of that device.
 *
 * Results: the closest hit is the minimum of (t, -prim_index) over the triangles the reference's fp32
 * Moeller-Trumbore arithmetic (include/mitsuba/render/mesh.h:357-380) reports as hit -- what a brute force over
 * every triangle returns, bit for bit.  That test is itself ill-conditioned for distant origins on needle terrain
 * (cells hundreds of times taller than wide: the computed barycentrics are off by half a cell from 8 object units
 * away at 4096^2, by more than a cell from 50): a few rays in 10^5 then carry a hit of fp32 noise, which this library
 * reports whenever the noise stays within its margins -- in every such case resolved against the full brute force so
 * far (profiles/r03_far_origin.txt) -- and which a BVH over the same triangles may or may not report. */

/* The `coherent` hint of Scene::ray_intersect / ray_test / ray_intersect_preliminary
 * (include/mitsuba/render/scene.h:117-146, 188-207, 237-259: "a hint that can improve performance in the first step of
 * finding the PreliminaryInteraction"; integrators pass coherent = true for camera rays, reparam.py:95 traces its
 * auxiliary rays with coherent = false).  A property of the handle, read by every trace launch that follows:
 *   HF_COHERENCE_AUTO        (default) every 64-ray batch decides for itself: packets take the beam sweep, the others
 *                            the per-lane walk; auxiliary rays (hf_reparam_trace*) as _INCOHERENT when kappa < 4e6
 *   HF_COHERENCE_INCOHERENT  = coherent false: kernels without the sweep (fewer registers, no LDS, 6-7 instead of 5
 *                            waves per SIMD).  Bounce rays -8 %, the reparameterisation backward -7 %; camera rays +25 %.
 *   HF_COHERENCE_COHERENT    = coherent true: as _AUTO, and auxiliary rays through the full kernel whatever kappa
 * The results do not depend on it (bit for bit: tests/test_gpu_parity.py).  Serial per handle like everything else.
 * (Shadow rays towards one light share a direction: hf_ray_test is 7 % slower with _INCOHERENT than with _AUTO on the bench's
 * 16.5 M shadow rays -- a caller that maps an integrator's default `coherent = false` onto this should do so for the
 * closest-hit launches only.) */
enum { HF_COHERENCE_AUTO = 0, HF_COHERENCE_INCOHERENT = 1, HF_COHERENCE_COHERENT = 2 };
int hf_set_ray_coherence(hf_field_t *hf, int coherence);
int hf_get_ray_coherence(const hf_field_t *hf);

/* Replaces: Shape::ray_intersect_preliminary(const Ray3f&, Mask)
 * (include/mitsuba/render/shape.h:137-138, wrapper shape.h:621-629; called from
 * include/mitsuba/render/kdtree.h:2509-2510 and src/render/shape.cpp:211). */
int hf_ray_intersect_preliminary(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                                 const uint8_t *active, const hf_pi_t *out,
                                 hf_stream_t stream);

/* Replaces: Shape::ray_test(const Ray3f&, Mask) (shape.h:153, 630-633;
 * semantics == ray_intersect_preliminary().is_valid(), src/render/shape.cpp:430-434). */
int hf_ray_test(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                const uint8_t *active, uint8_t *out_hit, hf_stream_t stream);

/* Replaces: Shape::compute_surface_interaction(ray, pi, ray_flags, recursion_depth=0, active)
 * (shape.h:179-183) followed by SurfaceInteraction::finalize_surface_interaction
 * (interaction.h:476-499), i.e. PreliminaryIntersection::compute_surface_interaction
 * (interaction.h:658-684). */
int hf_compute_surface_interaction(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                                   const hf_pi_const_t *pi, uint32_t ray_flags,
                                   const uint8_t *active, const hf_si_t *out,
                                   hf_stream_t stream);

/* Replaces: Shape::ray_intersect(ray, ray_flags, active) = preliminary + SI
 * (src/render/shape.cpp:436-446); one fused kernel.  out_pi may be NULL. */
int hf_ray_intersect(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                     uint32_t ray_flags, const uint8_t *active,
                     const hf_pi_t *out_pi, const hf_si_t *out_si, hf_stream_t stream);

/* Replaces: the Dr.Jit reverse-mode pass through compute_surface_interaction that
 * dr.backward_from() triggers (src/python/python/ad/integrators/prb_reparam.py:586-587):
 * every dr::gather from the parameter buffer becomes scatter_reduce(Add).
 * Accumulates dL/dheight (width*height floats, row-major) with float atomics;
 * grad_o / grad_d (3 arrays each, may be NULL) receive dL/d(ray.o), dL/d(ray.d)
 * per lane (overwritten).  grad_heights may be NULL when only ray gradients are wanted. */
int hf_adjoint(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
               const hf_pi_const_t *pi, uint32_t ray_flags, const uint8_t *active,
               const hf_si_grad_t *grad_si, float *grad_heights,
               float *const grad_o[3], float *const grad_d[3], hf_stream_t stream);
/* hf_adjoint that also reports WHICH texture rows it added to: row_band (device, 2 x uint32, may be NULL) is updated
 * with atomicMin / atomicMax to {lowest row touched, highest row touched + 1}; the caller initialises it to
 * {height, 0} before the launches it wants covered.  Rows outside the band of every rank hold zeros in every rank's
 * private gradient texture, so a multi-GPU host may all-reduce rows [lo, hi) only (hf_allreduce_grad on
 * grad_heights + lo * width with count (hi - lo) * width).  (Reference: the gradient of the whole parameter buffer is
 * what dr.backward leaves in params.grad, src/python/python/ad/integrators/common.py:312-329; no band there.) */
int hf_adjoint_rows(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                    const hf_pi_const_t *pi, uint32_t ray_flags, const uint8_t *active,
                    const hf_si_grad_t *grad_si, float *grad_heights,
                    float *const grad_o[3], float *const grad_d[3], uint32_t *row_band, hf_stream_t stream);

/* ---- next row: minimal direct lighting on the wavefront ------- */

/* A directional emitter (src/emitters/directional.cpp:82,174): unit direction TOWARDS the light, scalar irradiance. */
#define HF_MAX_LIGHTS 8
typedef struct hf_dir_light {
    float to_light[3];
    float irradiance;
} hf_dir_light_t;

/* Replaces, for diffuse surfaces under directional lights, the emitter-sampling term of the direct
 * integrator (src/python/python/ad/integrators/direct_reparam.py:149-175: detached emitter sample, delta
 * light => MIS weight 1, attached BSDF value) with the diffuse BSDF (src/bsdfs/diffuse.cpp:135-140:
 * albedo/pi * cos_o, zero unless cos_i > 0 and cos_o > 0 in the shading frame) and the box-filter film
 * (mean of the spp samples of a pixel, sample i belongs to pixel i / spp as in
 * src/render/integrator.cpp:251-268):
 *   image[k][i / spp] = 1/spp * sum_s  albedo/pi * E_k * max(0, <sh_n, l_k>) * vis_k      (t finite, <sh_n,-d> > 0)
 * sh_n, d: 3 device arrays of n floats each (SoA, as hf_si_t.sh_n / hf_rays_t.d); t: n floats (inf = miss);
 * lights: n_lights <= HF_MAX_LIGHTS structs in HOST memory; vis: NULL or n_lights device arrays of n bytes
 * (0 = shadowed: the negated hf_ray_test result of the shadow ray towards light k); image: device,
 * n_lights * (n / spp) floats, overwritten.  n must be a multiple of spp. */
int hf_direct_lighting(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3], const float *t,
                       uint32_t n_lights, const hf_dir_light_t *lights, float albedo, const uint8_t *const *vis,
                       float *image, hf_stream_t stream);
/* Reverse mode of the above with respect to sh_n (what dr.backward propagates into si before it reaches
 * compute_surface_interaction; the cos > 0 masks and the visibility are piecewise constant):
 *   grad_sh_n[i] = 1/spp * sum_k  albedo/pi * E_k * vis_k * grad_image[k][i / spp] * l_k      (same masks)
 * grad_sh_n: 3 device arrays of n floats, overwritten; feed them to hf_adjoint as hf_si_grad_t.sh_n. */
int hf_direct_lighting_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                               const float *t, uint32_t n_lights, const hf_dir_light_t *lights, float albedo,
                               const uint8_t *const *vis, const float *grad_image, float *const grad_sh_n[3],
                               hf_stream_t stream);
/* The same pair with a per-sample WEIGHT (n floats, device): every light's contribution of sample i is multiplied by
 * weight[i] before the film -- the determinant of a reparameterised camera ray, which direct_reparam.py:164-180 /
 * prb_reparam.py:317-366 multiply the sample by.  The adjoint also returns dL/dweight (grad_weight, n floats,
 * overwritten; may be NULL), the gradient that goes on to hf_reparam_*'s divergence input.  weight == NULL: as above. */
int hf_direct_lighting_weighted(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                const float *t, const float *weight, uint32_t n_lights, const hf_dir_light_t *lights,
                                float albedo, const uint8_t *const *vis, float *image, hf_stream_t stream);
int hf_direct_lighting_weighted_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                        const float *t, const float *weight, uint32_t n_lights,
                                        const hf_dir_light_t *lights, float albedo, const uint8_t *const *vis,
                                        const float *grad_image, float *const grad_sh_n[3], float *grad_weight,
                                        hf_stream_t stream);

/* The same under POINT lights (src/emitters/point.cpp:106-123: direction d = position - si.p, radiance
 * intensity / |d|^2): sample value albedo/pi * intensity / r^2 * max(0, <sh_n, l>) with l = (position - p) / r, same
 * masks and film.  p: si.p, 3 device arrays of n floats.  The adjoint returns the gradient with respect to sh_n AND
 * to p (the light direction and the falloff depend on the hit point):
 *   grad_sh_n[i] = sum_k w_k / r^2 * l,   grad_p[i] = sum_k w_k / r^3 * (3 <sh_n, l> l - sh_n),
 *   w_k = 1/spp * albedo/pi * intensity_k * vis_k * grad_image[k][i / spp]            (same masks)
 * both overwritten; feed them to hf_adjoint as hf_si_grad_t.sh_n / .p. */
typedef struct {
    float position[3];
    float intensity; /* radiant intensity (W/sr) */
} hf_point_light_t;
int hf_point_lighting(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3], const float *t,
                      const float *const p[3], uint32_t n_lights, const hf_point_light_t *lights, float albedo,
                      const uint8_t *const *vis, float *image, hf_stream_t stream);
int hf_point_lighting_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                              const float *t, const float *const p[3], uint32_t n_lights,
                              const hf_point_light_t *lights, float albedo, const uint8_t *const *vis,
                              const float *grad_image, float *const grad_sh_n[3], float *const grad_p[3],
                              hf_stream_t stream);

/* Film with a Gaussian reconstruction filter (the reference's default rfilter, src/rfilters/gaussian.cpp:48-101:
 * w(x) = max(0, exp(-x^2 / (2 stddev^2)) - exp(-r^2 / (2 stddev^2))), r = 4 stddev), splatted as ImageBlock::put does
 * (src/render/imageblock.cpp:258-330): sample i at film position (pos_x[i], pos_y[i]) (pixel units; pixel (x, y)
 * covers [x, x+1) x [y, y+1)) adds w(x - (pos_x - 0.5)) w(y - (pos_y - 0.5)) * values[c][i] to image[c][y*width + x]
 * and the weight alone to weight[y*width + x] for every pixel within r; image and weight are ACCUMULATED (zero them
 * first), the film is image / weight (HDRFilm::develop).  values: `channels` (<= HF_MAX_LIGHTS) device arrays of n
 * floats -- e.g. the rows of hf_direct_lighting(..., spp = 1, ...), which are the per-sample values.  0 < stddev <= 1.
 * The adjoint gathers: grad_values[c][i] = sum_pixels w * grad_image[c][pixel], grad_image = dL/d(accumulated image)
 * (for a loss on the normalised film: dL/d(film) / weight); the weights do not depend on the samples' values. */
int hf_film_splat(size_t n, uint32_t channels, const float *const *values, const float *pos_x, const float *pos_y,
                  uint32_t width, uint32_t height, float stddev, float *image, float *weight, hf_stream_t stream);
int hf_film_splat_adjoint(size_t n, uint32_t channels, const float *pos_x, const float *pos_y, uint32_t width,
                          uint32_t height, float stddev, const float *grad_image, float *const *grad_values,
                          hf_stream_t stream);

/* ---- next row: warped-area reparameterisation of rays ---------- */

/* The auxiliary-ray machinery of mitsuba.ad.reparameterize_ray (src/python/python/ad/reparam.py:10-123,
 * 224-333; Bangaru et al. 2020) for a scene that is this one shape, split into the two per-sample kernels
 * its three loops are made of; the auxiliary rays themselves are traced by hf_ray_intersect with
 * HF_RAY_ALL | HF_RAY_FOLLOWSHAPE | HF_RAY_BOUNDARYTEST (reparam.py:93-95) and the gradient of an
 * auxiliary hit reaches the heights through hf_adjoint with the same flags.
 * Random numbers: the reference draws two PCG32 floats per auxiliary ray (reparam.py:186); PCG32 lives in
 * the absent Dr.Jit, so sample (k, ray i) is sample_tea_32(key, id_i) with key = sample_tea_32(seed, pair)[0]
 * (include/mitsuba/core/random.h:76-91) -> two 23-bit floats, pair = k/2 with antithetic sampling (the even
 * iteration of a pair reuses the sample with omega_local.xy negated, reparam.py:83-85,188-190), else k.
 * id_i = i, the ray's index in the launch, or ray_id[i] (device array of n uint32, may be NULL): the index of the ray
 * in a larger wavefront of which this launch traces a part (a rank's image tiles: index = pixel * spp + sample) -- the
 * sharded launches then draw exactly the samples of the unsharded one (the reference seeds its sampler with the
 * wavefront index the same way, src/render/sampler.cpp:116-130).  All four hf_reparam_* calls of one ray take the
 * same seed and ray_id.
 *
 * hf_reparam_aux_rays: auxiliary ray k of every primary ray (o, d; d unit length): direction =
 * Frame3f(d).to_world(square_to_von_mises_fisher(sample, kappa)) (warp.h:546-583, frame.h:39-41,
 * vector.h:116-136), origin o, maxt = inf; inactive lanes get maxt = -1 (a miss). kappa > 0. */
int hf_reparam_aux_rays(size_t n, const float *const o[3], const float *const d[3], const uint8_t *active,
                        uint32_t k, float kappa, int antithetic, uint32_t seed, const uint32_t *ray_id,
                        float *const aux_d[3], float *aux_maxt, hf_stream_t stream);
/* hf_reparam_weights, mode 0 (reparam.py:97-121, first loop of backward :224-256): from the auxiliary hit
 * (si_t, si_boundary_test) the harmonic weight w = (1 / (D - 1 + B))^exponent * D and its tangential
 * gradient d_w_omega; accumulates Z[n] += w, dZ[3][n] += d_w_omega.
 * mode 1 (third loop :283-325 for the shape parameter): with the totals Z, dZ and the upstream gradients of
 * the outputs, grad_direction[3][n] and grad_divergence[n] (reparam.py:262-281: direction =
 * normalize(d + V / Z), divergence = (div - <V / Z, dZ>) / Z at V = 0), the gradient of this sample's
 * V_direct = (si.p - o) / si.t is written as upstream gradient of the auxiliary hit: grad_p[3][n] and
 * grad_t[n] (zero for misses), to be handed to hf_adjoint as hf_si_grad_t {p, t} with HF_RAY_FOLLOWSHAPE.
 * grad_vd (optional, NULL = not wanted): the gradient with respect to this sample's V_direct itself,
 * grad_vd[3][n] (zero for inactive lanes).  It is what the RAY needs (reparam.py:296-325 back-propagates to
 * ray.o and ray.d as well): for a hit, dL/d(ray.o) = [grad_o of hf_adjoint] - grad_p; for a miss V_direct is
 * ray.d itself and dL/d(ray.d) = grad_vd; dL/d(auxiliary direction) = [grad_d of hf_adjoint] is carried to
 * ray.d through Frame3f(ray.d) by the host (mirror: hf_amd.reparameterize_ray). */
int hf_reparam_weights(int mode, size_t n, const float *const o[3], const float *const d[3],
                       const uint8_t *active, uint32_t k, float kappa, float exponent, int antithetic,
                       uint32_t seed, const uint32_t *ray_id, const float *si_t, const float *const si_p[3],
                       const float *si_boundary_test,
                       float *Z, float *const dZ[3], const float *const grad_direction[3],
                       const float *grad_divergence, float *const grad_p[3], float *grad_t,
                       float *const grad_vd[3], hf_stream_t stream);
/* hf_reparam_aux_rays + hf_ray_intersect in one launch: traces auxiliary ray k of every ray (o, d) with
 * HF_RAY_ALL | HF_RAY_FOLLOWSHAPE | HF_RAY_BOUNDARYTEST (reparam.py:93-95) without materialising the auxiliary rays;
 * inactive lanes are misses.  out_pi / out_si as in hf_ray_intersect (si.wi is expressed for the auxiliary direction).
 * Bitwise the two-call sequence. */
int hf_reparam_trace(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                     const uint8_t *active, uint32_t k, float kappa, int antithetic, uint32_t seed,
                     const uint32_t *ray_id, const hf_pi_t *out_pi, const hf_si_t *out_si, hf_stream_t stream);
/* hf_reparam_trace for samples 0 .. num_rays - 1 in ONE launch: sample k of ray i goes to [k * sample_stride + i] of
 * every row of out_pi / out_si (the layout hf_reparam_backward reads).  A ray is fetched once and its samples are
 * traced back to back; and because every von Mises-Fisher sample lies within theta_max of the ray (cos theta_max =
 * 1 - 13.82 / kappa, warp.h:557-566), a batch of rays whose cones all miss the bound is answered with num_rays miss
 * records without a sample being drawn (not when out_si->wi is asked for).  Bitwise num_rays x hf_reparam_trace. */
int hf_reparam_trace_all(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                         const uint8_t *active, uint32_t num_rays, float kappa, int antithetic, uint32_t seed,
                         const uint32_t *ray_id, const hf_pi_t *out_pi, const hf_si_t *out_si, size_t sample_stride,
                         hf_stream_t stream);
/* The same backward pass in ONE kernel for the case that only the heights are differentiated (grad(ray) not wanted):
 * for every ray the weights of its num_rays samples and their sums Z, dZ (reparam.py:236-256), then for every
 * auxiliary HIT the gradient of its V_direct through the FollowShape surface interaction into grad_heights[H*W]
 * (accumulated, +=; reparam.py:296-325).  Inputs are the auxiliary hits only -- what hf_ray_intersect returned for
 * the rays of hf_reparam_aux_rays with HF_RAY_ALL | HF_RAY_FOLLOWSHAPE | HF_RAY_BOUNDARYTEST: sample k of ray i at
 * index [k * sample_stride + i] of pi->t / prim_uv / prim_index and of si_boundary_test.  The auxiliary directions are
 * regenerated from (d, k, seed); si.p and si.t are re-derived from pi with compute_surface_interaction's expressions,
 * so the result equals num_rays x (hf_reparam_weights mode 0), then num_rays x (mode 1 + hf_adjoint) up to the order
 * of the float atomics.  1 <= num_rays <= 32. */
int hf_reparam_backward(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                        const uint8_t *active, uint32_t num_rays, float kappa, float exponent, int antithetic,
                        uint32_t seed, const uint32_t *ray_id, const hf_pi_const_t *pi, const float *si_boundary_test,
                        size_t sample_stride,
                        const float *const grad_direction[3], const float *grad_divergence, float *grad_heights,
                        hf_stream_t stream);

/* ---- scalar / packet entry -----------------------------------------------------
 * Shape::ray_intersect_preliminary_scalar / _packet and ray_test_scalar / _packet
 * (include/mitsuba/render/shape.h:220-240, wrapper macros :594-641): the forms the scalar and LLVM
 * variants call per kd-tree leaf (include/mitsuba/render/kdtree.h:2490-2520) and from Embree's user-geometry
 * callbacks (src/render/shape.cpp:125-223) with 1, 4, 8 or 16 rays held in HOST registers.  These two entry
 * points take HOST pointers (SoA, n <= HF_PACKET_MAX), stage the packet through a per-thread pinned buffer and
 * a per-thread stream, run the same traversal kernel and return when the results are back in the host arrays
 * (synchronous, thread-safe, any number of threads).  One call costs a kernel launch and two PCIe round trips
 * (tens of microseconds): it exists so that an adapter can serve those call sites with the SAME arithmetic;
 * a renderer should hand whole wavefronts to the device entry points above.  `active` NULL = all lanes.
 * Inactive / missed lanes: t = +inf, prim_uv = 0, prim_index = 0; hit = 0. */
#define HF_PACKET_MAX 16
int hf_ray_intersect_preliminary_packet(const hf_field_t *hf, uint32_t n, const float *const h_o[3],
                                        const float *const h_d[3], const float *h_maxt, const uint8_t *h_active,
                                        float *h_t, float *const h_prim_uv[2], uint32_t *h_prim_index);
int hf_ray_test_packet(const hf_field_t *hf, uint32_t n, const float *const h_o[3], const float *const h_d[3],
                       const float *h_maxt, const uint8_t *h_active, uint8_t *h_hit);

/* ---- multi-GPU -------------------------------------------------------------------
 * Rays shard over image tiles, heights and acceleration data are replicated, every GPU accumulates a private
 * dL/dheight texture; this is the ONE collective of the path: an in-place sum all-reduce (float32) of that
 * texture over RCCL / xGMI, enqueued on `stream` (so it is ordered after the hf_adjoint launches of that stream
 * and can overlap the next wavefront's forward pass running on another stream).
 * `rccl_comm` is the caller's ncclComm_t for this rank (ncclCommInitRank by the host application), as void*.
 * RCCL is bound at first use: the ncclAllReduce already present in the process (the host's own RCCL, so that
 * the communicator and the call come from the same library), otherwise librccl.so is loaded.
 * Returns HF_EDEVICE if RCCL cannot be found or reports an error. */
int hf_allreduce_grad(float *d_grad, size_t count, void *rccl_comm, hf_stream_t stream);

/* ---- introspection (tests / tools) --------------------------------------------- */
int hf_num_levels(const hf_field_t *hf);
/* copies mip level `level` (1..num_levels) to HOST memory as (min,max) pairs,
 * row-major w x h; out may be NULL to query w,h.  Synchronises. */
int hf_get_mip(const hf_field_t *hf, int level, float *h_out, uint32_t *w, uint32_t *h);
/* inverse of a row-major 3x4 affine matrix (double precision, rounded to float) */
int hf_invert_affine(const float in[12], float out[12]);
const char *hf_last_error_string(void);
int hf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HF_H */
