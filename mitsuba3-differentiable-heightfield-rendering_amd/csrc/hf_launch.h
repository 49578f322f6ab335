// hf_launch.h -- host-side launch entry points implemented in hf_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/hf.h"
#include "hf_device.h"

void hf_launch_build_mips(const hf_dev_field &f, float2 *mip, float4 *shear, hipStream_t stream);
// scratch block a trace launch needs (its work counters; the launcher zeroes it): hf_trace_scratch_bytes(n)
// bytes, exclusively this launch's until it has completed
size_t hf_trace_scratch_bytes(size_t n);
// mode 0: closest hit -> pi; 1: any hit -> hit; 2: closest hit + fused surface interaction
struct hf_reparam_args;
// aux (mode 2 only, may be NULL): trace auxiliary ray aux->k of every ray (k, seed, kappa, antithetic are read)
void hf_launch_trace(int mode, const hf_dev_field &f, size_t n, const hf_rays_t *rays, const uint8_t *active,
                     const hf_pi_t *pi, uint8_t *hit, const hf_si_t *si, uint32_t flags, void *scratch,
                     hipStream_t stream, const hf_reparam_args *aux = nullptr, bool lean = false); // lean: the launch is declared incoherent (hf_set_ray_coherence)
void hf_launch_si(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                  const uint8_t *active, const hf_si_t *si, uint32_t flags, hipStream_t stream);
void hf_launch_adjoint(const hf_dev_field &f, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                       const uint8_t *active, const hf_si_grad_t *gs, uint32_t flags, float *grad_h,
                       float *const grad_o[3], float *const grad_d[3], uint32_t *row_band, hipStream_t stream);
// c1 = (float)(1 - beta1), c2 = (float)(1 - beta2): the differences are Python doubles in optimizers.py:279-280,
// rounded once when they meet the float32 gradient
hipError_t hf_launch_adam(size_t n, float *h, const float *g, float *m, float *v, float lr_t, float beta1, float beta2,
                    float c1, float c2, float eps, int mask_updates, hipStream_t stream,
                    uint32_t *uniform_scratch = nullptr, const float *sched = nullptr, uint32_t *ctr = nullptr); // non-NULL: the UniformAdam variant (one device word of scratch)
struct hf_lights_dev {
    float l[HF_MAX_LIGHTS][3];
    float w[HF_MAX_LIGHTS]; // albedo/pi * irradiance
    const uint8_t *vis[HF_MAX_LIGHTS];
    uint32_t n;
    const float *weight;  // optional per-sample factor of every light's contribution (hf_direct_lighting_weighted)
    float *grad_weight;   // adjoint, optional: dL/dweight per sample
};
// p == nullptr: directional lights (lights.l = unit direction towards the light, lights.w = albedo/pi * irradiance);
// p != nullptr: point lights (lights.l = position, lights.w = albedo/pi * intensity), grad_p is then written too
void hf_launch_direct(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3], const float *t,
                      const float *const p[3], const hf_lights_dev &lights, float *image, hipStream_t stream);
void hf_launch_direct_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                              const float *t, const float *const p[3], const hf_lights_dev &lights,
                              const float *grad_image, float *const grad_sh_n[3], float *const grad_p[3],
                              hipStream_t stream);
struct hf_splat_args {
    size_t n;
    uint32_t channels, width, height;
    float alpha, bias, radius; // -1 / (2 stddev^2), exp(alpha r^2), r
    const float *pos_x, *pos_y;
    const float *values[HF_MAX_LIGHTS];   // forward: per-channel sample values
    float *image, *weight;                // forward: accumulated [channels][H*W], [H*W]
    const float *grad_image;              // adjoint: dL/d(accumulated image) [channels][H*W]
    float *grad_values[HF_MAX_LIGHTS];    // adjoint: per-channel, overwritten
};
void hf_launch_film_splat(const hf_splat_args &a, bool adjoint, hipStream_t stream);
struct hf_reparam_args {
    size_t n;
    const float *o[3], *d[3];
    const uint8_t *active;
    uint32_t k, seed;
    uint32_t num; size_t stride;                      // hf_launch_trace: samples k .. k + num - 1 in one launch, sample j at [j * stride + i] (num <= 1: sample k only)
    const uint32_t *ray_id;                           // optional: the id of ray i in the sample streams (NULL: i)
    float kappa, exponent;
    int antithetic, mode;
    float *aux_d[3], *aux_maxt;                       // aux-ray generation
    const float *si_t, *si_p[3], *si_bt;              // auxiliary hit
    float *Z, *dZ[3];                                 // mode 0: accumulated, mode 1: read
    const float *g_dir[3], *g_div;                    // mode 1
    float *g_p[3], *g_t;                              // mode 1 outputs
    float *g_vd[3];                                   // mode 1, optional: gradient w.r.t. V_direct (ray gradients)
};
void hf_launch_reparam_aux(const hf_reparam_args &a, hipStream_t stream);
void hf_launch_reparam_weights(const hf_reparam_args &a, hipStream_t stream);
void hf_launch_reparam_backward(const hf_dev_field &f, const hf_reparam_args &a, uint32_t num_rays, size_t stride,
                                const hf_pi_const_t *pi, float *grad_h, hipStream_t stream);
