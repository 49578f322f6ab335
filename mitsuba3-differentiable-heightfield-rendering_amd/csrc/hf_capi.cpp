// hf_capi.cpp -- the C ABI of libhf (include/hf.h): handle management, argument
// validation, error strings.  All device work is in hf_kernels.hip.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <new>
#include <vector>

#include "hf_launch.h"

#define HF_NUM_SLOTS 64
struct hf_field {
    hf_dev_field dev;   // device view handed to kernels by value
    float *d_heights;   // owned copy of the heights
    float2 *d_mip;      // owned min/max pyramid
    float4 *d_shear;    // owned sheared bounds of the fine levels
    size_t mip_nodes;
    int device;
    uint32_t *d_misc;   // a few device words of the handle's own (the max of UniformAdam's second moments)
    hipEvent_t built;   // completion of the last hf_set_heights*
    // Ring of scratch blocks (the work counters of a launch), one slot per in-flight trace launch.  Every slot
    // carries the completion event of the launch that used it last: a launch that re-uses the slot first makes
    // its stream wait for that event (a device-side wait, normally long past), so two pending launches never
    // share a block, and hf_destroy waits for exactly the launches of this handle instead of the whole device.
    // Choice of the slot, launch and event record are ONE critical section (slot_lease below).
    // Blocks are allocated by hf_create (hf_trace_scratch_bytes is a constant today; a larger request re-allocates).
    // HIP-graph capture: a launch issued while its stream is being captured takes its block from the upper half of
    // the ring, touches no event, and keeps the block until hf_capture_reset (slot_lease).
    char *slot_buf[HF_NUM_SLOTS];
    size_t slot_cap[HF_NUM_SLOTS];
    hipEvent_t slot_done[HF_NUM_SLOTS];
    bool slot_used[HF_NUM_SLOTS];
    uint32_t next_slot, next_capture_slot;
    std::mutex *slot_mutex;
    int coherence;      // hf_set_ray_coherence: HF_COHERENCE_AUTO / _INCOHERENT (which instantiation the trace launches take)
};

static bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void) hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

// keeps the caller's current device across a call that has to work on the handle's device
struct hf_device_guard {
    int prev = -1;
    bool ok = true;
    explicit hf_device_guard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            prev = cur;
        }
    }
    ~hf_device_guard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

// Lease of one scratch block for ONE trace launch on `stream`.  The handle's slot mutex is held from the choice of
// the slot until the launch's completion event has been recorded (the destructor), so the three steps are one
// critical section: a later lease of the same slot -- from any thread, after the ring has wrapped -- always finds the
// event of the launch that used the block last and makes its stream wait for it (a device-side wait, normally long
// past).  The launch itself is asynchronous; the section costs microseconds.  (acquire / launch / record
// were three sections, and 32 other leases between a thread's acquire and its record could hand the block to a second
// launch that neither waited for the first nor kept its counters.)
// Captured launches (the stream is being captured into a HIP graph): the block comes from the upper half of the
// ring, no event is touched (an event recorded inside a capture cannot be waited for outside it), and every captured
// launch KEEPS its block for as long as the graph may be replayed: the 33rd captured trace launch of a handle is refused
// with HF_EINVAL instead of sharing work counters with the first; hf_capture_reset() returns the blocks once the
// graphs captured so far have been destroyed.
struct slot_lease {
    hf_field *m;
    hipStream_t stream;
    uint32_t slot = 0;
    void *buf = nullptr;
    const char *why = "scratch allocation failed";
    int code = HF_ENOMEM;
    std::unique_lock<std::mutex> lock;
    slot_lease(const hf_field *hf, hipStream_t st, size_t bytes)
        : m(const_cast<hf_field *>(hf)), stream(st), lock(*const_cast<hf_field *>(hf)->slot_mutex) {
        if (stream_capturing(stream)) { // no allocation, no events
            if (m->next_capture_slot >= HF_NUM_SLOTS / 2) {
                code = HF_EINVAL;
                why = "more than 32 captured trace launches on this handle (their scratch blocks stay reserved for "
                      "replays; destroy the graphs and call hf_capture_reset)";
                return;
            }
            slot = HF_NUM_SLOTS / 2 + m->next_capture_slot;
            if (m->slot_cap[slot] >= bytes) { buf = m->slot_buf[slot]; ++m->next_capture_slot; }
            return;
        }
        slot = m->next_slot++ % (HF_NUM_SLOTS / 2);
        if (m->slot_cap[slot] < bytes) { // grow: the previous user must be done before its block goes away
            if (m->slot_used[slot]) (void) hipEventSynchronize(m->slot_done[slot]);
            if (m->slot_buf[slot]) (void) hipFree(m->slot_buf[slot]);
            m->slot_buf[slot] = nullptr; m->slot_cap[slot] = 0;
            size_t cap = bytes < 65536 ? 65536 : bytes + bytes / 2;
            if (hipMalloc((void **) &m->slot_buf[slot], cap) != hipSuccess) return;
            m->slot_cap[slot] = cap;
        } else if (m->slot_used[slot]) {
            (void) hipStreamWaitEvent(stream, m->slot_done[slot], 0);
        }
        buf = m->slot_buf[slot];
    }
    ~slot_lease() {
        if (!buf || slot >= HF_NUM_SLOTS / 2) return; // nothing launched, or a captured launch
        (void) hipEventRecord(m->slot_done[slot], stream);
        m->slot_used[slot] = true;
    }
};

static thread_local char g_err[512] = "no error";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HF_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? HF_ENOMEM : HF_EDEVICE, "%s: %s", #call,     \
                        hipGetErrorString(e_));                                                   \
    } while (0)

extern "C" const char *hf_last_error_string(void) { return g_err; }
extern "C" int hf_version(void) { return HF_VERSION; }

// inverse of [A | t] = [A^-1 | -A^-1 t], adjugate formula in double
extern "C" int hf_invert_affine(const float in[12], float out[12]) {
    if (!in || !out) return fail(HF_EINVAL, "hf_invert_affine: NULL argument");
    double a[3][3], t[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) a[r][c] = in[4 * r + c];
        t[r] = in[4 * r + 3];
    }
    double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
           c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    if (det == 0.0 || !isfinite(det)) return fail(HF_EINVAL, "hf_invert_affine: singular to_world");
    double inv[3][3];
    inv[0][0] = c00 / det;
    inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    inv[1][0] = c01 / det;
    inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
    inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    inv[2][0] = c02 / det;
    inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[4 * r + c] = (float) inv[r][c];
        out[4 * r + 3] = (float) -(inv[r][0] * t[0] + inv[r][1] * t[1] + inv[r][2] * t[2]);
    }
    return HF_OK;
}

static int set_transform(hf_field *hf, const float *to_world, const float *to_object) {
    memcpy(hf->dev.to_world, to_world, sizeof(float) * 12);
    if (to_object) {
        memcpy(hf->dev.to_object, to_object, sizeof(float) * 12);
        return HF_OK;
    }
    return hf_invert_affine(to_world, hf->dev.to_object);
}

// frees whatever a (possibly half-constructed) handle owns
static void release(hf_field *hf) {
    if (hf->built) (void) hipEventDestroy(hf->built);
    for (int k = 0; k < HF_NUM_SLOTS; ++k) {
        if (hf->slot_done[k]) (void) hipEventDestroy(hf->slot_done[k]);
        if (hf->slot_buf[k]) (void) hipFree(hf->slot_buf[k]);
    }
    if (hf->d_heights) (void) hipFree(hf->d_heights);
    if (hf->d_mip) (void) hipFree(hf->d_mip);
    if (hf->d_shear) (void) hipFree(hf->d_shear);
    if (hf->d_misc) (void) hipFree(hf->d_misc);
    delete hf->slot_mutex;
    free(hf);
}

extern "C" int hf_create(const hf_desc_t *desc, hf_field_t **out) {
    if (!desc || !out) return fail(HF_EINVAL, "hf_create: NULL argument");
    *out = nullptr;
    if (desc->width < 2 || desc->height < 2)
        return fail(HF_EINVAL, "hf_create: heightfield resolution must be at least 2x2 (got %ux%u)", desc->width,
                    desc->height);
    if ((uint64_t) (desc->width - 1) * (desc->height - 1) >= (1ull << 31))
        return fail(HF_EINVAL, "hf_create: too many cells for a 32-bit prim_index");
    int ndev = 0;
    HF_HIP(hipGetDeviceCount(&ndev));
    if (desc->device < 0 || desc->device >= ndev)
        return fail(HF_EDEVICE, "hf_create: device %d not available (%d devices)", desc->device, ndev);
    hf_device_guard guard(desc->device); // the caller's current device is restored on return
    if (!guard.ok) return fail(HF_EDEVICE, "hf_create: cannot select device %d", desc->device);

    hf_field *hf = (hf_field *) calloc(1, sizeof(hf_field)); // zero-initialised POD
    if (!hf) return fail(HF_ENOMEM, "hf_create: host allocation failed");
    hf->device = desc->device;
    hf_dev_field &d = hf->dev;
    d.W = (int) desc->width; d.H = (int) desc->height;
    d.s = desc->max_height;
    d.sx = 2.0f / (float) (d.W - 1); d.sy = 2.0f / (float) (d.H - 1);
    d.iu = 1.0f / (float) (d.W - 1); d.iv = 1.0f / (float) (d.H - 1);
    d.hx = 0.5f * (float) (d.W - 1); d.hy = 0.5f * (float) (d.H - 1);
    d.flip = desc->flip_normals ? 1 : 0;
    int rc = set_transform(hf, desc->to_world, desc->has_to_object ? desc->to_object : nullptr);
    if (rc != HF_OK) { free(hf); return rc; }

    const int cw = d.W - 1, ch = d.H - 1;
    int top = 0;
    while ((1 << top) < cw || (1 << top) < ch) ++top;
    if (top < 1) top = 1;
    if (top > 15) { free(hf); return fail(HF_EINVAL, "hf_create: grid too large (more than 32768 cells per side)"); }
    if ((size_t) d.W * (size_t) d.H > ((size_t) 1 << 30)) { // the kernels address heights with 32-bit byte offsets
        free(hf);
        return fail(HF_EINVAL, "hf_create: grid too large (more than 2^30 vertices)");
    }
    d.top = top;
    const size_t off = hf_depth_off(top); // (4^top - 1)/3 nodes, depths 0..top-1
    hf->mip_nodes = off;
    hipError_t e = hipMalloc((void **) &hf->d_heights, sizeof(float) * (size_t) d.W * d.H);
    if (e == hipSuccess) e = hipMalloc((void **) &hf->d_mip, sizeof(float2) * off);
    if (e == hipSuccess) e = hipMalloc((void **) &hf->d_shear, sizeof(float4) * 3 * (hf_shear_records(top) + 1));
    if (e == hipSuccess) e = hipMalloc((void **) &hf->d_misc, 256);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&hf->built, hipEventDisableTiming);
    for (int k = 0; k < HF_NUM_SLOTS && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&hf->slot_done[k], hipEventDisableTiming);
        const size_t cap = hf_trace_scratch_bytes(0) < 4096 ? 4096 : hf_trace_scratch_bytes(0);
        if (e == hipSuccess) e = hipMalloc((void **) &hf->slot_buf[k], cap);
        if (e == hipSuccess) hf->slot_cap[k] = cap;
    }
    hf->coherence = HF_COHERENCE_AUTO;
    hf->slot_mutex = new (std::nothrow) std::mutex();
    if (e == hipSuccess && !hf->slot_mutex) e = hipErrorOutOfMemory;
    d.h = hf->d_heights;
    d.mip = hf->d_mip;
    d.shear = hf->d_shear;
    // heights start as zero; build the pyramid so the handle is always traceable
    if (e == hipSuccess) e = hipMemsetAsync(hf->d_heights, 0, sizeof(float) * (size_t) d.W * d.H, nullptr);
    if (e == hipSuccess) {
        hf_launch_build_mips(d, hf->d_mip, hf->d_shear, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(hf->built, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { // one exit for every failure after the first allocation: nothing leaks
        release(hf);
        return fail(e == hipErrorOutOfMemory ? HF_ENOMEM : HF_EDEVICE, "hf_create: %s", hipGetErrorString(e));
    }
    *out = hf;
    return HF_OK;
}

extern "C" int hf_destroy(hf_field_t *hf) {
    if (!hf) return HF_OK;
    hf_device_guard guard(hf->device);
    // wait for the work of THIS handle only (its last rebuild and the launches still holding a counter slot);
    // other streams of the host application keep running
    (void) hipEventSynchronize(hf->built);
    for (int k = 0; k < HF_NUM_SLOTS; ++k)
        if (hf->slot_used[k]) (void) hipEventSynchronize(hf->slot_done[k]);
    release(hf);
    return HF_OK;
}

extern "C" int hf_capture_reset(hf_field_t *hf) {
    if (!hf) return fail(HF_EINVAL, "hf_capture_reset: NULL handle");
    std::lock_guard<std::mutex> lock(*hf->slot_mutex);
    hf->next_capture_slot = 0;
    return HF_OK;
}

extern "C" int hf_set_heights(hf_field_t *hf, const float *d_heights, hf_stream_t stream) {
    if (!hf || !d_heights) return fail(HF_EINVAL, "hf_set_heights: NULL argument");
    hf_device_guard guard(hf->device);
    if (!guard.ok) return fail(HF_EDEVICE, "hf_set_heights: cannot select device %d", hf->device);
    hipStream_t st = (hipStream_t) stream;
    const size_t bytes = sizeof(float) * (size_t) hf->dev.W * hf->dev.H;
    if (d_heights != hf->d_heights)
        HF_HIP(hipMemcpyAsync(hf->d_heights, d_heights, bytes, hipMemcpyDeviceToDevice, st));
    hf_launch_build_mips(hf->dev, hf->d_mip, hf->d_shear, st);
    HF_HIP(hipGetLastError());
    if (!stream_capturing(st)) HF_HIP(hipEventRecord(hf->built, st)); // (a captured rebuild is ordered by its graph)
    return HF_OK;
}

extern "C" int hf_adam_step(hf_field_t *hf, float *d_heights, const float *d_grad, float *d_m, float *d_v, double lr,
                            double beta1, double beta2, double eps, uint32_t step, int mask_updates,
                            hf_stream_t stream) {
    if (!hf || !d_heights || !d_grad || !d_m || !d_v) return fail(HF_EINVAL, "hf_adam_step: NULL argument");
    // optimizers.py:248-249
    if (!(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(lr > 0.) || !(eps > 0.) || step == 0)
        return fail(HF_EINVAL, "hf_adam_step: need 0 <= beta < 1, lr > 0, eps > 0, step >= 1");
    hf_device_guard guard(hf->device);
    if (!guard.ok) return fail(HF_EDEVICE, "hf_adam_step: cannot select device %d", hf->device);
    // lr_scale in double, rounded once, like the Python scalar the reference makes opaque (optimizers.py:267-268)
    const float lr_scale = (float) (sqrt(1.0 - pow(beta2, (double) step)) / (1.0 - pow(beta1, (double) step)));
    const float lr_t = (float) lr * lr_scale;
    // mask_updates: bit 0 = mask_updates, bit 1 = the 'uniform' variant (optimizers.py:259, 290-291)
    HF_HIP(hf_launch_adam((size_t) hf->dev.W * hf->dev.H, d_heights, d_grad, d_m, d_v, lr_t, (float) beta1, (float) beta2,
                          (float) (1.0 - beta1), (float) (1.0 - beta2), (float) eps, mask_updates & 1, (hipStream_t) stream,
                          (mask_updates & 2) ? hf->d_misc : nullptr));
    HF_HIP(hipGetLastError());
    return hf_set_heights(hf, d_heights, stream);
}

extern "C" float hf_adam_lr_t(double lr, double beta1, double beta2, uint32_t step) {
    // the arithmetic of hf_adam_step: lr_scale in double, rounded once, times the float learning rate
    const float lr_scale = (float) (sqrt(1.0 - pow(beta2, (double) step)) / (1.0 - pow(beta1, (double) step)));
    return (float) lr * lr_scale;
}

extern "C" int hf_adam_step_scheduled(hf_field_t *hf, float *d_heights, const float *d_grad, float *d_m, float *d_v,
                                      const float *d_lr_t, uint32_t *d_step, double beta1, double beta2, double eps,
                                      int mask_updates, hf_stream_t stream) {
    if (!hf || !d_heights || !d_grad || !d_m || !d_v || !d_lr_t || !d_step)
        return fail(HF_EINVAL, "hf_adam_step_scheduled: NULL argument");
    if (!(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps > 0.))
        return fail(HF_EINVAL, "hf_adam_step_scheduled: need 0 <= beta < 1, eps > 0");
    hf_device_guard guard(hf->device);
    if (!guard.ok) return fail(HF_EDEVICE, "hf_adam_step_scheduled: cannot select device %d", hf->device);
    HF_HIP(hf_launch_adam((size_t) hf->dev.W * hf->dev.H, d_heights, d_grad, d_m, d_v, 0.f, (float) beta1, (float) beta2,
                          (float) (1.0 - beta1), (float) (1.0 - beta2), (float) eps, mask_updates & 1, (hipStream_t) stream,
                          (mask_updates & 2) ? hf->d_misc : nullptr, d_lr_t, d_step));
    HF_HIP(hipGetLastError());
    return hf_set_heights(hf, d_heights, stream);
}

extern "C" int hf_set_heights_host(hf_field_t *hf, const float *h_heights, hf_stream_t stream) {
    if (!hf || !h_heights) return fail(HF_EINVAL, "hf_set_heights_host: NULL argument");
    hf_device_guard guard(hf->device);
    if (!guard.ok) return fail(HF_EDEVICE, "hf_set_heights_host: cannot select device %d", hf->device);
    hipStream_t st = (hipStream_t) stream;
    const size_t bytes = sizeof(float) * (size_t) hf->dev.W * hf->dev.H;
    HF_HIP(hipMemcpyAsync(hf->d_heights, h_heights, bytes, hipMemcpyHostToDevice, st));
    HF_HIP(hipStreamSynchronize(st)); // the host buffer may be reused by the caller
    return hf_set_heights(hf, hf->d_heights, stream);
}

extern "C" int hf_set_transform(hf_field_t *hf, const float to_world[12], const float *to_object_or_null) {
    if (!hf || !to_world) return fail(HF_EINVAL, "hf_set_transform: NULL argument");
    return set_transform(hf, to_world, to_object_or_null);
}

extern "C" int hf_heights_device(hf_field_t *hf, const float **out) {
    if (!hf || !out) return fail(HF_EINVAL, "hf_heights_device: NULL argument");
    *out = hf->d_heights;
    return HF_OK;
}

extern "C" int hf_dims(const hf_field_t *hf, uint32_t *width, uint32_t *height) {
    if (!hf) return fail(HF_EINVAL, "hf_dims: NULL argument");
    if (width) *width = (uint32_t) hf->dev.W;
    if (height) *height = (uint32_t) hf->dev.H;
    return HF_OK;
}

extern "C" int hf_num_levels(const hf_field_t *hf) { return hf ? hf->dev.top : 0; }

extern "C" int hf_get_mip(const hf_field_t *hf, int level, float *h_out, uint32_t *w, uint32_t *h) {
    if (!hf || level < 1 || level > hf->dev.top) return fail(HF_EINVAL, "hf_get_mip: bad level %d", level);
    const int cw = hf->dev.W - 1, ch = hf->dev.H - 1, k = hf->dev.top - level;
    const int wl = hf_level_w(cw, level), hl = hf_level_w(ch, level);
    if (w) *w = (uint32_t) wl;
    if (h) *h = (uint32_t) hl;
    if (h_out) { // existing nodes of the padded level, row-major (min,max) pairs
        hf_device_guard guard(hf->device);
        std::vector<float> tmp(2 * ((size_t) 1 << (2 * k)));
        HF_HIP(hipEventSynchronize(hf->built));
        HF_HIP(hipMemcpy(tmp.data(), hf->d_mip + hf_depth_off(k), sizeof(float) * tmp.size(), hipMemcpyDeviceToHost));
        for (int iy = 0; iy < hl; ++iy)
            for (int ix = 0; ix < wl; ++ix) {
                const size_t slot = ((size_t) iy << k) + ix;
                h_out[2 * ((size_t) iy * wl + ix) + 0] = tmp[2 * slot + 0];
                h_out[2 * ((size_t) iy * wl + ix) + 1] = tmp[2 * slot + 1];
            }
    }
    return HF_OK;
}

// world-space box of the 8 corners of the object-space bound (analog: rectangle.cpp:114-124)
extern "C" int hf_bbox(hf_field_t *hf, float out[6]) {
    if (!hf || !out) return fail(HF_EINVAL, "hf_bbox: NULL argument");
    float zr[2];
    hf_device_guard guard(hf->device);
    HF_HIP(hipEventSynchronize(hf->built));
    HF_HIP(hipMemcpy(zr, hf->d_mip + 1, sizeof(zr), hipMemcpyDeviceToHost));
    const hf_dev_field &d = hf->dev;
    const float lo[3] = { fmaf(0.f, d.sx, -1.f), fmaf(0.f, d.sy, -1.f), zr[0] };
    const float hi[3] = { fmaf((float) (d.W - 1), d.sx, -1.f), fmaf((float) (d.H - 1), d.sy, -1.f), zr[1] };
    for (int k = 0; k < 3; ++k) { out[k] = INFINITY; out[3 + k] = -INFINITY; }
    const float *m = d.to_world;
    for (int c = 0; c < 8; ++c) {
        const float p[3] = { (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2] };
        for (int k = 0; k < 3; ++k) {
            float acc = m[4 * k + 3];
            acc = fmaf(m[4 * k + 0], p[0], acc);
            acc = fmaf(m[4 * k + 1], p[1], acc);
            acc = fmaf(m[4 * k + 2], p[2], acc);
            out[k] = fminf(out[k], acc);
            out[3 + k] = fmaxf(out[3 + k], acc);
        }
    }
    return HF_OK;
}

static int check_rays(const char *fn, const hf_field_t *hf, size_t n, const hf_rays_t *rays) {
    if (!hf || !rays) return fail(HF_EINVAL, "%s: NULL argument", fn);
    if (n == 0) return HF_OK;
    for (int k = 0; k < 3; ++k)
        if (!rays->o[k] || !rays->d[k]) return fail(HF_EINVAL, "%s: NULL ray component array", fn);
    if (!rays->maxt) return fail(HF_EINVAL, "%s: NULL ray maxt array", fn);
    // the launch goes to the calling thread's current device: it must be the handle's (the query functions do
    // not switch devices behind the caller's back)
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != hf->device)
        return fail(HF_EDEVICE, "%s: current HIP device is %d, the heightfield lives on device %d", fn, cur, hf->device);
    return HF_OK;
}

static int check_flags(const char *fn, uint32_t flags) {
    if ((flags & HF_RAY_DETACHSHAPE) && (flags & HF_RAY_FOLLOWSHAPE))
        return fail(HF_EFLAGS, "%s: Invalid combination of RayFlags: DetachShape | FollowShape", fn);
    return HF_OK;
}

static int check_pi(const char *fn, size_t n, const hf_pi_const_t *pi) {
    if (!pi) return fail(HF_EINVAL, "%s: NULL pi", fn);
    if (n && (!pi->t || !pi->prim_uv[0] || !pi->prim_uv[1] || !pi->prim_index))
        return fail(HF_EINVAL, "%s: NULL preliminary-intersection array", fn);
    return HF_OK;
}

extern "C" int hf_ray_intersect_preliminary(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                                            const uint8_t *active, const hf_pi_t *out, hf_stream_t stream) {
    int rc = check_rays("hf_ray_intersect_preliminary", hf, n, rays);
    if (rc) return rc;
    if (!out || (n && !out->t)) return fail(HF_EINVAL, "hf_ray_intersect_preliminary: NULL output");
    {
        slot_lease lease(hf, (hipStream_t) stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "trace launch: %s", lease.why);
        hf_launch_trace(0, hf->dev, n, rays, active, out, nullptr, nullptr, 0, lease.buf, (hipStream_t) stream, nullptr,
                        hf->coherence == HF_COHERENCE_INCOHERENT);
    }
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_ray_test(const hf_field_t *hf, size_t n, const hf_rays_t *rays, const uint8_t *active,
                           uint8_t *out_hit, hf_stream_t stream) {
    int rc = check_rays("hf_ray_test", hf, n, rays);
    if (rc) return rc;
    if (n && !out_hit) return fail(HF_EINVAL, "hf_ray_test: NULL output");
    {
        slot_lease lease(hf, (hipStream_t) stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "trace launch: %s", lease.why);
        hf_launch_trace(1, hf->dev, n, rays, active, nullptr, out_hit, nullptr, 0, lease.buf, (hipStream_t) stream, nullptr,
                        hf->coherence == HF_COHERENCE_INCOHERENT);
    }
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_compute_surface_interaction(const hf_field_t *hf, size_t n, const hf_rays_t *rays,
                                              const hf_pi_const_t *pi, uint32_t ray_flags, const uint8_t *active,
                                              const hf_si_t *out, hf_stream_t stream) {
    int rc = check_rays("hf_compute_surface_interaction", hf, n, rays);
    if (rc) return rc;
    if ((rc = check_flags("hf_compute_surface_interaction", ray_flags))) return rc;
    if ((rc = check_pi("hf_compute_surface_interaction", n, pi))) return rc;
    if (!out) return fail(HF_EINVAL, "hf_compute_surface_interaction: NULL output");
    hf_launch_si(hf->dev, n, rays, pi, active, out, ray_flags, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_ray_intersect(const hf_field_t *hf, size_t n, const hf_rays_t *rays, uint32_t ray_flags,
                                const uint8_t *active, const hf_pi_t *out_pi, const hf_si_t *out_si,
                                hf_stream_t stream) {
    int rc = check_rays("hf_ray_intersect", hf, n, rays);
    if (rc) return rc;
    if ((rc = check_flags("hf_ray_intersect", ray_flags))) return rc;
    if (!out_si) return fail(HF_EINVAL, "hf_ray_intersect: NULL output");
    {
        slot_lease lease(hf, (hipStream_t) stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "trace launch: %s", lease.why);
        hf_launch_trace(2, hf->dev, n, rays, active, out_pi, nullptr, out_si, ray_flags, lease.buf, (hipStream_t) stream, nullptr,
                        hf->coherence == HF_COHERENCE_INCOHERENT);
    }
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_adjoint_rows(const hf_field_t *hf, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                               uint32_t ray_flags, const uint8_t *active, const hf_si_grad_t *grad_si,
                               float *grad_heights, float *const grad_o[3], float *const grad_d[3],
                               uint32_t *row_band, hf_stream_t stream) {
    int rc = check_rays("hf_adjoint", hf, n, rays);
    if (rc) return rc;
    if ((rc = check_flags("hf_adjoint", ray_flags))) return rc;
    if ((rc = check_pi("hf_adjoint", n, pi))) return rc;
    if (!grad_si) return fail(HF_EINVAL, "hf_adjoint: NULL grad_si");
    if (grad_o && (!grad_o[0] || !grad_o[1] || !grad_o[2])) return fail(HF_EINVAL, "hf_adjoint: NULL grad_o array");
    if (grad_d && (!grad_d[0] || !grad_d[1] || !grad_d[2])) return fail(HF_EINVAL, "hf_adjoint: NULL grad_d array");
    hf_launch_adjoint(hf->dev, n, rays, pi, active, grad_si, ray_flags, grad_heights, grad_o, grad_d, row_band,
                      (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_adjoint(const hf_field_t *hf, size_t n, const hf_rays_t *rays, const hf_pi_const_t *pi,
                          uint32_t ray_flags, const uint8_t *active, const hf_si_grad_t *grad_si,
                          float *grad_heights, float *const grad_o[3], float *const grad_d[3],
                          hf_stream_t stream) {
    return hf_adjoint_rows(hf, n, rays, pi, ray_flags, active, grad_si, grad_heights, grad_o, grad_d, nullptr, stream);
}

// ---- minimal direct lighting --------------------------------------------------
static int pack_lights(const char *who, size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                       const float *t, uint32_t n_lights, const hf_dir_light_t *lights, float albedo,
                       const uint8_t *const *vis, hf_lights_dev &L) {
    if (!sh_n || !d || !t || !lights) return fail(HF_EINVAL, "%s: NULL argument", who);
    for (int k = 0; k < 3; ++k)
        if (!sh_n[k] || !d[k]) return fail(HF_EINVAL, "%s: NULL component array", who);
    if (spp == 0 || n % spp != 0) return fail(HF_EINVAL, "%s: n (%zu) must be a multiple of spp (%u)", who, n, spp);
    if (n_lights == 0 || n_lights > HF_MAX_LIGHTS)
        return fail(HF_EINVAL, "%s: 1..%d lights supported (got %u)", who, HF_MAX_LIGHTS, n_lights);
    if ((size_t) (n / spp) * n_lights >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "%s: image too large", who);
    L.n = n_lights;
    L.weight = nullptr; L.grad_weight = nullptr;
    for (uint32_t k = 0; k < HF_MAX_LIGHTS; ++k) {
        const bool on = k < n_lights;
        for (int c = 0; c < 3; ++c) L.l[k][c] = on ? lights[k].to_light[c] : 0.f;
        L.w[k] = on ? (albedo * 0.31830988618379067154f) * lights[k].irradiance : 0.f; // dr::InvPi
        L.vis[k] = (on && vis) ? vis[k] : nullptr;
    }
    return HF_OK;
}

extern "C" int hf_direct_lighting_weighted(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                           const float *t, const float *weight, uint32_t n_lights,
                                           const hf_dir_light_t *lights, float albedo, const uint8_t *const *vis,
                                           float *image, hf_stream_t stream) {
    hf_lights_dev L;
    const int rc = pack_lights("hf_direct_lighting", n, spp, sh_n, d, t, n_lights, lights, albedo, vis, L);
    if (rc != HF_OK) return rc;
    if (!image) return fail(HF_EINVAL, "hf_direct_lighting: NULL image");
    L.weight = weight;
    hf_launch_direct(n, spp, sh_n, d, t, nullptr, L, image, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}
extern "C" int hf_direct_lighting(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                  const float *t, uint32_t n_lights, const hf_dir_light_t *lights, float albedo,
                                  const uint8_t *const *vis, float *image, hf_stream_t stream) {
    return hf_direct_lighting_weighted(n, spp, sh_n, d, t, nullptr, n_lights, lights, albedo, vis, image, stream);
}
extern "C" int hf_direct_lighting_weighted_adjoint(size_t n, uint32_t spp, const float *const sh_n[3],
                                                   const float *const d[3], const float *t, const float *weight,
                                                   uint32_t n_lights, const hf_dir_light_t *lights, float albedo,
                                                   const uint8_t *const *vis, const float *grad_image,
                                                   float *const grad_sh_n[3], float *grad_weight, hf_stream_t stream) {
    hf_lights_dev L;
    const int rc = pack_lights("hf_direct_lighting_adjoint", n, spp, sh_n, d, t, n_lights, lights, albedo, vis, L);
    if (rc != HF_OK) return rc;
    if (!grad_image || !grad_sh_n || !grad_sh_n[0] || !grad_sh_n[1] || !grad_sh_n[2])
        return fail(HF_EINVAL, "hf_direct_lighting_adjoint: NULL gradient array");
    L.weight = weight; L.grad_weight = grad_weight;
    hf_launch_direct_adjoint(n, spp, sh_n, d, t, nullptr, L, grad_image, grad_sh_n, nullptr, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}
extern "C" int hf_direct_lighting_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                          const float *t, uint32_t n_lights, const hf_dir_light_t *lights,
                                          float albedo, const uint8_t *const *vis, const float *grad_image,
                                          float *const grad_sh_n[3], hf_stream_t stream) {
    return hf_direct_lighting_weighted_adjoint(n, spp, sh_n, d, t, nullptr, n_lights, lights, albedo, vis, grad_image,
                                               grad_sh_n, nullptr, stream);
}

static bool all3(const float *const p[3]) { return p && p[0] && p[1] && p[2]; }

// point lights: same packing (hf_point_light_t has hf_dir_light_t's layout: three floats + one)
extern "C" int hf_point_lighting(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                 const float *t, const float *const p[3], uint32_t n_lights,
                                 const hf_point_light_t *lights, float albedo, const uint8_t *const *vis, float *image,
                                 hf_stream_t stream) {
    static_assert(sizeof(hf_point_light_t) == sizeof(hf_dir_light_t), "light structs share one packing");
    hf_lights_dev L;
    const int rc = pack_lights("hf_point_lighting", n, spp, sh_n, d, t, n_lights, (const hf_dir_light_t *) lights, albedo, vis, L);
    if (rc != HF_OK) return rc;
    if (!all3(p)) return fail(HF_EINVAL, "hf_point_lighting: NULL position array");
    if (!image) return fail(HF_EINVAL, "hf_point_lighting: NULL image");
    hf_launch_direct(n, spp, sh_n, d, t, p, L, image, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_point_lighting_adjoint(size_t n, uint32_t spp, const float *const sh_n[3], const float *const d[3],
                                         const float *t, const float *const p[3], uint32_t n_lights,
                                         const hf_point_light_t *lights, float albedo, const uint8_t *const *vis,
                                         const float *grad_image, float *const grad_sh_n[3], float *const grad_p[3],
                                         hf_stream_t stream) {
    hf_lights_dev L;
    const int rc = pack_lights("hf_point_lighting_adjoint", n, spp, sh_n, d, t, n_lights, (const hf_dir_light_t *) lights,
                               albedo, vis, L);
    if (rc != HF_OK) return rc;
    if (!all3(p)) return fail(HF_EINVAL, "hf_point_lighting_adjoint: NULL position array");
    if (!grad_image || !grad_sh_n || !grad_sh_n[0] || !grad_sh_n[1] || !grad_sh_n[2] || !grad_p || !grad_p[0] ||
        !grad_p[1] || !grad_p[2])
        return fail(HF_EINVAL, "hf_point_lighting_adjoint: NULL gradient array");
    hf_launch_direct_adjoint(n, spp, sh_n, d, t, p, L, grad_image, grad_sh_n, grad_p, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

// ---- Gaussian reconstruction filter (film) ------------------------------------------------------
static int splat_args(const char *who, size_t n, uint32_t channels, const float *pos_x, const float *pos_y,
                      uint32_t width, uint32_t height, float stddev, hf_splat_args &a) {
    if (!pos_x || !pos_y) return fail(HF_EINVAL, "%s: NULL film positions", who);
    if (channels == 0 || channels > HF_MAX_LIGHTS) return fail(HF_EINVAL, "%s: 1..%d channels (got %u)", who, HF_MAX_LIGHTS, channels);
    if (width == 0 || height == 0 || (size_t) width * height >= ((size_t) 1 << 31)) return fail(HF_EINVAL, "%s: bad film size", who);
    if (!(stddev > 0.f) || stddev > 1.f) return fail(HF_EINVAL, "%s: stddev must be in (0, 1] pixels", who);
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "%s: more than 2^32 samples", who);
    a = {};
    a.n = n; a.channels = channels; a.width = width; a.height = height; a.pos_x = pos_x; a.pos_y = pos_y;
    a.radius = 4.f * stddev;                               // gaussian.cpp:52-53
    a.alpha = -1.f / (2.f * stddev * stddev);              // gaussian.cpp:97-99
    a.bias = expf(a.alpha * a.radius * a.radius);
    return HF_OK;
}

extern "C" int hf_film_splat(size_t n, uint32_t channels, const float *const *values, const float *pos_x,
                             const float *pos_y, uint32_t width, uint32_t height, float stddev, float *image,
                             float *weight, hf_stream_t stream) {
    hf_splat_args a;
    const int rc = splat_args("hf_film_splat", n, channels, pos_x, pos_y, width, height, stddev, a);
    if (rc != HF_OK) return rc;
    if (!values || !image || !weight) return fail(HF_EINVAL, "hf_film_splat: NULL argument");
    for (uint32_t k = 0; k < channels; ++k) {
        if (!values[k]) return fail(HF_EINVAL, "hf_film_splat: NULL channel array");
        a.values[k] = values[k];
    }
    a.image = image; a.weight = weight;
    hf_launch_film_splat(a, false, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_film_splat_adjoint(size_t n, uint32_t channels, const float *pos_x, const float *pos_y,
                                     uint32_t width, uint32_t height, float stddev, const float *grad_image,
                                     float *const *grad_values, hf_stream_t stream) {
    hf_splat_args a;
    const int rc = splat_args("hf_film_splat_adjoint", n, channels, pos_x, pos_y, width, height, stddev, a);
    if (rc != HF_OK) return rc;
    if (!grad_image || !grad_values) return fail(HF_EINVAL, "hf_film_splat_adjoint: NULL argument");
    for (uint32_t k = 0; k < channels; ++k) {
        if (!grad_values[k]) return fail(HF_EINVAL, "hf_film_splat_adjoint: NULL channel array");
        a.grad_values[k] = grad_values[k];
    }
    a.grad_image = grad_image;
    hf_launch_film_splat(a, true, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

// ---- warped-area reparameterisation --------------------------------------------

extern "C" int hf_reparam_aux_rays(size_t n, const float *const o[3], const float *const d[3], const uint8_t *active,
                                   uint32_t k, float kappa, int antithetic, uint32_t seed, const uint32_t *ray_id,
                                   float *const aux_d[3], float *aux_maxt, hf_stream_t stream) {
    if (!all3(o) || !all3(d) || !aux_d || !aux_d[0] || !aux_d[1] || !aux_d[2] || !aux_maxt)
        return fail(HF_EINVAL, "hf_reparam_aux_rays: NULL argument");
    if (!(kappa > 0.f)) return fail(HF_EINVAL, "hf_reparam_aux_rays: kappa must be > 0");
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "hf_reparam_aux_rays: more than 2^32 rays");
    hf_reparam_args a = {};
    a.n = n; a.active = active; a.k = k; a.seed = seed; a.kappa = kappa; a.antithetic = antithetic; a.ray_id = ray_id;
    for (int c = 0; c < 3; ++c) { a.o[c] = o[c]; a.d[c] = d[c]; a.aux_d[c] = aux_d[c]; }
    a.aux_maxt = aux_maxt;
    hf_launch_reparam_aux(a, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_reparam_weights(int mode, size_t n, const float *const o[3], const float *const d[3],
                                  const uint8_t *active, uint32_t k, float kappa, float exponent, int antithetic,
                                  uint32_t seed, const uint32_t *ray_id, const float *si_t, const float *const si_p[3],
                                  const float *si_boundary_test, float *Z, float *const dZ[3],
                                  const float *const grad_direction[3], const float *grad_divergence,
                                  float *const grad_p[3], float *grad_t, float *const grad_vd[3],
                                  hf_stream_t stream) {
    if (!all3(o) || !all3(d) || !si_t || !si_boundary_test || !Z || !dZ || !dZ[0] || !dZ[1] || !dZ[2])
        return fail(HF_EINVAL, "hf_reparam_weights: NULL argument");
    if (mode != 0 && mode != 1) return fail(HF_EINVAL, "hf_reparam_weights: mode must be 0 or 1");
    if (mode == 1 && (!all3(si_p) || !all3(grad_direction) || !grad_divergence || !grad_p || !grad_p[0] || !grad_p[1] ||
                      !grad_p[2] || !grad_t))
        return fail(HF_EINVAL, "hf_reparam_weights: mode 1 needs si_p, grad_direction, grad_divergence, grad_p, grad_t");
    if (!(kappa > 0.f)) return fail(HF_EINVAL, "hf_reparam_weights: kappa must be > 0");
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "hf_reparam_weights: more than 2^32 rays");
    hf_reparam_args a = {};
    a.n = n; a.active = active; a.k = k; a.seed = seed; a.kappa = kappa; a.exponent = exponent;
    a.antithetic = antithetic; a.mode = mode; a.ray_id = ray_id;
    a.si_t = si_t; a.si_bt = si_boundary_test; a.Z = Z; a.g_div = grad_divergence; a.g_t = grad_t;
    for (int c = 0; c < 3; ++c) {
        a.o[c] = o[c]; a.d[c] = d[c]; a.dZ[c] = dZ[c];
        a.si_p[c] = si_p ? si_p[c] : nullptr;
        a.g_dir[c] = grad_direction ? grad_direction[c] : nullptr;
        a.g_p[c] = grad_p ? grad_p[c] : nullptr;
        a.g_vd[c] = (grad_vd && grad_vd[0] && grad_vd[1] && grad_vd[2]) ? grad_vd[c] : nullptr;
    }
    hf_launch_reparam_weights(a, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

// Which instantiation traces auxiliary rays: the lean one (every wave walks per lane) when the handle is set to
// incoherent rays, and in the automatic mode when kappa is below HF_AUX_LEAN_KAPPA -- the reference traces its auxiliary
// rays with coherent = false (reparam.py:95), and on the bench wavefront (camera 2.6 units from the terrain, cells of
// 1/2048 unit) the lean kernel wins up to kappa = 1.6e6 (15.7 vs 16.5 ms), ties at 6.4e6 and loses beyond (1e8, where the
// auxiliary rays are the primary ray: 13.9 vs 11.1 ms; profiles/r04_ab/r04_lean).
#ifndef HF_AUX_LEAN_KAPPA
#define HF_AUX_LEAN_KAPPA 4e6f
#endif
static bool aux_lean(const hf_field_t *hf, float kappa) {
    if (hf->coherence == HF_COHERENCE_INCOHERENT) return true;
    if (hf->coherence == HF_COHERENCE_COHERENT) return false;
    return kappa < HF_AUX_LEAN_KAPPA;
}

extern "C" int hf_set_ray_coherence(hf_field_t *hf, int coherence) {
    if (!hf) return fail(HF_EINVAL, "hf_set_ray_coherence: NULL handle");
    if (coherence != HF_COHERENCE_AUTO && coherence != HF_COHERENCE_INCOHERENT && coherence != HF_COHERENCE_COHERENT)
        return fail(HF_EINVAL, "hf_set_ray_coherence: unknown mode %d", coherence);
    hf->coherence = coherence;
    return HF_OK;
}
extern "C" int hf_get_ray_coherence(const hf_field_t *hf) { return hf ? hf->coherence : HF_COHERENCE_AUTO; }

extern "C" int hf_reparam_trace(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                                const uint8_t *active, uint32_t k, float kappa, int antithetic, uint32_t seed,
                                const uint32_t *ray_id, const hf_pi_t *out_pi, const hf_si_t *out_si, hf_stream_t stream) {
    if (!all3(o) || !all3(d)) return fail(HF_EINVAL, "hf_reparam_trace: NULL argument");
    hf_rays_t rays; // maxt is not read for auxiliary rays (infinity); any readable array of n floats will do
    for (int c = 0; c < 3; ++c) { rays.o[c] = o[c]; rays.d[c] = d[c]; }
    rays.maxt = o[0];
    int rc = check_rays("hf_reparam_trace", hf, n, &rays);
    if (rc) return rc;
    if (!out_si) return fail(HF_EINVAL, "hf_reparam_trace: NULL output");
    if (!(kappa > 0.f)) return fail(HF_EINVAL, "hf_reparam_trace: kappa must be > 0");
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "hf_reparam_trace: more than 2^32 rays");
    hf_reparam_args a = {};
    a.k = k; a.seed = seed; a.kappa = kappa; a.antithetic = antithetic; a.ray_id = ray_id;
    {
        slot_lease lease(hf, (hipStream_t) stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "trace launch: %s", lease.why);
        hf_launch_trace(2, hf->dev, n, &rays, active, out_pi, nullptr, out_si,
                        HF_RAY_ALL | HF_RAY_FOLLOWSHAPE | HF_RAY_BOUNDARYTEST, lease.buf, (hipStream_t) stream, &a, aux_lean(hf, kappa));
    }
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_reparam_trace_all(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                                    const uint8_t *active, uint32_t num_rays, float kappa, int antithetic, uint32_t seed,
                                    const uint32_t *ray_id, const hf_pi_t *out_pi, const hf_si_t *out_si, size_t sample_stride,
                                    hf_stream_t stream) {
    if (!all3(o) || !all3(d)) return fail(HF_EINVAL, "hf_reparam_trace_all: NULL argument");
    hf_rays_t rays; // maxt is not read for auxiliary rays (infinity); any readable array of n floats will do
    for (int c = 0; c < 3; ++c) { rays.o[c] = o[c]; rays.d[c] = d[c]; }
    rays.maxt = o[0];
    int rc = check_rays("hf_reparam_trace_all", hf, n, &rays);
    if (rc) return rc;
    if (!out_si) return fail(HF_EINVAL, "hf_reparam_trace_all: NULL output");
    if (!(kappa > 0.f)) return fail(HF_EINVAL, "hf_reparam_trace_all: kappa must be > 0");
    if (num_rays == 0 || num_rays > 32) return fail(HF_EINVAL, "hf_reparam_trace_all: 1..32 auxiliary rays per ray (got %u)", num_rays);
    if (num_rays > 1 && sample_stride < n) return fail(HF_EINVAL, "hf_reparam_trace_all: sample_stride < n");
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "hf_reparam_trace_all: more than 2^32 rays");
    hf_reparam_args a = {};
    a.k = 0; a.num = num_rays; a.stride = sample_stride; a.seed = seed; a.kappa = kappa; a.antithetic = antithetic; a.ray_id = ray_id;
    {
        slot_lease lease(hf, (hipStream_t) stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "trace launch: %s", lease.why);
        hf_launch_trace(2, hf->dev, n, &rays, active, out_pi, nullptr, out_si,
                        HF_RAY_ALL | HF_RAY_FOLLOWSHAPE | HF_RAY_BOUNDARYTEST, lease.buf, (hipStream_t) stream, &a, aux_lean(hf, kappa));
    }
    HF_HIP(hipGetLastError());
    return HF_OK;
}

extern "C" int hf_reparam_backward(const hf_field_t *hf, size_t n, const float *const o[3], const float *const d[3],
                                   const uint8_t *active, uint32_t num_rays, float kappa, float exponent,
                                   int antithetic, uint32_t seed, const uint32_t *ray_id, const hf_pi_const_t *pi,
                                   const float *si_boundary_test, size_t sample_stride,
                                   const float *const grad_direction[3], const float *grad_divergence,
                                   float *grad_heights, hf_stream_t stream) {
    if (!hf) return fail(HF_EINVAL, "hf_reparam_backward: NULL handle");
    if (!all3(o) || !all3(d) || !si_boundary_test || !all3(grad_direction) || !grad_divergence || !grad_heights)
        return fail(HF_EINVAL, "hf_reparam_backward: NULL argument");
    int rc = check_pi("hf_reparam_backward", n, pi);
    if (rc) return rc;
    if (!(kappa > 0.f)) return fail(HF_EINVAL, "hf_reparam_backward: kappa must be > 0");
    if (num_rays == 0 || num_rays > 32) return fail(HF_EINVAL, "hf_reparam_backward: 1..32 auxiliary rays per ray (got %u)", num_rays);
    if (num_rays > 1 && sample_stride < n) return fail(HF_EINVAL, "hf_reparam_backward: sample_stride < n");
    if (n >= ((size_t) 1 << 32)) return fail(HF_EINVAL, "hf_reparam_backward: more than 2^32 rays");
    int cur = -1; // like the other query functions: the launch goes to the caller's current device
    if (hipGetDevice(&cur) != hipSuccess || cur != hf->device)
        return fail(HF_EDEVICE, "hf_reparam_backward: current HIP device is %d, the heightfield lives on device %d", cur,
                    hf->device);
    hf_reparam_args a = {};
    a.n = n; a.active = active; a.seed = seed; a.kappa = kappa; a.exponent = exponent; a.antithetic = antithetic;
    a.ray_id = ray_id;
    a.si_bt = si_boundary_test; a.g_div = grad_divergence;
    for (int c = 0; c < 3; ++c) { a.o[c] = o[c]; a.d[c] = d[c]; a.g_dir[c] = grad_direction[c]; }
    hf_launch_reparam_backward(hf->dev, a, num_rays, sample_stride, pi, grad_heights, (hipStream_t) stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

// ---- scalar / packet entry: host pointers, per-thread staging ------------------------
namespace {
// layout of the staging block, in floats: 7 ray rows, t, u, v, prim (u32), then HF_PACKET_MAX bytes of mask / hit
enum { PK_ROWS = 11, PK_FLOATS = PK_ROWS * HF_PACKET_MAX, PK_BYTES = PK_FLOATS * 4 + 2 * HF_PACKET_MAX };
struct packet_stage {
    int device = -1;
    hipStream_t stream = nullptr;
    char *h = nullptr, *d = nullptr; // pinned host mirror, device block
    void release() {
        if (device < 0) return;
        hf_device_guard guard(device);
        if (stream) (void) hipStreamDestroy(stream);
        if (h) (void) hipHostFree(h);
        if (d) (void) hipFree(d);
        device = -1; stream = nullptr; h = d = nullptr;
    }
    ~packet_stage() { release(); }
};
thread_local packet_stage g_stage;

int stage_for(const char *fn, int device, packet_stage **out) {
    packet_stage &s = g_stage;
    if (s.device != device) {
        s.release();
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc((void **) &s.h, PK_BYTES, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **) &s.d, PK_BYTES);
        if (e != hipSuccess) {
            s.device = device; // so that release() frees what exists
            s.release();
            return fail(e == hipErrorOutOfMemory ? HF_ENOMEM : HF_EDEVICE, "%s: %s", fn, hipGetErrorString(e));
        }
        s.device = device;
    }
    *out = &s;
    return HF_OK;
}

// mode 0: closest hit, mode 1: any hit
int packet_trace(const char *fn, int mode, const hf_field_t *hf, uint32_t n, const float *const h_o[3],
                 const float *const h_d[3], const float *h_maxt, const uint8_t *h_active, float *h_t,
                 float *const h_uv[2], uint32_t *h_prim, uint8_t *h_hit) {
    if (!hf || !h_o || !h_d || !h_maxt) return fail(HF_EINVAL, "%s: NULL argument", fn);
    if (n == 0) return HF_OK;
    if (n > HF_PACKET_MAX) return fail(HF_EINVAL, "%s: packet of %u rays (at most %d)", fn, n, HF_PACKET_MAX);
    for (int k = 0; k < 3; ++k)
        if (!h_o[k] || !h_d[k]) return fail(HF_EINVAL, "%s: NULL ray component array", fn);
    hf_device_guard guard(hf->device);
    if (!guard.ok) return fail(HF_EDEVICE, "%s: cannot select device %d", fn, hf->device);
    packet_stage *st;
    int rc = stage_for(fn, hf->device, &st);
    if (rc) return rc;
    float *hf32 = (float *) st->h;
    for (int k = 0; k < 3; ++k) {
        memcpy(hf32 + (size_t) k * HF_PACKET_MAX, h_o[k], sizeof(float) * n);
        memcpy(hf32 + (size_t) (3 + k) * HF_PACKET_MAX, h_d[k], sizeof(float) * n);
    }
    memcpy(hf32 + 6 * HF_PACKET_MAX, h_maxt, sizeof(float) * n);
    uint8_t *hmask = (uint8_t *) (st->h + PK_FLOATS * 4), *hhit = hmask + HF_PACKET_MAX;
    for (uint32_t k = 0; k < n; ++k) hmask[k] = h_active ? h_active[k] : 1;
    float *d32 = (float *) st->d;
    uint8_t *dmask = (uint8_t *) (st->d + PK_FLOATS * 4), *dhit = dmask + HF_PACKET_MAX;
    HF_HIP(hipStreamWaitEvent(st->stream, hf->built, 0)); // the last rebuild of the acceleration data
    HF_HIP(hipMemcpyAsync(st->d, st->h, PK_BYTES, hipMemcpyHostToDevice, st->stream));
    hf_rays_t rays;
    for (int k = 0; k < 3; ++k) { rays.o[k] = d32 + (size_t) k * HF_PACKET_MAX; rays.d[k] = d32 + (size_t) (3 + k) * HF_PACKET_MAX; }
    rays.maxt = d32 + 6 * HF_PACKET_MAX;
    hf_pi_t pi;
    pi.t = d32 + 7 * HF_PACKET_MAX; pi.prim_uv[0] = d32 + 8 * HF_PACKET_MAX; pi.prim_uv[1] = d32 + 9 * HF_PACKET_MAX;
    pi.prim_index = (uint32_t *) (d32 + 10 * HF_PACKET_MAX);
    {
        slot_lease lease(hf, st->stream, hf_trace_scratch_bytes(n));
        if (!lease.buf) return fail(lease.code, "%s: %s", fn, lease.why);
        hf_launch_trace(mode, hf->dev, n, &rays, dmask, mode == 0 ? &pi : nullptr, mode == 1 ? dhit : nullptr, nullptr, 0,
                        lease.buf, st->stream);
    }
    HF_HIP(hipGetLastError());
    HF_HIP(hipMemcpyAsync(st->h, st->d, PK_BYTES, hipMemcpyDeviceToHost, st->stream));
    HF_HIP(hipStreamSynchronize(st->stream));
    if (mode == 0) {
        if (h_t) memcpy(h_t, hf32 + 7 * HF_PACKET_MAX, sizeof(float) * n);
        if (h_uv && h_uv[0]) memcpy(h_uv[0], hf32 + 8 * HF_PACKET_MAX, sizeof(float) * n);
        if (h_uv && h_uv[1]) memcpy(h_uv[1], hf32 + 9 * HF_PACKET_MAX, sizeof(float) * n);
        if (h_prim) memcpy(h_prim, hf32 + 10 * HF_PACKET_MAX, sizeof(uint32_t) * n);
    } else {
        memcpy(h_hit, hhit, n);
    }
    return HF_OK;
}
} // namespace

extern "C" int hf_ray_intersect_preliminary_packet(const hf_field_t *hf, uint32_t n, const float *const h_o[3],
                                                   const float *const h_d[3], const float *h_maxt,
                                                   const uint8_t *h_active, float *h_t, float *const h_prim_uv[2],
                                                   uint32_t *h_prim_index) {
    if (n && !h_t) return fail(HF_EINVAL, "hf_ray_intersect_preliminary_packet: NULL output");
    return packet_trace("hf_ray_intersect_preliminary_packet", 0, hf, n, h_o, h_d, h_maxt, h_active, h_t, h_prim_uv,
                        h_prim_index, nullptr);
}

extern "C" int hf_ray_test_packet(const hf_field_t *hf, uint32_t n, const float *const h_o[3],
                                  const float *const h_d[3], const float *h_maxt, const uint8_t *h_active,
                                  uint8_t *h_hit) {
    if (n && !h_hit) return fail(HF_EINVAL, "hf_ray_test_packet: NULL output");
    return packet_trace("hf_ray_test_packet", 1, hf, n, h_o, h_d, h_maxt, h_active, nullptr, nullptr, nullptr, h_hit);
}

// ---- multi-GPU: the one collective of the path --------------------------------------------
namespace {
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);
nccl_allreduce_fn g_allreduce = nullptr;
nccl_errstr_fn g_nccl_errstr = nullptr;
std::once_flag g_nccl_once;
void bind_rccl() {
    // the host's own RCCL first (communicator and call must come from one library), else the system library
    void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    void *lib = nullptr;
    if (!sym) {
        const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char *nm : names)
            if ((lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (lib) sym = dlsym(lib, "ncclAllReduce");
    }
    g_allreduce = (nccl_allreduce_fn) sym;
    g_nccl_errstr = (nccl_errstr_fn) (lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
}
} // namespace

extern "C" int hf_allreduce_grad(float *d_grad, size_t count, void *rccl_comm, hf_stream_t stream) {
    if (!d_grad || !rccl_comm) return fail(HF_EINVAL, "hf_allreduce_grad: NULL argument");
    if (count == 0) return HF_OK;
    std::call_once(g_nccl_once, bind_rccl);
    if (!g_allreduce) return fail(HF_EDEVICE, "hf_allreduce_grad: RCCL (ncclAllReduce) not found in the process or as librccl.so");
    // ncclFloat32 = 7, ncclSum = 0 (rccl.h: ncclDataType_t / ncclRedOp_t)
    const int rc = g_allreduce(d_grad, d_grad, count, 7, 0, rccl_comm, (hipStream_t) stream);
    if (rc != 0)
        return fail(HF_EDEVICE, "hf_allreduce_grad: ncclAllReduce failed: %s", g_nccl_errstr ? g_nccl_errstr(rc) : "error");
    return HF_OK;
}
