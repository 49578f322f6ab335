// host_loop.cpp -- the C ABI of include/hf.h driven from plain C++ (no Python, no torch): the host side a
// compiled caller such as the Mitsuba adapter plugin would write.  A small inverse problem:
// recover a heightfield from K directionally lit orthographic images + depth,
//   trace (hf_ray_intersect) -> shade (hf_direct_lighting) -> image-space loss gradient (host, tiny) ->
//   hf_direct_lighting_adjoint -> hf_adjoint (dL/dheight) -> hf_adam_step (update + rebuild),
// every array a caller-owned device buffer, every call on one stream.  Prints the loss; exit code 0 when the
// loss dropped by 5x.      build: see __graft_entry__.build();   run: examples/host_loop [grid film spp steps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../include/hf.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define HF(x) do { int rc_ = (x); if (rc_ != HF_OK) { fprintf(stderr, "%s: %s\n", #x, hf_last_error_string()); exit(3); } } while (0)

static float *dev_floats(size_t n) { float *p; CK(hipMalloc((void **) &p, sizeof(float) * n)); CK(hipMemset(p, 0, sizeof(float) * n)); return p; }

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64, film = argc > 2 ? atoi(argv[2]) : 96, spp = argc > 3 ? atoi(argv[3]) : 4,
              steps = argc > 4 ? atoi(argv[4]) : 60;
    const size_t npix = (size_t) film * film, n = npix * spp;
    const int K = 3;
    hf_dir_light_t lights[K] = { { { 0.5f, 0.2f, 0.84f }, 3.14159265f }, { { -0.5f, 0.3f, 0.81f }, 3.14159265f }, { { 0.f, 0.f, 1.f }, 3.14159265f } };
    for (auto &l : lights) { const float s = 1.f / sqrtf(l.to_light[0] * l.to_light[0] + l.to_light[1] * l.to_light[1] + l.to_light[2] * l.to_light[2]); for (float &c : l.to_light) c *= s; }

    // target heights (the sine field of the benchmark) and the flat start
    std::vector<float> h_target((size_t) N * N), h_start((size_t) N * N, 0.5f);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            const double u = (double) j / (N - 1), v = (double) i / (N - 1);
            h_target[(size_t) i * N + j] = (float) (0.5 + 0.25 * sin(2 * M_PI * 4 * u) * cos(2 * M_PI * 4 * v) + 0.125 * sin(2 * M_PI * 7 * (u + v)));
        }
    // orthographic rays looking down at 17 degrees off vertical, stratified samples (SoA, 7 arrays)
    std::vector<float> r(7 * n);
    const double dir[3] = { -0.28, -0.16, -0.947 };
    for (size_t i = 0; i < n; ++i) {
        const size_t pix = i / spp, s = i % spp, py = pix / film, px = pix % film;
        const double sx = (px + (s % 2 + 0.5) / 2.0) / film, sy = (py + (s / 2 % 2 + 0.5) / 2.0) / film;
        const double x = 0.9 * (2 * sx - 1), y = 0.9 * (2 * sy - 1);
        r[0 * n + i] = (float) (x - 2.0 * dir[0]); r[1 * n + i] = (float) (y - 2.0 * dir[1]); r[2 * n + i] = (float) (0.25 - 2.0 * dir[2]);
        r[3 * n + i] = (float) dir[0]; r[4 * n + i] = (float) dir[1]; r[5 * n + i] = (float) dir[2];
        r[6 * n + i] = INFINITY;
    }
    float *d_r = dev_floats(7 * n);
    CK(hipMemcpy(d_r, r.data(), sizeof(float) * 7 * n, hipMemcpyHostToDevice));
    hf_rays_t rays = { { d_r, d_r + n, d_r + 2 * n }, { d_r + 3 * n, d_r + 4 * n, d_r + 5 * n }, d_r + 6 * n };

    hipStream_t st; CK(hipStreamCreate(&st));
    hf_desc_t desc = {};
    desc.width = desc.height = (uint32_t) N; desc.max_height = 0.5f;
    const float eye[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    for (int k = 0; k < 12; ++k) desc.to_world[k] = eye[k];
    hf_field_t *hf = nullptr;
    HF(hf_create(&desc, &hf));

    // caller-owned device buffers: parameter, optimiser state, records, gradients
    float *d_h = dev_floats((size_t) N * N), *d_g = dev_floats((size_t) N * N), *d_m = dev_floats((size_t) N * N), *d_v = dev_floats((size_t) N * N);
    float *d_pi = dev_floats(4 * n), *d_si = dev_floats(18 * n), *d_gsi = dev_floats(18 * n), *d_img = dev_floats(K * npix), *d_gimg = dev_floats(K * npix);
    hf_pi_t pi = { d_pi, { d_pi + n, d_pi + 2 * n }, (uint32_t *) (d_pi + 3 * n) };
    hf_pi_const_t pic = { d_pi, { d_pi + n, d_pi + 2 * n }, (const uint32_t *) (d_pi + 3 * n) };
    hf_si_t si = {};
    si.t = d_si;
    for (int k = 0; k < 3; ++k) { si.p[k] = d_si + (1 + k) * n; si.n[k] = d_si + (4 + k) * n; si.sh_n[k] = d_si + (9 + k) * n; si.dp_du[k] = d_si + (12 + k) * n; si.dp_dv[k] = d_si + (15 + k) * n; }
    si.uv[0] = d_si + 7 * n; si.uv[1] = d_si + 8 * n;
    hf_si_grad_t gsi = {};                      // upstream gradient: only t and sh_n are non-zero here
    gsi.t = d_gsi;
    for (int k = 0; k < 3; ++k) gsi.sh_n[k] = d_gsi + (9 + k) * n;
    float *g_shn[3] = { d_gsi + 9 * n, d_gsi + 10 * n, d_gsi + 11 * n };
    const float *shn[3] = { si.sh_n[0], si.sh_n[1], si.sh_n[2] };

    std::vector<float> img(K * npix), tgt_img(K * npix), t(n), tgt_t(n), gimg(K * npix), gt(n);
    auto render = [&](void) {
        HF(hf_ray_intersect(hf, n, &rays, HF_RAY_ALL, nullptr, &pi, &si, st));
        HF(hf_direct_lighting(n, spp, shn, rays.d, si.t, K, lights, 1.0f, nullptr, d_img, st));
    };
    HF(hf_set_heights_host(hf, h_target.data(), st));
    render();
    CK(hipMemcpyAsync(tgt_img.data(), d_img, sizeof(float) * K * npix, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync(tgt_t.data(), si.t, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));

    CK(hipMemcpy(d_h, h_start.data(), sizeof(float) * N * N, hipMemcpyHostToDevice));
    HF(hf_set_heights(hf, d_h, st));
    double first = 0, last = 0;
    for (int it = 1; it <= steps; ++it) {
        render();
        CK(hipMemcpyAsync(img.data(), d_img, sizeof(float) * K * npix, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(t.data(), si.t, sizeof(float) * n, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        // loss = mean_pix sum_k (I - I*)^2 + 10 mean_samples (t - t*)^2 and its image-space gradient (host: tiny)
        double loss = 0;
        for (size_t q = 0; q < K * npix; ++q) { const double e = img[q] - tgt_img[q]; loss += e * e / npix; gimg[q] = (float) (2 * e / npix); }
        size_t both = 0;
        for (size_t i = 0; i < n; ++i) both += isfinite(t[i]) && isfinite(tgt_t[i]);
        for (size_t i = 0; i < n; ++i) {
            const bool ok = isfinite(t[i]) && isfinite(tgt_t[i]);
            const double e = ok ? t[i] - tgt_t[i] : 0.0;
            loss += 10 * e * e / both; gt[i] = (float) (20 * e / both);
        }
        if (it == 1) first = loss;
        last = loss;
        if (it % 10 == 0 || it == 1) printf("step %3d  loss %.6f\n", it, loss);
        CK(hipMemcpyAsync(d_gimg, gimg.data(), sizeof(float) * K * npix, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(d_gsi, gt.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        HF(hf_direct_lighting_adjoint(n, spp, shn, rays.d, si.t, K, lights, 1.0f, nullptr, d_gimg, g_shn, st));
        CK(hipMemsetAsync(d_g, 0, sizeof(float) * N * N, st));
        HF(hf_adjoint(hf, n, &rays, &pic, HF_RAY_ALL, nullptr, &gsi, d_g, nullptr, nullptr, st));
        HF(hf_adam_step(hf, d_h, d_g, d_m, d_v, 0.02, 0.9, 0.999, 1e-8, (uint32_t) it, 0, st));
    }
    CK(hipStreamSynchronize(st));
    std::vector<float> h_out((size_t) N * N);
    CK(hipMemcpy(h_out.data(), d_h, sizeof(float) * N * N, hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t q = 0; q < h_out.size(); ++q) err += fabs(h_out[q] - h_target[q]) / h_out.size();
    printf("loss %.6f -> %.6f, mean |h - h*| %.5f (start 0.17)\n", first, last, err);
    HF(hf_destroy(hf));
    return last < 0.2 * first ? 0 : 1;
}
