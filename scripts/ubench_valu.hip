// ubench_valu.hip -- what does one wave64 vector instruction cost on a gfx950 SIMD?
//
// Settles the question VERDICT r01 raised about DESIGN 4.1: a wave64 v_fma_f32 occupies its SIMD for
// 2 cycles (32-lane SIMD) or 4 (16-lane SIMD)?  Independent v_fma_f32 streams (8 accumulators, no
// dependency between consecutive instructions) at 1, 2, 4 and 8 resident waves per SIMD on every CU;
// reports cycles (s_memtime) per wave-instruction per SIMD and the wall-clock instruction rate.
// A dependent chain (1 accumulator) gives the back-to-back dependent-issue latency, and a
// variant interleaving scalar ALU work shows whether SALU issues beside VALU.
//
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o scratch_so/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define ITERS 65536 // long enough (tens of ms per launch) for the clocks to settle

// 8 independent chains x 8 = 64 v_fma_f32 per iteration
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float *out, long long *cyc, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int s0 = blockIdx.x, s1 = 3;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) { // independent
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (KIND == 1) { // dependent chain: 64 back-to-back dependent fmas
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(x0) : "v"(a), "v"(b));
            }
        } else if (KIND == 2) { // independent VALU interleaved 1:1 with SALU (64 + 64)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %1, %1, %10, %11\n s_xor_b32 %9, %9, %8\n"
                             "v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %3, %3, %10, %11\n s_xor_b32 %9, %9, %8\n"
                             "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %5, %5, %10, %11\n s_xor_b32 %9, %9, %8\n"
                             "v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %7, %7, %10, %11\n s_xor_b32 %9, %9, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s0), "+s"(s1)
                             : "v"(a), "v"(b) : "scc");
            }
        } else if (KIND == 3) { // independent IEEE-ish heavy op: v_rcp_f32 (transcendental unit)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        } else if (KIND == 5) { // packed: 2 floats per lane per instruction
            float2 p0 = make_float2(x0, x1), p1 = make_float2(x2, x3), p2 = make_float2(x4, x5), p3 = make_float2(x6, x7);
            float2 pa = make_float2(a, a), pb = make_float2(b, b);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            }
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (KIND == 6) { // the instruction mix of a slab/box test: fma, max3, min3, cmp, cndmask
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %0, %8\n v_min3_f32 %2, %2, %0, %9\n v_cmp_le_f32 vcc, %1, %2\n"
                             "v_cndmask_b32 %3, %3, %8, vcc\n v_sub_f32 %4, %4, %9\n v_mul_f32 %5, %5, %8\n v_add_f32 %6, %6, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            }
        } else if (KIND == 4) { // independent v_cmp + v_cndmask pairs (VOPC writes vcc: dependent pair)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float) (s0 + s1);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int cus, float *d_out, long long *d_cyc) {
    // 256 threads = 4 waves = one wave per SIMD of a CU; waves_per_simd workgroups per CU
    const int blocks = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0001f, 0.5f); // warm up
    hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * cyc.size(), hipMemcpyDeviceToHost));
    double avg = 0;
    for (long long c : cyc) avg += (double) c;
    avg /= (double) cyc.size();
    const double n_inst = 64.0 * ITERS; // vector instructions per wave
    // cycles a SIMD spends per wave-instruction = wave loop cycles / (instructions per wave x waves sharing the SIMD)
    printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"wave_cycles_per_vinst\": %.3f, \"simd_cycles_per_vinst\": %.3f, "
           "\"kernel_ms\": %.4f, \"vinst_per_simd_per_us\": %.1f, \"memtime_ticks_per_us\": %.1f}\n",
           name, waves_per_simd, avg / n_inst, avg / (n_inst * waves_per_simd), ms,
           n_inst * waves_per_simd / (ms * 1e3), avg / (ms * 1e3));
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", p.gcnArchName, cus, p.clockRate);
    fflush(stdout);
    float *d_out; long long *d_cyc;
    CHECK(hipMalloc((void **) &d_out, sizeof(float) * 256 * cus * 8));
    CHECK(hipMalloc((void **) &d_cyc, sizeof(long long) * 4 * cus * 8));
    const int wps[4] = { 1, 2, 4, 8 };
    for (int w : wps) run<0>("independent v_fma_f32", w, cus, d_out, d_cyc);
    for (int w : wps) run<1>("dependent v_fma_f32 chain", w, cus, d_out, d_cyc);
    for (int w : wps) run<2>("v_fma_f32 + salu 1:1", w, cus, d_out, d_cyc);
    for (int w : wps) run<3>("independent v_rcp_f32", w, cus, d_out, d_cyc);
    for (int w : wps) run<4>("v_cmp + v_cndmask pairs", w, cus, d_out, d_cyc);
    for (int w : wps) run<5>("independent v_pk_fma_f32", w, cus, d_out, d_cyc);
    for (int w : wps) run<6>("box-test mix (fma max3 min3 cmp cndmask sub mul add)", w, cus, d_out, d_cyc);
    return 0;
}
