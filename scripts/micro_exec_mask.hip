// Micro-benchmark (GPU box): does a wave64 VALU instruction cost less when whole 16-lane quarters of EXEC are empty?
// A dependent chain of v_fma_f32 under four lane masks: all 64 lanes, lanes 0..15, lanes 0..31, every 4th lane (16 lanes, one
// per quad of every quarter).  One workgroup of 256 threads per CU x 5 (the traversal kernel's occupancy); time per launch.
// build: hipcc --offload-arch=gfx950 -O3 [-DCHAINS=1] scripts/micro_exec_mask.hip -o scratch_so/micro_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef CHAINS
#define CHAINS 4
#endif
__global__ __launch_bounds__(256) void chain(float *out, unsigned long long mask, int iters) {
    const unsigned lane = threadIdx.x & 63u;
    float a = (float) threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
    float a1 = a + 1.f, a2 = a + 2.f, a3 = a + 3.f; // CHAINS independent accumulators: 1 = latency-bound per wave, 4 = issue-bound
    if ((mask >> lane) & 1ull) {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64 / CHAINS; ++k) {
                a = __builtin_fmaf(a, b, c);
                if (CHAINS > 1) { a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + a1 + a2 + a3;
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 5 * 256 * sizeof(float));
    const struct { const char *name; unsigned long long m; } cases[] = {
        { "all 64 lanes", ~0ull }, { "lanes 0..15", 0xffffull }, { "lanes 0..31", 0xffffffffull },
        { "lanes 0..15 and 32..47", 0x0000ffff0000ffffull }, { "every 4th lane (16 lanes)", 0x1111111111111111ull },
        { "lane 0 only", 1ull }, { "lanes 0, 16, 32, 48", 0x0001000100010001ull } };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &cs : cases) {
        hipLaunchKernelGGL(chain, dim3(256 * 5), dim3(256), 0, 0, out, cs.m, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain, dim3(256 * 5), dim3(256), 0, 0, out, cs.m, 2000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 5 waves per SIMD x 2000 x 64 dependent FMAs each
        printf("chains %d  %-28s %8.3f ms  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", CHAINS, cs.name, ms, ms * 1e-3 * 2.4e9 / (5.0 * 2000 * 64));
    }
    return 0;
}
