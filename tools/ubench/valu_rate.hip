// Micro-benchmark: issue rate of plain 32-bit integer VALU ops (v_add_u32 / v_and_b32 / v_bfe_i32 /
// v_cndmask) on gfx950, per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
            if (KIND == 1) { a0 &= a1 | 1; a1 ^= a2; a2 &= a3 | 3; a3 ^= a4; a4 &= a5 | 7; a5 ^= a6; a6 &= a7 | 15; a7 ^= a0; }
            if (KIND == 2) {
                a0 = __builtin_amdgcn_sbfe(a1, a2 & 31, 1); a1 = __builtin_amdgcn_sbfe(a2, a3 & 31, 1);
                a2 = __builtin_amdgcn_sbfe(a3, a4 & 31, 1) + a2; a3 = __builtin_amdgcn_sbfe(a4, a5 & 31, 1) + a3;
                a4 += a0; a5 += a1; a6 ^= a2; a7 ^= a3;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int kind = 0; kind < 3; ++kind)
        for (int blocks : {1024, 2048}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // nominal instruction count per wave: kind0: 8 adds x8 ; kind1: 12 ; kind2: ~16
                double waves = blocks * 4.0;
                double insts = waves * iters * 8.0 * (kind == 0 ? 8 : kind == 1 ? 12 : 16);
                if (rep) printf("kind %d blocks %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz, 1024 SIMDs)\n", kind, blocks, ms,
                                ms * 1e-3 * 2.4e9 * 1024.0 / insts);
            }
        }
    return 0;
}
