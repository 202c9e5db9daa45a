// Micro-benchmark: per-instruction issue cost of the integer VALU ops the CIGAR walk uses (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int i = 0; i < iters; ++i) { \
        REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "s"(sc));) \
    }

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters, uint32_t sc) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 & 15, b1 = a0 | 1;
    if (KIND == 0) BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5")
    if (KIND == 1) BODY("v_bfe_i32 %0, %6, %4, 1\n v_bfe_i32 %1, %6, %5, 1\n v_bfe_i32 %2, %6, %4, 1\n v_bfe_i32 %3, %6, %5, 1")
    if (KIND == 2) BODY("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %5")
    if (KIND == 3) BODY("v_lshrrev_b32 %0, 4, %0\n v_lshrrev_b32 %1, 4, %1\n v_lshrrev_b32 %2, 4, %2\n v_lshrrev_b32 %3, 4, %3")
    if (KIND == 4) BODY("v_cmp_gt_u32 vcc, %4, %0\n v_cmp_gt_u32 vcc, %5, %1\n v_cmp_gt_u32 vcc, %4, %2\n v_cmp_gt_u32 vcc, %5, %3")
    if (KIND == 5) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc")
    if (KIND == 6) BODY("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %5, %4")
    if (KIND == 7) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (KIND == 8) BODY("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %4\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %5, %4")
    if (KIND == 9) BODY("v_bfe_u32 %0, %0, 4, 28\n v_bfe_u32 %1, %1, 4, 28\n v_bfe_u32 %2, %2, 4, 28\n v_bfe_u32 %3, %3, 4, 28")
    if (KIND == 10) BODY("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %5")
    if (KIND == 11) BODY("v_cmp_gt_u32 s[20:21], %4, %0\n v_cmp_gt_u32 s[22:23], %5, %1\n v_cmp_gt_u32 s[24:25], %4, %2\n v_cmp_gt_u32 s[26:27], %5, %3")
    if (KIND == 12) BODY("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %5, %3")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

const char* names[] = {"v_add_u32", "v_bfe_i32", "v_and_b32", "v_lshrrev_b32", "v_cmp_gt_u32 vcc", "v_cndmask_b32", "v_add3_u32",
                       "v_mov_b32_dpp", "v_and_or_b32", "v_bfe_u32 imm", "v_add_co_u32", "v_cmp_gt_u32 sgpr", "v_bcnt_u32_b32"};
#define LAUNCH(K) case K: hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters, 0x185u); break;
int main() {
    uint32_t* d; (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, blocks = 2048;  // 8 waves per SIMD
    for (int kind = 0; kind < 13; ++kind) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            switch (kind) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double insts = blocks * 4.0 * iters * 8.0 * 4.0;
        printf("%-20s %.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", names[kind], best, best * 1e-3 * 2.4e9 * 1024.0 / insts);
    }
    return 0;
}
