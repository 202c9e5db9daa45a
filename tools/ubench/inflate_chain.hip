// Floor of a wave-per-member DEFLATE decoder on gfx950 (round 5, verdict item 2, route ii): the SERIAL chain of one
// member — bit buffer -> LDS table look-up -> shift -> next look-up — run by ONE lane of a wave, the other 63 lanes
// joining for the window copy of a match, as such a design would.  Nothing is decoded: the table is filled so that
// code lengths and the literal / match mix are those of a BGZF member of an assembly BAM's SEQ bytes (~16 k literals of
// 4-5 bits, ~15 k matches of 3-4 bytes per 64 KiB member); what is measured is the time per symbol of the dependent
// chain with W waves resident per SIMD.  A member's window (32 KiB) + tables (4 KiB) in LDS allow 4 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/inflate_chain tools/ubench/inflate_chain.hip && /tmp/inflate_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kWindow = 32768;

// one wave per workgroup; LDS: 1024-entry literal/length table + 256-entry distance table + 32 KiB window
__global__ __launch_bounds__(64) void k_chain(const uint32_t* __restrict__ in, uint32_t n_words, uint32_t n_syms, uint32_t* out,
                                              uint64_t* cycles) {
    __shared__ uint16_t tab_ll[1024];
    __shared__ uint16_t tab_d[256];
    __shared__ uint8_t window[kWindow];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) {
        // half of the entries literals (4- and 5-bit codes), half matches (7-bit codes): entry = symbol << 4 | code length
        const uint32_t lit = (i * 2654435761u >> 20) & 1u;
        tab_ll[i] = lit ? (uint16_t)(((i & 0xFF) << 4) | (4 + (i & 1))) : (uint16_t)(((257 + (i & 3)) << 4) | 7);
    }
    for (int i = lane; i < 256; i += 64) tab_d[i] = (uint16_t)(((4 + (i & 15)) << 4) | 5);
    for (int i = lane; i < kWindow; i += 64) window[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t* src = in + (size_t)blockIdx.x * n_words;
    uint64_t buf = 0;
    uint32_t cnt = 0, pos_in = 0, produced = 0, acc = 0;
    uint32_t next = src[0];
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t s = 0; s < n_syms; ++s) {
        uint32_t len = 0, dist = 0;
        if (lane == 0) {  // the serial part: one lane
            if (cnt <= 32) {
                buf |= (uint64_t)next << cnt;
                cnt += 32;
                pos_in = pos_in + 1 < n_words ? pos_in + 1 : 0;
                next = src[pos_in];  // requested one refill ahead
            }
            const uint32_t e = tab_ll[(uint32_t)buf & 1023u];
            const uint32_t n = e & 15u;
            buf >>= n;
            cnt -= n;
            const uint32_t sym = e >> 4;
            if (sym < 256) {
                window[produced & (kWindow - 1)] = (uint8_t)sym;
                ++produced;
            } else {
                len = 3 + (sym - 257);
                const uint32_t d = tab_d[(uint32_t)buf & 255u];
                const uint32_t dn = d & 15u, ex = ((d >> 4) - 2u) >> 1;
                buf >>= dn;
                cnt -= dn;
                dist = 1u + ((2u + ((d >> 4) & 1u)) << ex) + ((uint32_t)buf & ((1u << ex) - 1u));
                buf >>= ex;
                cnt -= ex;
            }
        }
        // a match: every lane learns (len, dist) and copies its byte of the window
        len = __builtin_amdgcn_readfirstlane(len);
        if (len) {
            dist = __builtin_amdgcn_readfirstlane(dist);
            const uint32_t p = __builtin_amdgcn_readfirstlane(produced);
            if ((uint32_t)lane < len) {
                const uint32_t from = (p - dist + ((uint32_t)lane % dist)) & (kWindow - 1);
                window[(p + lane) & (kWindow - 1)] = window[from];
            }
            if (lane == 0) produced += len;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    acc = window[lane] + produced + (uint32_t)buf;
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const uint32_t n_words = 4096, n_syms = 31000;
    const int max_waves = n_cu * 8;
    std::vector<uint32_t> h((size_t)max_waves * n_words);
    uint32_t x = 12345;
    for (auto& w : h) { x = x * 1664525u + 1013904223u; w = x; }
    uint32_t *d_in, *d_out;
    uint64_t* d_cyc;
    (void)hipMalloc(&d_in, h.size() * 4);
    (void)hipMalloc(&d_out, (size_t)max_waves * 64 * 4);
    (void)hipMalloc(&d_cyc, (size_t)max_waves * 8);
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("device: %s, %d CUs, shader clock %.2f GHz\n", prop.name, n_cu, prop.clockRate / 1e6);
    // LDS per workgroup is 34.5 KiB, so at most 4 workgroups (= waves) fit a CU whatever the grid: 1, 2, 4 per CU
    for (int per_cu : {1, 2, 4}) {
        const int waves = n_cu * per_cu;
        hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, d_in, n_words, n_syms, d_out, d_cyc);  // warm-up
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, d_in, n_words, n_syms, d_out, d_cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%d wave(s) per CU (%5d members in flight): %8.3f ms per member-equivalent of %u symbols = %6.1f ns per symbol; "
               "%9.0f members/s for the chip\n", per_cu, waves, ms, n_syms, ms * 1e6 / n_syms, waves / (ms * 1e-3));
    }
    return 0;
}
