// svx_pair.hip — pair sort + partition on gfx950.
//
// Replaces form_partitions (reference SVIM_COMBINE.py:15-32): a STABLE sort of the
// candidates by Candidate.get_key() followed by a sweep that opens a new partition when
// type/contig differ or the key positions are more than max_distance apart.
//
// Keys are packed by the host as  group << 32 | pos  (group = (type, rank of the contig
// name under Python str order), pos = non-negative key position), so unsigned 64-bit
// order == the reference's tuple order and ties keep input order (hap-1 list, then hap-2).
//
// Kernels (8 B key + 4 B index per candidate in HBM, ping-pong buffers):
//   k_pair_init      copy keys, idx[i] = i, OR-reduce the keys (which 8-bit digits are live)
//   per live digit:  k_radix_hist → k_radix_scan → k_radix_scatter  (LSD, 8 bits per pass;
//                    one wave per 1024-key chunk; stable in-wave ranking with __ballot match
//                    masks; dead digits exit immediately)
//   k_partition      boundary flags + inclusive scan → partition ids, perm, n_parts
// Integer/HBM-bound work, no MFMA.  Determinism: ranks come from prefix sums only.
#include "svx_internal.h"

namespace {

constexpr int kChunk = 1024;
constexpr int kIters = kChunk / 64;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct PairBufs {
    uint64_t* keys[2];
    uint32_t* idx[2];
    uint32_t* hist;     // [256 * n_chunks], digit-major
    uint64_t* or_bits;  // OR of all keys
    uint32_t n;
    uint32_t n_chunks;
};

__device__ __forceinline__ bool pass_live(uint64_t orb, int d) { return ((orb >> (8 * d)) & 0xFFu) != 0; }
// buffer holding the data BEFORE pass d (= number of live passes below d, mod 2)
__device__ __forceinline__ int pass_src(uint64_t orb, int d) {
    int c = 0;
    for (int i = 0; i < d; ++i) c += pass_live(orb, i) ? 1 : 0;
    return c & 1;
}

__global__ __launch_bounds__(256) void k_pair_init(const uint64_t* __restrict__ keys, PairBufs b) {
    uint64_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        b.keys[0][i] = k;
        b.idx[0][i] = i;
        acc |= k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        acc |= ((uint64_t)__shfl_xor((uint32_t)(acc >> 32), d) << 32) | __shfl_xor((uint32_t)acc, d);
    }
    if ((threadIdx.x & 63) == 0 && acc) atomicOr((unsigned long long*)b.or_bits, (unsigned long long)acc);
}

__global__ __launch_bounds__(256) void k_radix_hist(PairBufs b, int d) {
    __shared__ uint32_t s_h[4][256];
    const uint64_t orb = *b.or_bits;
    if (!pass_live(orb, d)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t* src = b.keys[pass_src(orb, d)];
    uint32_t* h = s_h[wave];
    for (uint32_t chunk = blockIdx.x * 4 + wave; chunk < b.n_chunks; chunk += gridDim.x * 4) {
        for (int i = lane; i < 256; i += 64) h[i] = 0;
        wave_lds_sync();
        const uint32_t base = chunk * kChunk;
        for (int it = 0; it < kIters; ++it) {
            const uint32_t i = base + it * 64 + lane;
            if (i < b.n) atomicAdd(&h[(uint32_t)(src[i] >> (8 * d)) & 0xFFu], 1u);
        }
        wave_lds_sync();
        for (int i = lane; i < 256; i += 64) b.hist[(size_t)i * b.n_chunks + chunk] = h[i];
        wave_lds_sync();
    }
}

// exclusive scan of hist[256 * n_chunks] in place (single workgroup, chunked with carry)
__global__ __launch_bounds__(1024) void k_radix_scan(PairBufs b, int d) {
    __shared__ uint32_t s_w[16];
    const uint64_t orb = *b.or_bits;
    if (!pass_live(orb, d)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t total = (size_t)256 * b.n_chunks;
    uint32_t carry = 0;
    for (size_t base = 0; base < total; base += 1024) {
        const size_t i = base + tid;
        const uint32_t v = i < total ? b.hist[i] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            uint32_t t = __shfl_up(s, k);
            if (lane >= k) s += t;
        }
        if (lane == 63) s_w[wave] = s;
        __syncthreads();
        uint32_t wp = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) wp += s_w[w];
            tot += s_w[w];
        }
        if (i < total) b.hist[i] = carry + wp + s - v;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_radix_scatter(PairBufs b, int d) {
    __shared__ uint32_t s_run[4][256];
    const uint64_t orb = *b.or_bits;
    if (!pass_live(orb, d)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sb = pass_src(orb, d);
    const uint64_t* __restrict__ sk = b.keys[sb];
    const uint32_t* __restrict__ si = b.idx[sb];
    uint64_t* __restrict__ dk = b.keys[sb ^ 1];
    uint32_t* __restrict__ di = b.idx[sb ^ 1];
    uint32_t* run = s_run[wave];
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t chunk = blockIdx.x * 4 + wave; chunk < b.n_chunks; chunk += gridDim.x * 4) {
        for (int i = lane; i < 256; i += 64) run[i] = b.hist[(size_t)i * b.n_chunks + chunk];
        wave_lds_sync();
        const uint32_t base = chunk * kChunk;
        for (int it = 0; it < kIters; ++it) {
            const uint32_t i = base + it * 64 + lane;
            const bool valid = i < b.n;
            uint64_t key = 0;
            uint32_t id = 0;
            if (valid) { key = sk[i]; id = si[i]; }
            const uint32_t dig = (uint32_t)(key >> (8 * d)) & 0xFFu;
            // lanes holding the same digit (match-any via 8 ballots)
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const uint64_t bal = __ballot((dig >> bit) & 1u);
                m &= ((dig >> bit) & 1u) ? bal : ~bal;
            }
            const uint32_t rank = __popcll(m & lt);
            uint32_t pos = 0;
            if (valid) pos = run[dig] + rank;
            wave_lds_sync();
            if (valid && rank == 0) run[dig] += __popcll(m);
            wave_lds_sync();
            if (valid) { dk[pos] = key; di[pos] = id; }
        }
        wave_lds_sync();
    }
}

// boundary flags + inclusive scan (single workgroup, chunked with carry)
__global__ __launch_bounds__(1024) void k_partition(PairBufs b, uint32_t max_dist,
                                                    uint32_t* __restrict__ perm,
                                                    uint32_t* __restrict__ part_id,
                                                    uint32_t* __restrict__ n_parts) {
    __shared__ uint32_t s_w[16];
    const uint64_t orb = *b.or_bits;
    const int fb = pass_src(orb, 8);
    const uint64_t* __restrict__ sk = b.keys[fb];
    const uint32_t* __restrict__ si = b.idx[fb];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < b.n; base += 1024) {
        const uint32_t j = base + tid;
        uint32_t flag = 0;
        if (j < b.n && j > 0) {
            const uint64_t a = sk[j - 1], c = sk[j];
            const uint32_t pa = (uint32_t)a, pc = (uint32_t)c;
            const uint32_t dist = pa > pc ? pa - pc : pc - pa;
            flag = ((a >> 32) != (c >> 32) || dist > max_dist) ? 1u : 0u;  // SVIM_COMBINE.py:24-26
        }
        uint32_t s = flag;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            uint32_t t = __shfl_up(s, k);
            if (lane >= k) s += t;
        }
        if (lane == 63) s_w[wave] = s;
        __syncthreads();
        uint32_t wp = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) wp += s_w[w];
            tot += s_w[w];
        }
        if (j < b.n) {
            part_id[j] = carry + wp + s;
            perm[j] = si[j];
        }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *n_parts = b.n ? carry + 1 : 0;
}

}  // namespace

extern "C" int svx_pair_partition_dev(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n,
                                      uint32_t max_dist, uint32_t* d_perm, uint32_t* d_part_id,
                                      uint32_t* d_n_parts) {
    if (!ctx || !d_n_parts) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {
        SVX_HIP(ctx, hipMemsetAsync(d_n_parts, 0, 4, ctx->stream));
        return SVX_OK;
    }
    if (!d_keys || !d_perm || !d_part_id) return SVX_E_INVALID;
    PairBufs b;
    b.n = n;
    b.n_chunks = (n + kChunk - 1) / kChunk;
    size_t need = 2 * svx_take_bytes(n, 8) + 2 * svx_take_bytes(n, 4) +
                  svx_take_bytes((size_t)256 * b.n_chunks, 4) + svx_take_bytes(1, 8);
    int rc = svx_ws_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    b.keys[0] = svx_ws_take<uint64_t>(ctx, n);
    b.keys[1] = svx_ws_take<uint64_t>(ctx, n);
    b.idx[0] = svx_ws_take<uint32_t>(ctx, n);
    b.idx[1] = svx_ws_take<uint32_t>(ctx, n);
    b.hist = svx_ws_take<uint32_t>(ctx, (size_t)256 * b.n_chunks);
    b.or_bits = svx_ws_take<uint64_t>(ctx, 1);
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemsetAsync(b.or_bits, 0, 8, ctx->stream));
    const uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    uint32_t g_init = (n + 255) / 256;
    if (g_init > cap) g_init = cap;
    uint32_t g_chunk = (b.n_chunks + 3) / 4;
    if (g_chunk > cap) g_chunk = cap;
    hipLaunchKernelGGL(k_pair_init, dim3(g_init), dim3(256), 0, ctx->stream, d_keys, b);
    svx_timing_mark(ctx, 1);
    for (int d = 0; d < 8; ++d) {
        hipLaunchKernelGGL(k_radix_hist, dim3(g_chunk), dim3(256), 0, ctx->stream, b, d);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, ctx->stream, b, d);
        hipLaunchKernelGGL(k_radix_scatter, dim3(g_chunk), dim3(256), 0, ctx->stream, b, d);
    }
    svx_timing_mark(ctx, 2);
    hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, ctx->stream, b, max_dist, d_perm,
                       d_part_id, d_n_parts);
    SVX_HIP(ctx, hipGetLastError());
    return svx_timing_end(ctx);
}

extern "C" int svx_pair_partition(svx_ctx* ctx, const uint64_t* keys, uint32_t n, uint32_t max_dist,
                                  uint32_t* perm, uint32_t* part_id, uint32_t* n_parts) {
    if (!ctx || !n_parts) return SVX_E_INVALID;
    *n_parts = 0;
    if (n == 0) return SVX_OK;
    if (!keys || !perm || !part_id) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n, 8) + 2 * svx_take_bytes(n, 4) + svx_take_bytes(1, 4);
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    uint64_t* d_k = svx_stage_take<uint64_t>(ctx, n);
    uint32_t* d_p = svx_stage_take<uint32_t>(ctx, n);
    uint32_t* d_id = svx_stage_take<uint32_t>(ctx, n);
    uint32_t* d_np = svx_stage_take<uint32_t>(ctx, 1);
    SVX_HIP(ctx, hipMemcpyAsync(d_k, keys, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = svx_pair_partition_dev(ctx, d_k, n, max_dist, d_p, d_id, d_np);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(perm, d_p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(part_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(n_parts, d_np, 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
