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
// Launch plan (P + 2 kernels; P = 4 for a human sample: 3 type + 5 contig + 28 position bits):
//   the caller (or one reduction) tells which key bits can be set at all; those bits are
//   squeezed into a dense key of L bits (up to four bit fields), sorted LSD with P = ceil(L / 9)
//   digits of ceil(L / P) bits:
//   k_pair_init      dense keys, idx[i] = i, per-workgroup histogram of digit 0 (one workgroup = one
//                    block of 2048 keys up to 128 k keys, 4096 beyond), zero the histograms of the later
//                    digits;
//   k_radix_pass × P one workgroup per block, its keys in registers.  Every workgroup sums the histogram
//                    rows of ALL blocks itself (two buckets per thread, 16 rows in flight) — no scan
//                    kernel, no look-back; the table is 2 KiB per block and L2-resident —, its four waves
//                    count their quarter's digits in LDS, rank the keys stably with __ballot match masks
//                    and scatter them, counting on the way the NEXT digit into the histogram row of the
//                    block each key lands in;
//   k_partition      boundary flags straight from the sorted dense keys (they expand back to the original
//                    keys: no gather) + inclusive scan → partition ids, perm, n_parts.  One workgroup up
//                    to 16 k candidates; beyond that at most half a workgroup per CU with one arrival
//                    counter (every workgroup of the grid is resident).
// Integer work, 20 B per candidate algorithmic, no MFMA; at the product's 60 k candidates it is bound by
// the six dependent launches (≈ 9 µs per pass even with no output), not by bytes.  Determinism: ranks
// come from prefix sums only; the histogram atomics are commutative counts.
#include "svx_internal.h"

namespace {

constexpr int kMaxDigitBits = 9;
constexpr int kBuckets = 1 << kMaxDigitBits;
constexpr uint32_t kSingleBlockMax = 16384;  // one workgroup sweeps up to two chunks itself
constexpr uint32_t kSmallSortMax = 131072;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct KeyFields {  // dense = OR over f of ((key >> shift[f]) & mask[f]) << off[f]
    uint32_t n;
    uint32_t shift[4], off[4];
    uint64_t mask[4];
};

struct PairBufs {
    uint64_t* keys[2];  // dense keys, ping-pong
    uint32_t* idx[2];
    uint32_t* hist;     // [passes][n_blocks][kBuckets]
    uint32_t n;
    uint32_t n_blocks;
    uint32_t passes;
    uint32_t digit_bits;
    KeyFields f;
};

__device__ __forceinline__ uint64_t dense_key(const KeyFields& f, uint64_t k) {
    uint64_t d = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
        if (i < f.n) d |= ((k >> f.shift[i]) & f.mask[i]) << f.off[i];
    return d;
}

__global__ __launch_bounds__(256) void k_key_or(const uint64_t* __restrict__ keys, uint32_t n,
                                                unsigned long long* __restrict__ out) {
    uint64_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc |= keys[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        acc |= ((uint64_t)__shfl_xor((uint32_t)(acc >> 32), d) << 32) | __shfl_xor((uint32_t)acc, d);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(out, (unsigned long long)acc);
}

// One workgroup (4 waves) per block of 256 * ITERS keys; wave w owns the w-th quarter of the block
// (contiguous, so wave order == key order), 64 consecutive keys per iteration.
template <int ITERS>
__global__ __launch_bounds__(256) void k_pair_init(const uint64_t* __restrict__ keys, PairBufs b) {
    __shared__ uint32_t s_h[kBuckets];
    const int tid = threadIdx.x;
    const uint32_t dmask = (1u << b.digit_bits) - 1u;
    // zero the histograms of the later passes (counted into by the scatter of the pass before)
    {
        const size_t total = (size_t)(b.passes - 1) * b.n_blocks * kBuckets;
        uint32_t* later = b.hist + (size_t)b.n_blocks * kBuckets;
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) later[i] = 0;
    }
    for (int v = tid; v < kBuckets; v += 256) s_h[v] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256u * ITERS);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t i = base + it * 256 + tid;
        if (i < b.n) {
            const uint64_t d = dense_key(b.f, keys[i]);
            b.keys[0][i] = d;
            b.idx[0][i] = i;
            atomicAdd(&s_h[(uint32_t)d & dmask], 1u);
        }
    }
    __syncthreads();
    for (int v = tid; v < kBuckets; v += 256) b.hist[(size_t)blockIdx.x * kBuckets + v] = s_h[v];
}

template <int ITERS>
__global__ __launch_bounds__(256) void k_radix_pass(PairBufs b, uint32_t pass) {
    __shared__ uint32_t s_run[4][kBuckets];  // per (wave, digit): count, then running output position
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t shift = pass * b.digit_bits;
    const uint32_t dmask = (1u << b.digit_bits) - 1u;
    const int sb = pass & 1;
    const uint64_t* __restrict__ sk = b.keys[sb];
    const uint32_t* __restrict__ si = b.idx[sb];
    uint64_t* __restrict__ dk = b.keys[sb ^ 1];
    uint32_t* __restrict__ di = b.idx[sb ^ 1];
    const uint32_t* __restrict__ hist = b.hist + (size_t)pass * b.n_blocks * kBuckets;
    uint32_t* __restrict__ hnext = pass + 1 < b.passes ? b.hist + (size_t)(pass + 1) * b.n_blocks * kBuckets : nullptr;
    constexpr uint32_t kBlockKeys = 256u * ITERS;

    // ---- this wave's keys (registers) and its digit counts (LDS)
    for (int v = tid; v < 4 * kBuckets; v += 256) (&s_run[0][0])[v] = 0;
    uint64_t key[ITERS];
    uint32_t id[ITERS];
    const uint32_t base = blockIdx.x * kBlockKeys + wave * (64u * ITERS);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t i = base + it * 64 + lane;
        key[it] = 0;
        id[it] = 0;
        if (i < b.n) { key[it] = sk[i]; id[it] = si[i]; }
    }
    // ---- global bucket offsets of this block: column sums over the blocks' histograms (two buckets per
    // thread, one 8-byte load per row, 16 rows in flight)
    uint32_t before0 = 0, before1 = 0, total0 = 0, total1 = 0;
    {
        const uint2* __restrict__ col = reinterpret_cast<const uint2*>(hist) + tid;
        // (rows past the end read as zero through the predicate: the tail is a batch like the others,
        // never one dependent load after another)
        for (uint32_t c = 0; c < b.n_blocks; c += 16) {
            uint2 x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                x[u] = c + u < b.n_blocks ? col[(size_t)(c + u) * (kBuckets / 2)] : make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                total0 += x[u].x; total1 += x[u].y;
                if (c + u < blockIdx.x) { before0 += x[u].x; before1 += x[u].y; }
            }
        }
    }
    __syncthreads();  // s_run zeroed
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (base + it * 64 + lane < b.n) atomicAdd(&s_run[wave][(uint32_t)(key[it] >> shift) & dmask], 1u);
    // exclusive scan of the bucket totals over the workgroup (thread t holds buckets 2t, 2t+1)
    {
        uint32_t s = total0 + total1;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const uint32_t t = __shfl_up(s, k);
            if (lane >= k) s += t;
        }
        if (lane == 63) s_w[wave] = s;
        __syncthreads();  // also: every wave's counts are in s_run
        uint32_t wp = 0;
        for (int w = 0; w < wave; ++w) wp += s_w[w];
        const uint32_t ex = wp + s - (total0 + total1);
        // counts -> starting positions: global base of the bucket + this block's earlier waves
        uint32_t acc0 = ex + before0, acc1 = ex + total0 + before1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c0 = s_run[w][2 * tid], c1 = s_run[w][2 * tid + 1];
            s_run[w][2 * tid] = acc0;
            s_run[w][2 * tid + 1] = acc1;
            acc0 += c0;
            acc1 += c1;
        }
    }
    __syncthreads();

    // ---- one wave per quarter block: stable ranking with match masks, scatter, next digit's counts
    uint32_t* run = s_run[wave];
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t nshift = shift + b.digit_bits;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t i = base + it * 64 + lane;
        const bool valid = i < b.n;
        const uint32_t dig = (uint32_t)(key[it] >> shift) & dmask;
        uint64_t m = __ballot(valid);
        if (m == 0) break;  // wave-uniform: nothing left in this quarter
        for (uint32_t bit = 0; bit < b.digit_bits; ++bit) {
            const uint64_t bal = __ballot((dig >> bit) & 1u);
            m &= ((dig >> bit) & 1u) ? bal : ~bal;
        }
        const uint32_t rank = __popcll(m & lt);
        uint32_t pos = 0;
        if (valid) pos = run[dig] + rank;
        wave_lds_sync();
        if (valid && rank == 0) run[dig] += __popcll(m);
        wave_lds_sync();
        if (valid) {
#ifndef SVX_EXP_NOSCATTER  // (ablation builds only)
            dk[pos] = key[it];
            di[pos] = id[it];
#endif
#ifndef SVX_EXP_NOHNEXT  // (ablation builds only)
            if (hnext) atomicAdd(&hnext[(size_t)(pos / kBlockKeys) * kBuckets + ((uint32_t)(key[it] >> nshift) & dmask)], 1u);
#endif
        }
    }
}

struct PartArgs {
    const uint64_t* sorted_keys;  // dense keys in sorted order (after the last pass)
    const uint32_t* sorted;       // the permutation that goes with them
    KeyFields f;                  // dense -> original key
    uint32_t n;
    uint32_t max_dist;
    uint32_t* perm;
    uint32_t* part_id;
    uint32_t* n_parts;
    uint32_t* block_tot;      // [gridDim.x] flags per workgroup (multi-block form)
    uint32_t* counter;        // self-cleaning arrival counter (workspace header)
};

// every set bit of an original key lies inside the fields (the caller's key_bits cover all keys), so the
// dense key expands back to it exactly: the sweep reads the sorted keys contiguously, no gather
__device__ __forceinline__ uint64_t original_key(const KeyFields& f, uint64_t d) {
    uint64_t k = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
        if (i < f.n) k |= ((d >> f.off[i]) & f.mask[i]) << f.shift[i];
    return k;
}

// inclusive scan of `v` over a 1024-thread workgroup; returns the scanned value, *total the sum
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = v;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const uint32_t t = __shfl_up(s, k);
        if (lane >= k) s += t;
    }
    if (lane == 63) s_w[wave] = s;
    __syncthreads();
    uint32_t wp = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wp += s_w[w];
        tot += s_w[w];
    }
    __syncthreads();
    *total = tot;
    return wp + s;
}

constexpr int kPartPer = 8;                       // consecutive sorted keys per thread and chunk
constexpr uint32_t kPartChunk = 1024u * kPartPer; // keys one workgroup sweeps per scan

// Boundary flags of the thread's kPartPer consecutive positions of chunk [base, base + kPartChunk) ∩ [.., hi):
// bit i of the result = position base + tid * kPartPer + i opens a partition (SVIM_COMBINE.py:24-26).
__device__ __forceinline__ uint32_t chunk_flags(const PartArgs& p, uint32_t base, uint32_t hi) {
    const uint32_t j0 = base + threadIdx.x * kPartPer;
    if (j0 >= hi) return 0;
    uint64_t prev = j0 ? original_key(p.f, p.sorted_keys[j0 - 1]) : 0;
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < kPartPer; ++i) {
        const uint32_t j = j0 + i;
        if (j < hi) {
            const uint64_t c = original_key(p.f, p.sorted_keys[j]);
            const uint32_t pa = (uint32_t)prev, pc = (uint32_t)c;
            const uint32_t dist = pa > pc ? pa - pc : pc - pa;
            if (j && ((prev >> 32) != (c >> 32) || dist > p.max_dist)) bits |= 1u << i;
            prev = c;
        }
    }
    return bits;
}

// Workgroup g owns the contiguous range [g * span, (g + 1) * span) of the sorted keys, swept in chunks of
// 8192 (eight consecutive keys per thread: one scan of the thread counts per chunk).  With several
// workgroups, each first counts its flags, publishes the count and meets the others at one arrival
// counter (every workgroup is resident: the grid never exceeds half a workgroup per CU); its carry-in
// is the sum of the earlier workgroups' counts.  A range of one chunk keeps its flags in registers.
__global__ __launch_bounds__(1024) void k_partition(PartArgs p, uint32_t span) {
    __shared__ uint32_t s_w[16];
    const uint32_t lo = blockIdx.x * span, hi = min(p.n, lo + span);
    const bool one_chunk = hi - lo <= kPartChunk;
    uint32_t carry = 0, kept = 0;
    if (gridDim.x > 1) {
        uint32_t cnt = 0;
        for (uint32_t base = lo; base < hi; base += kPartChunk) {
            kept = chunk_flags(p, base, hi);
            cnt += __popc(kept);
        }
        uint32_t tot;
        (void)block_scan_1024(cnt, s_w, &tot);
        if (threadIdx.x == 0) {
            // block_tot is the only data another workgroup reads inside this launch: agent-scope atomic
            // store and loads on both sides, no cache-wide release / acquire fences needed
            __hip_atomic_store(&p.block_tot[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
                __builtin_amdgcn_s_sleep(2);
            // the last workgroup to get here puts the counter back to zero for the next call
            if (__hip_atomic_fetch_add(p.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        // exclusive sum of the earlier workgroups' flags: one load per thread (fewer than 1024 workgroups)
        const uint32_t v = threadIdx.x < blockIdx.x
                               ? __hip_atomic_load(&p.block_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        (void)block_scan_1024(v, s_w, &carry);
    }
    for (uint32_t base = lo; base < hi; base += kPartChunk) {
        const uint32_t bits = (one_chunk && gridDim.x > 1) ? kept : chunk_flags(p, base, hi);
        uint32_t tot;
        uint32_t id = carry + block_scan_1024(__popc(bits), s_w, &tot) - __popc(bits);  // partitions opened before this thread
        const uint32_t j0 = base + threadIdx.x * kPartPer;
#pragma unroll
        for (int i = 0; i < kPartPer; ++i) {
            const uint32_t j = j0 + i;
            if (j < hi) {
                id += (bits >> i) & 1u;
                p.part_id[j] = id;
                p.perm[j] = p.sorted[j];
            }
        }
        carry += tot;
    }
    if (threadIdx.x == 0 && hi == p.n) *p.n_parts = p.n ? carry + 1 : 0;
}

// bit fields of `bits` (set bits = key bits that can be non-zero), at most four: nearby runs are merged
KeyFields fields_of(uint64_t bits, uint32_t* live) {
    struct Run { uint32_t lo, hi; };  // [lo, hi)
    std::vector<Run> runs;
    for (uint32_t i = 0; i < 64;) {
        if (!((bits >> i) & 1)) { ++i; continue; }
        uint32_t j = i;
        while (j < 64 && ((bits >> j) & 1)) ++j;
        runs.push_back({i, j});
        i = j;
    }
    while (runs.size() > 4) {  // merge the two runs separated by the smallest gap
        size_t best = 0;
        for (size_t r = 1; r + 1 < runs.size(); ++r)
            if (runs[r + 1].lo - runs[r].hi < runs[best + 1].lo - runs[best].hi) best = r;
        runs[best].hi = runs[best + 1].hi;
        runs.erase(runs.begin() + best + 1);
    }
    KeyFields f;
    memset(&f, 0, sizeof(f));
    uint32_t off = 0;
    for (const Run& r : runs) {
        const uint32_t w = r.hi - r.lo;
        f.shift[f.n] = r.lo;
        f.off[f.n] = off;
        f.mask[f.n] = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
        off += w;
        ++f.n;
    }
    *live = off;
    return f;
}

int pair_partition_bits(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n, uint32_t max_dist, uint64_t key_bits,
                        uint32_t* d_perm, uint32_t* d_part_id, uint32_t* d_n_parts) {
    PairBufs b;
    uint32_t live = 0;
    b.f = fields_of(key_bits, &live);
    if (live == 0) {  // every key is zero: one digit of one bit keeps the code path uniform
        live = 1;
        b.f.n = 1;
        b.f.shift[0] = 0; b.f.off[0] = 0; b.f.mask[0] = 1;
    }
    b.passes = (live + kMaxDigitBits - 1) / kMaxDigitBits;
    b.digit_bits = (live + b.passes - 1) / b.passes;
    b.n = n;
    // keys per workgroup: 2048 up to 128 k keys (more workgroups for a batch that cannot fill the chip
    // anyway), 4096 beyond (fewer histogram rows for every workgroup to sum)
    const bool small = n <= kSmallSortMax;
    const uint32_t block_keys = small ? 256u * 8 : 256u * 16;
    b.n_blocks = (n + block_keys - 1) / block_keys;
    // the multi-workgroup sweep has one arrival barrier: its grid stays at half a workgroup per CU (a CU
    // holds two of these 1024-thread workgroups), so that up to four such grids — other contexts or
    // processes on the same device — are resident together and none can starve another's barrier
    const uint32_t part_grid = n <= kSingleBlockMax ? 1u
        : std::min<uint32_t>(std::max<uint32_t>(1u, (uint32_t)ctx->n_cu / 2), (n + 4095) / 4096);
    const size_t hist_words = (size_t)b.passes * b.n_blocks * kBuckets;
    size_t need = 2 * svx_take_bytes(n, 8) + 2 * svx_take_bytes(n, 4) + svx_take_bytes(hist_words, 4) +
                  svx_take_bytes(part_grid, 4);
    int rc = svx_ws_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    b.keys[0] = svx_ws_take<uint64_t>(ctx, n);
    b.keys[1] = svx_ws_take<uint64_t>(ctx, n);
    b.idx[0] = svx_ws_take<uint32_t>(ctx, n);
    b.idx[1] = svx_ws_take<uint32_t>(ctx, n);
    b.hist = svx_ws_take<uint32_t>(ctx, hist_words);
    uint32_t* block_tot = svx_ws_take<uint32_t>(ctx, part_grid);
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    if (small) hipLaunchKernelGGL(k_pair_init<8>, dim3(b.n_blocks), dim3(256), 0, ctx->stream, d_keys, b);
    else hipLaunchKernelGGL(k_pair_init<16>, dim3(b.n_blocks), dim3(256), 0, ctx->stream, d_keys, b);
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    for (uint32_t pass = 0; pass < b.passes; ++pass) {
        if (small) hipLaunchKernelGGL(k_radix_pass<8>, dim3(b.n_blocks), dim3(256), 0, ctx->stream, b, pass);
        else hipLaunchKernelGGL(k_radix_pass<16>, dim3(b.n_blocks), dim3(256), 0, ctx->stream, b, pass);
    }
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    PartArgs pa;
    pa.sorted_keys = b.keys[b.passes & 1];
    pa.sorted = b.idx[b.passes & 1];
    pa.f = b.f;
    pa.n = n;
    pa.max_dist = max_dist;
    pa.perm = d_perm;
    pa.part_id = d_part_id;
    pa.n_parts = d_n_parts;
    pa.block_tot = block_tot;
    pa.counter = reinterpret_cast<uint32_t*>(ctx->ws) + 64;  // workspace header: zero between calls
    const uint32_t span = ((n + part_grid - 1) / part_grid + 1023) / 1024 * 1024;
    hipLaunchKernelGGL(k_partition, dim3((n + span - 1) / span), dim3(1024), 0, ctx->stream, pa, span);
    SVX_HIP(ctx, hipGetLastError());
    return svx_timing_end(ctx);
}

}  // namespace

extern "C" int svx_pair_partition_dev_bits(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n, uint32_t max_dist,
                                           uint64_t key_bits, uint32_t* d_perm, uint32_t* d_part_id,
                                           uint32_t* d_n_parts) {
    if (!ctx || !d_n_parts) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {
        SVX_HIP(ctx, hipMemsetAsync(d_n_parts, 0, 4, ctx->stream));
        return SVX_OK;
    }
    if (!d_keys || !d_perm || !d_part_id) return SVX_E_INVALID;
    return pair_partition_bits(ctx, d_keys, n, max_dist, key_bits, d_perm, d_part_id, d_n_parts);
}

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
    // which key bits are used is not known here: one reduction and an 8-byte read-back (synchronises)
    int rc = svx_ws_reserve(ctx, svx_take_bytes(1, 8));
    if (rc != SVX_OK) return rc;
    unsigned long long* d_or = reinterpret_cast<unsigned long long*>(svx_ws_take<uint64_t>(ctx, 1));
    SVX_HIP(ctx, hipMemsetAsync(d_or, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_key_or, dim3(std::min<uint32_t>((n + 255) / 256, (uint32_t)ctx->n_cu * 4u)), dim3(256), 0,
                       ctx->stream, d_keys, n, d_or);
    SVX_HIP(ctx, hipGetLastError());
    uint64_t bits = 0;
    SVX_HIP(ctx, hipMemcpyAsync(&bits, d_or, 8, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return pair_partition_bits(ctx, d_keys, n, max_dist, bits, d_perm, d_part_id, d_n_parts);
}

extern "C" int svx_pair_partition(svx_ctx* ctx, const uint64_t* keys, uint32_t n, uint32_t max_dist,
                                  uint32_t* perm, uint32_t* part_id, uint32_t* n_parts) {
    if (!ctx || !n_parts) return SVX_E_INVALID;
    *n_parts = 0;
    if (n == 0) return SVX_OK;
    if (!keys || !perm || !part_id) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t bits = 0;  // the keys are here: which bits are in use costs one pass over host memory
    for (uint32_t i = 0; i < n; ++i) bits |= keys[i];
    size_t need = svx_take_bytes(n, 8) + 2 * svx_take_bytes(n, 4) + svx_take_bytes(1, 4);
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    uint64_t* d_k = svx_stage_take<uint64_t>(ctx, n);
    uint32_t* d_p = svx_stage_take<uint32_t>(ctx, n);
    uint32_t* d_id = svx_stage_take<uint32_t>(ctx, n);
    uint32_t* d_np = svx_stage_take<uint32_t>(ctx, 1);
    SVX_HIP(ctx, hipMemcpyAsync(d_k, keys, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = pair_partition_bits(ctx, d_k, n, max_dist, bits, d_p, d_id, d_np);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(perm, d_p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(part_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(n_parts, d_np, 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
