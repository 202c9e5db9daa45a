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
// Two plans, same results (tests run every case on both):
//
// * up to 131072 candidates (one diploid sample is 60-90 k): ONE launch, k_pair_single — slices of the
//   input are turned into dense keys and counted per bucket of the leading 9 key bits, the bucket sequence
//   is cut into windows of about n / grid keys, every workgroup gathers its window into LDS, sorts it there
//   with stable 9-bit counting passes, flags the partition borders and, after the second of two arrival
//   barriers, adds the partitions opened in earlier windows (see the comment at the kernel);
//
// * beyond that: P + 2 launches (P = 4 for a human sample: 3 type + 5 contig + 28 position bits):
//   the caller (or one reduction) tells which key bits can be set at all; those bits are
//   squeezed into a dense key of L bits (up to four bit fields), sorted LSD with P = ceil(L / 9)
//   digits of ceil(L / P) bits:
//   k_pair_init      dense keys, idx[i] = i, per-workgroup histogram of digit 0 (one workgroup = one
//                    block of 4096 keys), zero the histograms of the later digits;
//   k_radix_pass × P one workgroup per block, its keys in registers.  Every workgroup sums the histogram
//                    rows of ALL blocks itself (two buckets per thread, 16 rows in flight) — no scan
//                    kernel, no look-back; the table is 2 KiB per block and L2-resident —, its four waves
//                    count their quarter's digits in LDS, rank the keys stably with __ballot match masks
//                    and scatter them, counting on the way the NEXT digit into the histogram row of the
//                    block each key lands in;
//   k_partition      boundary flags straight from the sorted dense keys (they expand back to the original
//                    keys: no gather) + inclusive scan → partition ids, perm, n_parts.  At most half a
//                    workgroup per CU with one arrival counter (every workgroup of the grid is resident).
// Integer work, 20 B per candidate algorithmic, no MFMA; at the product's 60 k candidates it is bound by
// dependent latencies (launches, grid barriers, LDS passes), not by bytes.  Determinism: ranks
// come from prefix sums only; the histogram atomics are commutative counts.
#include "svx_internal.h"

namespace {

constexpr int kBarrierNoteWord = SVX_WS_BARRIER_NOTE_WORD;
constexpr int kMaxDigitBits = 9;
constexpr int kBuckets = 1 << kMaxDigitBits;
constexpr uint32_t kSingleBlockMax = 16384;  // one workgroup sweeps up to two chunks itself
constexpr uint32_t kSmallSortMax = 131072;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Waiting for the other workgroups of a launch is bounded: after kBarrierTicks of the 100 MHz clock (20 s —
// every workgroup of these grids is resident unless more than four contexts run them on one device) the
// workgroup notes it in the workspace header and leaves the kernel (what the others would have published is
// not there: nothing may be computed from it); the host finds the note at its next synchronisation
// (svx_barrier_check) and reports the call as failed instead of the device hanging.
#ifndef SVX_EXP_BARRIER_TICKS
#define SVX_EXP_BARRIER_TICKS 2000000000ull
#endif
__device__ __forceinline__ bool spin_until(const uint32_t* c, uint32_t want, uint32_t* timed_out) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > SVX_EXP_BARRIER_TICKS) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
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

// DPP lane movement (gfx9 family): 0 flows into lanes without a source
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive sum over the wave: shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 / lane 31 broadcasts
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += dpp0<0x111, 0xF>(v);
    v += dpp0<0x112, 0xF>(v);
    v += dpp0<0x114, 0xF>(v);
    v += dpp0<0x118, 0xF>(v);
    v += dpp0<0x142, 0xA>(v);
    v += dpp0<0x143, 0xC>(v);
    return v;
}

// inclusive scan of `v` over a 1024-thread workgroup; returns the scanned value, *total the sum.
// TAIL_SYNC = false: the caller has a barrier of its own before s_w is written again.
template <bool TAIL_SYNC = true>
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t s = wave_scan_incl(v);
    if (lane == 63) s_w[wave] = s;
    __syncthreads();
    uint32_t wp = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t x = s_w[w];
        if (w < wave) wp += x;
        tot += x;
    }
    if (TAIL_SYNC) __syncthreads();
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
// MODE 0: one launch, the workgroups meet at the arrival counter.  MODE 1 / MODE 2: the same sweep as two
// launches without any waiting between workgroups — 1 publishes the flag counts, 2 picks them up after the launch
// boundary: the plan svx_pair_partition falls back to when a wait of the one-launch forms runs out.
template <int MODE>
__global__ __launch_bounds__(1024) void k_partition(PartArgs p, uint32_t span) {
    __shared__ uint32_t s_w[16];
    const uint32_t lo = blockIdx.x * span, hi = min(p.n, lo + span);
    const bool one_chunk = MODE == 0 && hi - lo <= kPartChunk;
    uint32_t carry = 0, kept = 0;
    if (MODE == 1) {
        uint32_t cnt = 0;
        for (uint32_t base = lo; base < hi; base += kPartChunk) cnt += __popc(chunk_flags(p, base, hi));
        uint32_t tot;
        (void)block_scan_1024(cnt, s_w, &tot);
        if (threadIdx.x == 0) p.block_tot[blockIdx.x] = tot;
        return;
    }
    if (MODE == 2) {
        const uint32_t v = threadIdx.x < blockIdx.x ? p.block_tot[threadIdx.x] : 0u;
        (void)block_scan_1024(v, s_w, &carry);
    }
    if (MODE == 0 && gridDim.x > 1) {
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
            // (after a wait that ran out the carry-in is wrong, every store stays inside the outputs)
            (void)spin_until(p.counter, gridDim.x, p.counter + (kBarrierNoteWord - 64));
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

// ------------------------------------------------------------------------------------------------
// One launch for up to 128 k candidates (a diploid human sample has 60-90 k): bucket by the leading
// bits, then sort every window of buckets inside one workgroup's LDS.
//
//   slice pass   workgroup g turns slice g of the input into dense keys (written once) and counts the
//                slice's keys per fine bucket (the leading kFineBits of the dense key) into row g;
//   barrier A    (arrival counter; the grid is at most a quarter of the CUs, every workgroup resident)
//   windows      every workgroup sums the rows itself: bucket totals, their prefix, and from it the same
//                split of the bucket sequence into windows of about n / grid keys everywhere; workgroup g
//                owns window g: position `start` of the sorted order, M keys;
//   gather       the window's keys are collected in input order: a slice's share of the window and where
//                it goes follow from the rows, slices without a share are never read;
//   window sort  in LDS (M <= kWinCap): a window that arrives as at most kMergeRuns sorted runs — PAIR hands
//                over two lists ordered along the genome — is merged pairwise, one binary search per key
//                and round; otherwise stable LSD passes of 9-bit digits over the low bits and one over the
//                bucket number.  A window that does not fit (a single crowded bucket) runs the same
//                passes on a private stretch of HBM scratch: slower, same result, no second code path
//                for the caller;
//   flags        boundary flags between neighbours inside the window, local partition numbers;
//   barrier B    the windows publish their flag count and their first and last key; everybody derives the
//                flag between windows and the carry-in from that table;
//   output       perm / part_id at start .. start + M.
// Ranks come from prefix sums and match masks only: the result does not depend on scheduling.
constexpr uint32_t kFineBits = 9;
constexpr uint32_t kFine = 1u << kFineBits;
constexpr uint32_t kWinCap = 5120;          // keys of a window sorted in LDS: five per thread
constexpr uint32_t kSingleGridMax = 64;     // one wave tabulates the windows
constexpr int kGatherBatch = 8;            // groups of 128 keys in flight per wave
constexpr uint32_t kMergeRuns = 32;       // presorted windows: merged instead of counted (at most five rounds)
constexpr uint32_t kSingleDynLds = 2 * kWinCap * 8 + 2 * kWinCap * 4;  // beside 32 KiB of static counters

struct SingleArgs {
    const uint64_t* keys;
    uint64_t* dense;      // [n]
    uint32_t* rows;       // [grid][kFine]
    uint64_t* gkey;       // [2][n], windows that do not fit LDS
    uint32_t* gidx;       // [2][n]
    uint32_t* win;        // [grid][2]: flags inside the window, keys in the window
    uint64_t* win_keys;   // [grid][2]: first and last key of the sorted window
    KeyFields f;
    uint32_t n, live, max_dist, target, slice_len;
    uint32_t* perm;
    uint32_t* part_id;
    uint32_t* n_parts;
    uint32_t* counter;        // workspace header: this launch's two arrival counters
    uint32_t* counter_stale;  // the two of the launch before
    uint32_t* note;           // set when a wait ran out (svx_barrier_check)
};

#ifdef SVX_EXP_PAIRCLK  // (timeline builds only: 100 MHz clock stamps of thread 0 of every workgroup)
__device__ unsigned long long g_pair_clk[64 * 16];
#define PAIR_CLK(k) do { if (threadIdx.x == 0) g_pair_clk[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define PAIR_CLK(k)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void agent_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T agent_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads: the workgroup's earlier agent-scope stores are out, then one thread meets the other
// workgroups at counter c.  Nobody counts the leavers: the counters come in two sets used by alternate
// launches of a context, and the last workgroup to arrive at the FIRST barrier of a launch puts the set of
// the launch before (long finished: same stream) back to zero — one round trip less per barrier.
__device__ __forceinline__ bool grid_barrier(uint32_t* c, uint32_t* stale, uint32_t* note) {
    __shared__ uint32_t s_met;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stale && before == gridDim.x - 1) {
            agent_store(stale, 0u);
            agent_store(stale + 1, 0u);
        }
        s_met = (before == gridDim.x - 1 || spin_until(c, gridDim.x, note)) ? 1u : 0u;
    }
    __syncthreads();
    return s_met != 0;  // false: the wait ran out, the caller leaves the kernel
}

// hist[dig] += length of every run of equal digits among the wave's valid lanes (the valid lanes are a
// prefix of the wave): one LDS atomic per run instead of one per key — keys arrive nearly sorted
__device__ __forceinline__ void count_runs(uint32_t* hist, uint32_t dig, bool valid, int lane) {
    const uint32_t prev = __shfl_up(dig, 1);
    const bool head = lane == 0 || dig != prev;
    const uint64_t heads = __ballot(head);
    if (head && valid) {
        const uint64_t nxt = (heads >> lane) >> 1;
        atomicAdd(&hist[dig], nxt ? (uint32_t)__ffsll((unsigned long long)nxt) : 64u - (uint32_t)lane);
    }
}

__device__ __forceinline__ bool opens_partition(const KeyFields& f, uint64_t prev_dense, uint64_t cur_dense, uint32_t max_dist) {
    const uint64_t a = original_key(f, prev_dense), c = original_key(f, cur_dense);
    const uint32_t pa = (uint32_t)a, pc = (uint32_t)c;
    const uint32_t dist = pa > pc ? pa - pc : pc - pa;
    return (a >> 32) != (c >> 32) || dist > max_dist;
}

// One stable counting pass over the window: digit(key) = ((key >> shift) & mask) - sub, `bits` wide.
// Wave w owns the w-th sixteenth of the window, 64 consecutive keys per iteration.
__device__ __forceinline__ void window_pass(const uint64_t* sk, const uint32_t* si, uint64_t* dk, uint32_t* di, uint32_t M,
                                            uint32_t* s_cnt, uint32_t* s_w, uint32_t shift, uint32_t mask, uint32_t sub,
                                            uint32_t bits) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t per = (M + 1023) >> 10, wbase = wave * 64u * per;
    for (int v = tid; v < 16 * (int)kFine; v += 1024) s_cnt[v] = 0;
    __syncthreads();
    for (uint32_t it = 0; it < per; ++it) {
        const uint32_t j = wbase + it * 64 + lane;
        const bool valid = j < M;
        const uint32_t dig = valid ? ((uint32_t)(sk[j] >> shift) & mask) - sub : ~0u;
        if (__ballot(valid) == 0) break;
        count_runs(s_cnt + wave * kFine, dig, valid, lane);
    }
    __syncthreads();
    uint32_t c[16], tot = 0;
    if (tid < (int)kFine) {
#pragma unroll
        for (int w = 0; w < 16; ++w) { c[w] = s_cnt[w * kFine + tid]; tot += c[w]; }
    }
    uint32_t all;
    uint32_t base = block_scan_1024<false>(tot, s_w, &all) - tot;
    if (tid < (int)kFine) {
#pragma unroll
        for (int w = 0; w < 16; ++w) { s_cnt[w * kFine + tid] = base; base += c[w]; }
    }
    __syncthreads();
    uint32_t* run = s_cnt + wave * kFine;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t it = 0; it < per; ++it) {
        const uint32_t j = wbase + it * 64 + lane;
        const bool valid = j < M;
        uint64_t m = __ballot(valid);
        if (m == 0) break;
        uint64_t key = 0;
        uint32_t id = 0;
        if (valid) { key = sk[j]; id = si[j]; }
        const uint32_t dig = ((uint32_t)(key >> shift) & mask) - sub;
        for (uint32_t bit = 0; bit < bits; ++bit) {
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
    __syncthreads();
}

template <bool LDS>
__device__ __forceinline__ void window_work(const SingleArgs& a, char* s_dyn, uint32_t* s_cnt, uint32_t* s_w,
                                            const uint32_t* s_soff, const uint32_t* s_scnt, uint32_t lo, uint32_t hi,
                                            uint32_t start, uint32_t M, uint32_t low_bits) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // both halves of a ping-pong pair hang off one base pointer (LDS or HBM scratch): half h is base + h * stride
    uint64_t* kbase;
    uint32_t* ibase;
    uint32_t stride;
    if constexpr (LDS) {
        kbase = reinterpret_cast<uint64_t*>(s_dyn);
        ibase = reinterpret_cast<uint32_t*>(s_dyn + 2 * kWinCap * 8);
        stride = kWinCap;
    } else {
        kbase = a.gkey + start;
        ibase = a.gidx + start;
        stride = a.n;
    }
    auto K = [&](int h) { return kbase + (size_t)h * stride; };
    auto I = [&](int h) { return ibase + (size_t)h * stride; };
    // ---- gather, input order: slice s puts its share at s_soff[s].  One 16-byte sc1 load per lane and group
    // of 128 keys (keys 2 * lane and 2 * lane + 1 of the group), a batch of groups in flight before the first
    // is looked at; a slice is left as soon as its share has been found
    const uint64_t lt = (1ull << lane) - 1ull;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.dense, 0, (int)(a.n * 8u), 0x00020000);
        for (uint32_t s = wave; s < gridDim.x; s += 16) {
            const uint32_t share = s_scnt[s];
            if (!share) continue;
            uint32_t run = s_soff[s];
            const uint32_t end = run + share;
            const uint32_t sl = s * a.slice_len, sh = min(a.n, sl + a.slice_len);
            for (uint32_t base = sl; base < sh && run < end; base += 128 * kGatherBatch) {
                u32x4 d[kGatherBatch];
#pragma unroll
                for (int u = 0; u < kGatherBatch; ++u) {
                    const uint32_t i0 = base + u * 128 + 2 * lane;
                    d[u] = u32x4{0u, 0u, 0u, 0u};
                    if (i0 < sh) d[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 8u, 0, 16);
                }
#pragma unroll
                for (int u = 0; u < kGatherBatch; ++u) {
                    const uint32_t i0 = base + u * 128 + 2 * lane;
                    const uint64_t k0 = ((uint64_t)d[u].y << 32) | d[u].x, k1 = ((uint64_t)d[u].w << 32) | d[u].z;
                    const uint32_t f0 = (uint32_t)(k0 >> low_bits), f1 = (uint32_t)(k1 >> low_bits);
                    const bool in0 = i0 < sh && f0 >= lo && f0 < hi, in1 = i0 + 1 < sh && f1 >= lo && f1 < hi;
                    const uint64_t b0 = __ballot(in0), b1 = __ballot(in1);
                    const uint32_t below = run + __popcll(b0 & lt) + __popcll(b1 & lt);
                    if (in0) { K(0)[below] = k0; I(0)[below] = i0; }
                    if (in1) { K(0)[below + (in0 ? 1u : 0u)] = k1; I(0)[below + (in0 ? 1u : 0u)] = i0 + 1; }
                    run += __popcll(b0) + __popcll(b1);
                }
            }
        }
    }
    __syncthreads();
    PAIR_CLK(5);
    // ---- the window as it was gathered is a few sorted runs when the caller's lists were ordered (PAIR
    // hands over the haplotype-1 list, then the haplotype-2 list, each along the genome: two runs, a few more
    // where alignments overlap): up to kMergeRuns runs are merged pairwise, one binary search per key and
    // round; equal keys: the earlier run first, so input order is kept.  More runs: counting passes.
    int src = 0;
    bool sorted = false;
    if constexpr (LDS) {
        __shared__ uint32_t s_run[kMergeRuns + 2];
        const uint32_t per = (M + 1023) >> 10, j0 = tid * per;
        uint32_t bits = 0;
        if (j0 < M) {
            uint64_t prev = j0 ? K(0)[j0 - 1] : 0ull;
#pragma unroll
            for (uint32_t i = 0; i < kWinCap / 1024; ++i) {
                const uint32_t j = j0 + i;
                if (i < per && j < M) {
                    const uint64_t c = K(0)[j];
                    if (c < prev) bits |= 1u << i;  // (never at j = 0: prev = 0)
                    prev = c;
                }
            }
        }
        uint32_t descents;
        uint32_t r = block_scan_1024(__popc(bits), s_w, &descents) - __popc(bits);  // descents before this thread
        uint32_t runs = descents + 1;
        if (runs <= kMergeRuns) {
            sorted = true;
            if (tid == 0) { s_run[0] = 0; s_run[runs] = M; }
#pragma unroll
            for (uint32_t i = 0; i < kWinCap / 1024; ++i)
                if ((bits >> i) & 1u) s_run[++r] = j0 + i;
            __syncthreads();
            while (runs > 1) {
                const uint64_t* sk = K(src);
                const uint32_t* si = I(src);
                uint64_t* dk = K(src ^ 1);
                uint32_t* di = I(src ^ 1);
                for (uint32_t j = tid; j < M; j += 1024) {
                    const uint64_t x = sk[j];
                    uint32_t q = 0;  // the run of position j
                    for (uint32_t t = 1; t < runs; ++t) q += s_run[t] <= j ? 1u : 0u;
                    uint32_t out = j;
                    if ((q & 1u) == 0u && q + 1 < runs) {        // left run of a pair: keys of the right run below x
                        uint32_t lo = s_run[q + 1], hi = s_run[q + 2];
                        const uint32_t first = lo;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (sk[mid] < x) lo = mid + 1; else hi = mid;
                        }
                        out = j + (lo - first);
                    } else if (q & 1u) {                          // right run: keys of the left run up to x
                        uint32_t lo = s_run[q - 1], hi = s_run[q];
                        const uint32_t last = hi;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (sk[mid] <= x) lo = mid + 1; else hi = mid;
                        }
                        out = j - (last - lo);
                    }
                    dk[out] = x;
                    di[out] = si[j];
                }
                __syncthreads();
                const uint32_t merged = (runs + 1) >> 1;
                if (tid < 64) {  // run q of the next round starts where run 2q started
                    const uint32_t v = (uint32_t)tid <= merged ? ((uint32_t)tid == merged ? M : s_run[2 * tid]) : 0u;
                    wave_lds_sync();
                    if ((uint32_t)tid <= merged) s_run[tid] = v;
                }
                __syncthreads();
                runs = merged;
                src ^= 1;
            }
        }
    }
    // ---- stable LSD passes: the low bits in digits of at most 9, then the bucket number
    if (!sorted && low_bits) {
        const uint32_t passes = (low_bits + kFineBits - 1) / kFineBits;
        const uint32_t db = (low_bits + passes - 1) / passes;
        for (uint32_t p = 0; p < passes; ++p) {
            window_pass(K(src), I(src), K(src ^ 1), I(src ^ 1), M, s_cnt, s_w, p * db, (1u << db) - 1u, 0u, db);
            src ^= 1;
        }
    }
    if (!sorted && hi - lo > 1) {
        window_pass(K(src), I(src), K(src ^ 1), I(src ^ 1), M, s_cnt, s_w, low_bits, kFine - 1u, lo,
                    32u - (uint32_t)__clz((int)(hi - lo - 1)));
        src ^= 1;
    }
    PAIR_CLK(6);
    // ---- flags between neighbours, local partition numbers (into the free index buffer)
    uint32_t carry = 0;
    uint32_t* ID = I(src ^ 1);
    if constexpr (LDS) {  // at most five consecutive keys per thread: one scan for the whole window
        const uint32_t per = (M + 1023) >> 10, j0 = tid * per;
        uint32_t bits = 0;
        if (j0 < M) {
            uint64_t prev = j0 ? original_key(a.f, K(src)[j0 - 1]) : 0ull;
#pragma unroll
            for (uint32_t i = 0; i < kWinCap / 1024; ++i) {
                const uint32_t j = j0 + i;
                if (i < per && j < M) {
                    const uint64_t c = original_key(a.f, K(src)[j]);
                    const uint32_t pa = (uint32_t)prev, pc = (uint32_t)c;
                    const uint32_t dist = pa > pc ? pa - pc : pc - pa;
                    if (j && ((prev >> 32) != (c >> 32) || dist > a.max_dist)) bits |= 1u << i;
                    prev = c;
                }
            }
        }
        uint32_t id = block_scan_1024<false>(__popc(bits), s_w, &carry) - __popc(bits);
#pragma unroll
        for (uint32_t i = 0; i < kWinCap / 1024; ++i) {
            const uint32_t j = j0 + i;
            if (i < per && j < M) {
                id += (bits >> i) & 1u;
                ID[j] = id;
            }
        }
    } else {
        for (uint32_t base = 0; base < M; base += 1024) {
            const uint32_t j = base + tid;
            uint32_t flag = 0;
            if (j < M && j) flag = opens_partition(a.f, K(src)[j - 1], K(src)[j], a.max_dist) ? 1u : 0u;
            uint32_t tot;
            const uint32_t incl = block_scan_1024(flag, s_w, &tot);
            if (j < M) ID[j] = carry + incl;
            carry += tot;
        }
    }
    if (tid == 0) {
        agent_store(a.win + 2 * blockIdx.x, carry);
        agent_store(a.win + 2 * blockIdx.x + 1, M);
        agent_store(a.win_keys + 2 * blockIdx.x, K(src)[0]);
        agent_store(a.win_keys + 2 * blockIdx.x + 1, K(src)[M - 1]);
    }
    PAIR_CLK(7);
    if (!grid_barrier(a.counter + 1, nullptr, a.note)) return;
    PAIR_CLK(8);
    // ---- the table of windows: flag between windows, carry-in
    __shared__ uint32_t s_in[2];
    if (wave == 0) {
        const bool have = (uint32_t)lane < gridDim.x;
        const uint32_t cnt = have ? agent_load(a.win + 2 * lane) : 0u;
        const uint32_t m = have ? agent_load(a.win + 2 * lane + 1) : 0u;
        const uint64_t first = (have && m) ? agent_load(a.win_keys + 2 * lane) : 0ull;
        const uint64_t last = (have && m) ? agent_load(a.win_keys + 2 * lane + 1) : 0ull;
        const uint64_t filled = __ballot(m != 0) & lt;
        const int pred = filled ? 63 - __clzll((long long)filled) : 0;
        const uint64_t pl = ((uint64_t)__shfl((uint32_t)(last >> 32), pred) << 32) | __shfl((uint32_t)last, pred);
        const uint32_t f0 = (m && filled && opens_partition(a.f, pl, first, a.max_dist)) ? 1u : 0u;
        uint32_t sc = cnt + f0;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const uint32_t t = __shfl_up(sc, k);
            if (lane >= k) sc += t;
        }
        if ((uint32_t)lane == blockIdx.x) { s_in[0] = sc - cnt; s_in[1] = 0; }  // earlier windows + own leading flag
        if (lane == 63 && blockIdx.x == 0) *a.n_parts = sc + 1;
    }
    __syncthreads();
    const uint32_t add = s_in[0];
    PAIR_CLK(9);
    for (uint32_t j = tid; j < M; j += 1024) {
        a.part_id[start + j] = add + ID[j];
        a.perm[start + j] = I(src)[j];
    }
    PAIR_CLK(10);
}

__global__ __launch_bounds__(1024) void k_pair_single(SingleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ uint32_t s_cnt[16 * kFine];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_rng[4];
    __shared__ uint32_t s_soff[kSingleGridMax], s_scnt[kSingleGridMax];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t g = blockIdx.x, G = gridDim.x;
    const uint32_t fine_bits = a.live < kFineBits ? a.live : kFineBits, low_bits = a.live - fine_bits;
    for (int v = tid; v < (int)kFine; v += 1024) s_cnt[v] = 0;
    if (tid == 0) { s_rng[0] = 0; s_rng[1] = 0; s_rng[2] = 0; s_rng[3] = 0; }
    __syncthreads();
    PAIR_CLK(0);
    // ---- slice pass
    {
        const uint32_t sl = g * a.slice_len, sh = min(a.n, sl + a.slice_len);
        for (uint32_t base = sl + wave * 64; base < sh; base += 1024) {
            const uint32_t i = base + lane;
            const bool valid = i < sh;
            uint32_t fine = ~0u;
            if (valid) {
                const uint64_t d = dense_key(a.f, a.keys[i]);
                agent_store(a.dense + i, d);
                fine = (uint32_t)(d >> low_bits);
            }
            count_runs(s_cnt, fine, valid, lane);
        }
        __syncthreads();
        // the row is stored as its inclusive prefix over the buckets: a stretch of buckets is two loads
        uint32_t all;
        const uint32_t incl = block_scan_1024(tid < (int)kFine ? s_cnt[tid] : 0u, s_w, &all);
        if (tid < (int)kFine) agent_store(a.rows + (size_t)g * kFine + tid, incl);
    }
    PAIR_CLK(1);
    if (!grid_barrier(a.counter, a.counter_stale, a.note)) return;
    PAIR_CLK(2);
    // ---- windows: column sums of the prefix rows = keys in buckets 0..b over all slices (every load of a
    // thread in flight at once: thread t sums bucket t % 512 over one half of the slices)
    {
        const uint32_t b = tid & (kFine - 1), s0 = (tid >> 9) * 32;
        uint32_t x[32], cp = 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u] = s0 + u < G ? agent_load(a.rows + (size_t)(s0 + u) * kFine + b) : 0u;
#pragma unroll
        for (int u = 0; u < 32; ++u) cp += x[u];
        s_cnt[tid] = cp;
    }
    __syncthreads();
    // bucket b belongs to window (keys before b) / target: the buckets of a window are consecutive, the
    // first and the last one announce themselves (empty buckets at the edges count as members)
    if (tid < (int)kFine) {
        const uint32_t w_lo = g * a.target, w_hi = w_lo + a.target;  // (no overflow: g * target < n + target)
        const uint32_t cp = s_cnt[tid] + s_cnt[kFine + tid];                          // keys in buckets 0..tid
        const uint32_t ex = tid ? s_cnt[tid - 1] + s_cnt[kFine + tid - 1] : 0u;       // keys before bucket tid
        const uint32_t ex_prev = tid > 1 ? s_cnt[tid - 2] + s_cnt[kFine + tid - 2] : 0u;
        const bool mine = ex >= w_lo && ex < w_hi;
        const bool prev_mine = tid > 0 && ex_prev >= w_lo && ex_prev < w_hi;
        const bool next_mine = tid + 1 < (int)kFine && cp >= w_lo && cp < w_hi;
        if (mine && !prev_mine) { s_rng[0] = (uint32_t)tid; s_rng[2] = ex; }
        if (mine && !next_mine) { s_rng[1] = (uint32_t)tid + 1u; s_rng[3] = cp; }
    }
    __syncthreads();
    const uint32_t lo = s_rng[0], hi = s_rng[1], start = s_rng[2], M = s_rng[3] - s_rng[2];
    PAIR_CLK(3);
    if (M == 0) {  // no bucket starts inside this window's stretch
        if (tid == 0) { agent_store(a.win + 2 * g, 0u); agent_store(a.win + 2 * g + 1, 0u); }
        (void)grid_barrier(a.counter + 1, nullptr, a.note);
        return;
    }
    // every slice's share of the window and where it goes
    if (wave == 0) {
        uint32_t c = 0;
        if ((uint32_t)lane < G) {
            const uint32_t upto = agent_load(a.rows + (size_t)lane * kFine + hi - 1);
            const uint32_t below = lo ? agent_load(a.rows + (size_t)lane * kFine + lo - 1) : 0u;
            c = upto - below;
        }
        uint32_t sc = c;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const uint32_t t = __shfl_up(sc, k);
            if (lane >= k) sc += t;
        }
        s_soff[lane] = sc - c;
        s_scnt[lane] = c;
    }
    __syncthreads();
    PAIR_CLK(4);
    if (M <= kWinCap) window_work<true>(a, s_dyn, s_cnt, s_w, s_soff, s_scnt, lo, hi, start, M, low_bits);
    else window_work<false>(a, s_dyn, s_cnt, s_w, s_soff, s_scnt, lo, hi, start, M, low_bits);
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

int pair_single_launch(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n, uint32_t max_dist, const KeyFields& f,
                       uint32_t live, uint32_t* d_perm, uint32_t* d_part_id, uint32_t* d_n_parts) {
    if (!ctx->pair_lds_set) {  // once per context: the attribute is kept per device
        SVX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_single),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSingleDynLds));
        ctx->pair_lds_set = true;
    }
    SingleArgs a;
    // two arrival barriers: the grid stays at a quarter of the CUs (one of these workgroups fills a CU's
    // LDS), so that four such grids are resident together and none can starve another's barrier
    const uint32_t grid_max = std::min<uint32_t>(kSingleGridMax, std::max<uint32_t>(1u, (uint32_t)ctx->n_cu / 4));
    a.target = std::max<uint32_t>((n + grid_max - 1) / grid_max, 1024u);
    const uint32_t grid = (n + a.target - 1) / a.target;
    a.slice_len = ((n + grid - 1) / grid + 63) / 64 * 64;
    size_t need = svx_take_bytes(n, 8) + svx_take_bytes(2 * (size_t)n, 8) + svx_take_bytes(2 * (size_t)n, 4) +
                  svx_take_bytes((size_t)grid * kFine, 4) +
                  svx_take_bytes(2 * grid, 4) + svx_take_bytes(2 * grid, 8);
    int rc = svx_ws_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    a.keys = d_keys;
    a.dense = svx_ws_take<uint64_t>(ctx, n);
    a.gkey = svx_ws_take<uint64_t>(ctx, 2 * (size_t)n);
    a.gidx = svx_ws_take<uint32_t>(ctx, 2 * (size_t)n);
    a.rows = svx_ws_take<uint32_t>(ctx, (size_t)grid * kFine);
    a.win = svx_ws_take<uint32_t>(ctx, 2 * grid);
    a.win_keys = svx_ws_take<uint64_t>(ctx, 2 * grid);
    a.f = f;
    a.n = n;
    a.live = live;
    a.max_dist = max_dist;
    a.perm = d_perm;
    a.part_id = d_part_id;
    a.n_parts = d_n_parts;
    // workspace header, words 68..71: two sets of two arrival counters, alternate launches alternate sets
    a.counter = reinterpret_cast<uint32_t*>(ctx->ws) + 68 + 2 * (ctx->pair_launches & 1u);
    a.counter_stale = reinterpret_cast<uint32_t*>(ctx->ws) + 68 + 2 * ((ctx->pair_launches & 1u) ^ 1u);
    a.note = reinterpret_cast<uint32_t*>(ctx->ws) + kBarrierNoteWord;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_pair_single, dim3(grid), dim3(1024), kSingleDynLds, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    ++ctx->pair_launches;  // (only a launch that went out used its set of counters)
    ctx->barrier_pending = true;
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    return svx_timing_end(ctx);
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
    if (n <= ctx->pair_single_max && !ctx->pair_wait_free) return pair_single_launch(ctx, d_keys, n, max_dist, b.f, live, d_perm, d_part_id, d_n_parts);
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
    if (ctx->pair_wait_free) {
        hipLaunchKernelGGL(k_partition<1>, dim3((n + span - 1) / span), dim3(1024), 0, ctx->stream, pa, span);
        hipLaunchKernelGGL(k_partition<2>, dim3((n + span - 1) / span), dim3(1024), 0, ctx->stream, pa, span);
        SVX_HIP(ctx, hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_partition<0>, dim3((n + span - 1) / span), dim3(1024), 0, ctx->stream, pa, span);
        SVX_HIP(ctx, hipGetLastError());
        ctx->barrier_pending = true;
    }
    return svx_timing_end(ctx);
}

}  // namespace

#ifdef SVX_EXP_PAIRCLK
extern "C" int svx_debug_pair_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_clk), sizeof(g_pair_clk));
}
#endif

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
    rc = svx_barrier_check(ctx);
    if (rc != SVX_E_HIP || ctx->pair_wait_free) return rc;
    // A wait between the workgroups of that launch ran out (its other workgroups were not resident in time: too
    // many tenants on the device).  The keys are still staged: run the call again on the plan that never waits
    // inside a launch — radix passes and the partition sweep as two launches — instead of failing it.
    ctx->pair_wait_free = true;
    rc = pair_partition_bits(ctx, d_k, n, max_dist, bits, d_p, d_id, d_np);
    ctx->pair_wait_free = false;
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(perm, d_p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(part_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(n_parts, d_np, 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->err[0] = 0;
    ++ctx->pair_retries;
    return SVX_OK;
}

extern "C" int svx_ctx_set_pair_wait_free(svx_ctx* ctx, int enabled) {
    if (!ctx) return SVX_E_INVALID;
    ctx->pair_wait_free = enabled != 0;
    return SVX_OK;
}

extern "C" int svx_ctx_pair_retries(const svx_ctx* ctx) { return ctx ? (int)ctx->pair_retries : 0; }
