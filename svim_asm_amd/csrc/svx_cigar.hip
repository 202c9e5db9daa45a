// svx_cigar.hip — CIGAR walk → indel signatures on gfx950 (CDNA4, wave64).
//
// Replaces analyze_cigar_indel (reference SVIM_intra.py:8-30) + the `ref_start + pos_ref`
// of analyze_alignment_indel (SVIM_intra.py:36-43) for a whole batch of alignments.
//
// Data layout in HBM
//   cigar[n_ops]      u32, BAM-native `len << 4 | op`, all alignments back to back (4 B/op, read ONCE)
//   aln_off[n_aln+1]  u64 offsets; ref_start[n_aln] i32
//   out SoA           aln u32, ref_pos u32, read_pos u32, len u32, type u8 (17 B/signature)
//
// Pipeline, streaming path (no inter-workgroup waiting anywhere, so no dispatch-order assumption):
//   0  k_tile_alo             per tile, the number of alignments that start before it.
//   A  k_cigar_tiles          one WAVE per tile of 4096 ops; 4 rounds of 1024 ops; per round:
//                             coalesced dwordx4 loads (1 KiB/wave-instr) → wave-private XOR-swizzled LDS
//                             transpose → 16 consecutive ops per lane → lane-local segmented walk →
//                             wave scans (DPP row_shr/row_bcast) → signatures staged into the tile's
//                             slab (128 x 16 B) with tile-local cursors; tile descriptor
//                             {count, seen_head, ref_tail, read_tail, a_lo} written at the end.
//   B  k_desc_scan            segmented exclusive scan over the tile descriptors: per-tile carry-in
//                             (cursor sums since the last alignment start before the tile), output
//                             base (exclusive signature count) and the list of dense tiles.
//   C  k_cigar_finish         16 lanes per sparse tile: slab → final SoA at out_base, adding the
//                             carry to signatures that precede the tile's first alignment start
//                             and ref_start of the alignment.
//   D  k_cigar_dense          dense tiles (> 128 signatures, e.g. adversarial all-indel CIGARs) are
//                             re-walked with carry-in and output base known (process_tile<DIRECT>),
//                             writing final SoA; an empty launch in the common case.
// Batches of at most 8 M ops (both haplotype BAMs of an assembly: what the svim-asm CLI launches) take a two-launch
// variant of the same code: k_cigar_tiles with tiles of 1024 ops and the tile's start index from a wave
// search, then k_cigar_finish_small, in which every workgroup scans all (folded) descriptors itself, finishes its
// own 16 tiles and re-walks its share of the dense ones, dealt out round-robin (see the comment above that kernel).
// With chimeric reads in the submission (svx_collect_batch_dev) the split-segment chain of SVIM_inter.py:62-340 rides
// inside these launches (a3_chain_block): rows + decision tree in the tile launch of the two-launch path
// (k_tiles_a3) or in the finish launch of the streaming path (k_finish_a3), the post-passes in the last launch.
// Output order = (alignment, op) order by construction (prefix sums, no atomically-ordered appends).
#include "svx_internal.h"
#include "svx_postpass_dev.h"
#include "svx_segments_dev.h"

namespace {

#ifndef SVX_LANE_OPS
#define SVX_LANE_OPS 16
#endif
constexpr int kLaneOps = SVX_LANE_OPS;          // consecutive ops per lane per round (16 or 32)
constexpr int kLU = kLaneOps / 4;              // uint4 groups per lane per round
constexpr int kRoundOps = 64 * kLaneOps;       // 1024 ops per wave round (4 KiB)
#ifndef SVX_ROUNDS
#define SVX_ROUNDS 4
#endif
constexpr int kRounds = SVX_ROUNDS;
constexpr int kTileOps = kRoundOps * kRounds;  // 4096 ops per tile (16 KiB)
#ifndef SVX_SLAB
#define SVX_SLAB 128
#endif
// slab records per tile (1/32 of its ops); fuller tiles take the dense path.  The slabs are written
// once (one burst per tile) and read once: a 2 KiB stride instead of 4 KiB is worth 5 % of the
// streaming kernel (DRAM locality of the bursts), 1 KiB another 2 % but too tight for SV-dense regions
constexpr int kSlab = SVX_SLAB;
#ifndef SVX_WAVES
#define SVX_WAVES 4
#endif
constexpr int kWaves = SVX_WAVES;              // waves (= tiles) per workgroup
constexpr int kXposeU4 = 64 * kLU;             // transpose buffer: kLU uint4 per lane, XOR-swizzled
constexpr int kScanBlock = 1024;               // tile descriptors per scan workgroup

enum { MODE_STAGE = 0, MODE_DIRECT = 1 };

struct CigarArgs {
    const uint32_t* cigar;  // packed words, or len[] in SoA mode
    const uint8_t* op;      // SoA mode only
    const uint64_t* aln_off;
    uint32_t* tile_alo;   // per tile: number of alignments that start before the tile (k_tile_alo)
    const int32_t* ref_start;  // nullable
    uint64_t n_ops;
    uint32_t n_aln;
    uint32_t n_tiles;
    uint32_t min_len;
    uint4* desc;
    uint4* desc4;  // small-batch path: one folded descriptor per workgroup of four tiles
    uint4* slab;
    uint32_t* out_base;
    uint32_t* carry_ref;
    uint32_t* carry_read;
    uint32_t* dense_list;
    uint32_t* n_dense;   // [0] dense-tile counter, [1] scan ticket (both left at 0 by the scan), [2] published count
    uint4* blk_agg;      // per scan block: {has_start, ref_tail, read_tail, count}
    uint4* blk_prefix;   // exclusive scan of blk_agg
    svx_sig_soa out;
    uint64_t cap;
};

__device__ __forceinline__ void wave_lds_sync() {
    // wave-private LDS hand-off: LDS ops of one wave execute in order; this only stops the
    // compiler from moving the accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// first index i in [0, n) with a[i] >= key (n if none); all 64 lanes cooperate (64-ary search)
__device__ __forceinline__ uint32_t wave_lower_bound(const uint64_t* __restrict__ a, uint32_t n,
                                                     uint64_t key, int lane) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 128) {
        uint32_t step = (hi - lo + 63) / 64;
        uint64_t idx = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
        bool less = (idx < hi) ? (a[idx] < key) : false;
        uint32_t p = __popcll(__ballot(less));
        uint64_t nlo = (uint64_t)lo + (uint64_t)p * step;
        uint64_t nhi = (uint64_t)lo + (uint64_t)(p + 1) * step - 1;
        lo = (uint32_t)(nlo < hi ? nlo : hi);
        hi = (uint32_t)(nhi < hi ? nhi : hi);
    }
    // last level: up to 128 candidates, two per lane, both loads in flight together (for the ~5 k
    // alignments of a human assembly the whole search is two dependent rounds of loads)
    const uint32_t i0 = lo + lane, i1 = lo + 64 + lane;
    const uint64_t v0 = i0 < hi ? a[i0] : ~0ull, v1 = i1 < hi ? a[i1] : ~0ull;
    const bool less0 = i0 < hi && v0 < key, less1 = i1 < hi && v1 < key;
    return lo + __popcll(__ballot(less0)) + __popcll(__ballot(less1));
}

// largest a in [0, n_aln) with aln_off[a] <= g, given aln_off[a] < tile_start for all a < a_lo
__device__ __forceinline__ uint32_t find_aln(const uint64_t* __restrict__ aln_off, uint32_t n_aln,
                                             uint32_t a_lo, uint64_t g) {
    uint32_t lo = a_lo > 0 ? a_lo - 1 : 0;
    uint32_t step = 1;
    uint64_t probe = (uint64_t)lo + step;
    while (probe < n_aln && aln_off[probe] <= g) {
        lo = (uint32_t)probe;
        step <<= 1;
        probe = (uint64_t)lo + step;
    }
    uint32_t hi = probe < n_aln ? (uint32_t)probe : n_aln;
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (aln_off[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void store_final(const CigarArgs& p, uint64_t slot, uint32_t aln,
                                            uint32_t ref, uint32_t read, uint32_t len,
                                            uint32_t type) {
    if (slot < p.cap) {
        uint32_t rs = p.ref_start ? (uint32_t)p.ref_start[aln] : 0u;
        // default cache policy on purpose: neighbouring tiles complete each other's lines in L2
        // (streaming stores here cost k_cigar_finish 17 us per 1.6 GB batch)
        p.out.aln[slot] = aln;
        p.out.ref_pos[slot] = ref + rs;
        p.out.read_pos[slot] = read;
        p.out.len[slot] = len;
        p.out.type[slot] = (uint8_t)type;
    }
}

// DPP lane movement (gfx9 family): identity 0 flows into lanes without a source.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138;

// inclusive wave scan: segmented (flag f resets) sums of r and d, plain sum of c
#define SVX_SEG_STEP(CTRL, RM)                                                     \
    {                                                                              \
        const uint32_t f2 = dpp0<CTRL, RM>(f), r2 = dpp0<CTRL, RM>(sr),            \
                       d2 = dpp0<CTRL, RM>(sd), c2 = dpp0<CTRL, RM>(sc);           \
        sr += f ? 0u : r2;                                                         \
        sd += f ? 0u : d2;                                                         \
        f |= f2;                                                                   \
        sc += c2;                                                                  \
    }
#define SVX_SEG_SCAN()                                                             \
    SVX_SEG_STEP(kDppShr1, 0xF) SVX_SEG_STEP(kDppShr2, 0xF) SVX_SEG_STEP(kDppShr4, 0xF) \
    SVX_SEG_STEP(kDppShr8, 0xF) SVX_SEG_STEP(kDppBcast15, 0xA) SVX_SEG_STEP(kDppBcast31, 0xC)

// One round of a tile: lane reads uint4 #(k*64+lane), k = 0..3 (coalesced 1 KiB per wave
// instruction) through a per-tile buffer resource whose num_records is the tile's valid byte
// count: the hardware range check returns 0 for anything past the end of the batch (word 0 =
// "0M", a no-op for both cursors), so the ragged last tile needs no guarded scalar path.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// swizzle of the transpose buffer: lane c's p-th uint4 group lives at c*kLU + (p ^ xswz(c)); with
// kLU groups per lane this spreads any 16 consecutive lanes over the 16 uint4 slots of a 256-B row
__device__ __forceinline__ int xswz(int c) { return (c / (16 / kLU)) & (kLU - 1); }

#ifndef SVX_LOAD_AUX
#define SVX_LOAD_AUX 2  // cache policy of the streaming loads: 2 = nt (read once; 8.5 % faster than the default policy)
#endif
template <bool SOA>
__device__ __forceinline__ void load_round(__amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t ro_, uint32_t tile_len,
                                           uint32_t ro, int lane, uint4 (&q)[kLU], uint32_t (&o)[kLU]) {
#ifdef SVX_EXP_NOLOAD  // perf experiment only: no HBM traffic
    for (int k = 0; k < kLU; ++k) { q[k] = make_uint4(ro + lane, (400u << 4), (3u << 4) | 1u, (77u << 4)); o[k] = 0; }
    return;
#endif
#pragma unroll
    for (int k = 0; k < kLU; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            rc, (int)((ro + (uint32_t)(k * 64 + lane) * 4u) * 4u), 0, SVX_LOAD_AUX);
        q[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
    if (SOA) {  // the op codes of this lane's kLaneOps consecutive ops, 4 per dword
#pragma unroll
        for (int k = 0; k < kLU / 4; ++k) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                ro_, (int)(ro + (uint32_t)lane * kLaneOps + (uint32_t)k * 16u), 0, SVX_LOAD_AUX);
            o[4 * k + 0] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        }
        // bytes at or beyond the tile's last op (ragged last tile only) become op 0
#pragma unroll
        for (int j = 0; j < kLU; ++j) {
            const uint32_t first = ro + (uint32_t)lane * kLaneOps + (uint32_t)j * 4u;  // tile-local op of byte 0
            const uint32_t valid = first >= tile_len ? 0u : (tile_len - first >= 4u ? 4u : tile_len - first);
            o[j] &= valid >= 4u ? 0xFFFFFFFFu : ((1u << (8u * valid)) - 1u);
        }
    }
}

#ifndef SVX_TILE_MIN_WAVES
#define SVX_TILE_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for
#endif
#ifndef SVX_STAGE
#define SVX_STAGE 64
#endif
constexpr int kStage = SVX_STAGE;  // finished records staged in LDS per wave before one burst to the slab (>= kQueue)
#ifndef SVX_QUEUE
#define SVX_QUEUE 64  // (32 until round 3: a fifth of the 1024-op rounds of an SV-dense contig overflowed it)
#endif
#ifndef SVX_QUEUE_SMALL
#define SVX_QUEUE_SMALL 96
#endif
// Signatures one round may queue in LDS (flushed 64 at a time, one lane each); a round with more marks its tile for
// the dense re-walk.  The one-round tiles of the two-launch path take 96: there the re-walk sits on the critical path of
// a 25 us submission (a diploid sample's rounds with 65..96 signatures cost its finish launch 12.1 instead of 4.9 us);
// the streaming kernel stays at 64 — its dense tiles are a launch of their own beside 250 us, and the larger queue
// cost it 1-2 us on the dense probes (profiles/r05_ab_queue96.txt).
template <int TILE_OPS>
constexpr int queue_cap() { return TILE_OPS == kRoundOps ? SVX_QUEUE_SMALL : SVX_QUEUE; }
constexpr int kQueue = SVX_QUEUE;
static_assert(SVX_QUEUE <= 128 && SVX_QUEUE % 32 == 0 && SVX_QUEUE_SMALL <= 128 && SVX_QUEUE_SMALL % 32 == 0,
              "the flush takes the queue in at most two passes of 64 lanes");
// LDS words per wave for the start mask, which the queue re-uses once the mask has moved to registers
template <int TILE_OPS>
constexpr int head_words() { return (TILE_OPS / 32) > queue_cap<TILE_OPS>() * 4 ? (TILE_OPS / 32) : queue_cap<TILE_OPS>() * 4; }
constexpr int kHeadWords = head_words<kTileOps>();
// a round's records beyond the stage buffer's room (only a round with more than kStage of them) go straight to the slab
static_assert(kStage == 64, "drain() moves the stage buffer with one lane per record");
constexpr uint32_t kDescForceDense = 1u << 30;  // descriptor flag: a round overflowed the queue

enum { WALK_TOTALS = 0, WALK_QUEUE = 1, WALK_DIRECT = 2 };

// per-lane results of one walk over the lane's 16 consecutive ops
struct WalkOut {
    uint32_t tot_r, tot_d;    // plain cursor sums over the lane's 16 ops
    uint32_t tail_r, tail_d;  // cursor sums since the last alignment start inside the lane (or lane start)
    uint32_t n_emit;          // emitting ops of this lane
    uint32_t n_queued;        // wave-uniform: signatures queued this round (WALK_QUEUE)
};

// Everything WALK_DIRECT needs to finish a signature on the spot.  The alignment index is carried along the
// walk (it advances at the lane's own alignment starts) together with that alignment's ref_start, so a
// signature costs its five stores and no load; only batches with empty alignments (`dup`: two starts on one
// op, which the start mask cannot count) search aln_off, and only at a start.
struct DirectCtx {
    uint32_t in_r, in_d;  // lane carry-in (since the last start before the lane)
    uint64_t out0;        // output slot of this lane's first signature
    uint32_t a_lo;
    uint64_t g_lane0;     // global op index of the lane's op 0
    uint32_t aln0, rs0;   // alignment of the op before the lane's first one (0xFFFFFFFF: none) and its ref_start
    bool dup;
};

// Lane-local walk (SVIM_intra.py:13-29).  The 16 words stay in LDS (`myx`, 4 per uint4) and the
// loop over the four groups is rolled, so the working set is a handful of registers.  Alignment
// starts and emitting ops are rare: both are handled under wave-uniform branches (HU / ballot),
// the common per-op path is decode + two masked adds.
//
// FAST24 (packed layout only, every length of the round < 2^24 — checked by the caller): the op
// code never leaves the raw word.  v_bfe reads its bit offset from the word's low 5 bits, so the
// 9-bit op tables are replicated at bit 16 (offset op or op + 16, whatever the length's lowest bit
// is); the masked add becomes one 24-bit multiply-add per cursor, and "I or D and long enough"
// one compare of (word & -isID(op)) against (min_len << 4, at least 1).  8 VALU per op, not 11.
template <int WALK, bool SOA, bool FAST24, int QCAP = kQueue>
__device__ __forceinline__ WalkOut walk16(const CigarArgs& p, const uint4* myx, int swz, const uint32_t (&opw)[kLU],
                                          uint32_t hm, uint32_t HU, int lane, uint4* queue,
                                          const DirectCtx& dc) {
    static_assert(!(SOA && FAST24), "FAST24 reads op and length from one packed word");
    // emit threshold on the packed word: len >= min_len  <=>  (len << 4 | op) >= min_len << 4 for
    // min_len >= 1; for min_len == 0 every I/D op qualifies and its word is >= 1 (op bits)
    const uint32_t thr = p.min_len >= (1u << 28) ? 0xFFFFFFFFu : (p.min_len ? p.min_len << 4 : 1u);
    uint32_t rr = 0, rd = 0, base_r = 0, base_d = 0, n_emit = 0, qn = 0;
    uint32_t hs = 0;  // this lane has passed an alignment start
    uint32_t aln_cur = dc.aln0, rs_cur = dc.rs0;  // WALK_DIRECT only
#ifdef SVX_EXP_NOWALK  // perf experiment only: memory + scan floor without the per-op work
    { const uint4 v = myx[swz]; WalkOut o; o.tot_r = v.x; o.tot_d = v.y; o.tail_r = v.z; o.tail_d = v.w; o.n_emit = 0; o.n_queued = 0; return o; }
#endif
    uint4 nxt = myx[swz];
#pragma unroll 1
    for (int j = 0; j < kLU; ++j) {
        const uint4 v4 = nxt;
        nxt = myx[((j + 1) & (kLU - 1)) ^ swz];  // LDS read of the next group overlaps this group's math
        const uint32_t wv[4] = {v4.x, v4.y, v4.z, v4.w};
        const uint32_t hu4 = HU >> (4 * j), hm4 = hm >> (4 * j);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t op, len;
            if (SOA) { op = (opw[j] >> (t * 8)) & 0xFFu; len = wv[t]; }
            else { op = wv[t] & 15u; len = wv[t] >> 4; }
            if (__builtin_expect(((hu4 >> t) & 1u) != 0u, 0)) {  // scalar test: some lane starts an alignment at this slot
                asm volatile("" ::: "memory");  // keep this a real (rarely taken) branch, not two selects per op
                if ((hm4 >> t) & 1u) {
                    base_r = rr; base_d = rd; hs = 1u;
                    if (WALK == WALK_DIRECT) {
                        aln_cur = dc.dup ? find_aln(p.aln_off, p.n_aln, dc.a_lo, dc.g_lane0 + (uint32_t)(4 * j + t)) : aln_cur + 1u;
                        rs_cur = p.ref_start ? (uint32_t)p.ref_start[aln_cur] : 0u;
                    }
                }
            }
            // I or D with len >= min_len (inclusive threshold, :18,:22): compares straight into
            // scalar masks; the per-lane predicate is only derived inside the rarely taken branch
            bool emits;
            uint64_t eb;
            if (FAST24) {
                emits = (wv[t] & (uint32_t)__builtin_amdgcn_sbfe(0x00060006, wv[t], 1)) >= thr;
                eb = __builtin_amdgcn_ballot_w64(emits);
            } else {
                emits = (op - 1u) < 2u && len >= p.min_len;
                eb = __builtin_amdgcn_ballot_w64((op - 1u) < 2u) & __builtin_amdgcn_ballot_w64(len >= p.min_len);
            }
            if (__builtin_expect(eb != 0ull, 0)) {  // wave-uniform: most op slots emit nothing
                if (emits) {
                    const uint32_t i = 4 * j + t;
                    if (WALK == WALK_QUEUE) {
                        // rank among the lanes emitting at this slot: mbcnt over the scalar mask
                        const uint32_t qi = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(eb >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)eb, 0u));
                        if (qi < (uint32_t)QCAP) {
                            // meta: lane | slot << 6 | (start at/before op i inside this lane) << 11 |
                            //       index among the lane's signatures << 12 | op << 18 (SoA only)
                            const uint32_t meta = (uint32_t)lane | (i << 6) | (hs << 11) | (n_emit << 12) |
                                                  (SOA ? (op << 18) : 0u);
                            queue[qi] = make_uint4(rr - base_r, rd - base_d, SOA ? len : wv[t], meta);
                        }
                    } else if (WALK == WALK_DIRECT) {
                        const uint32_t ref = rr - base_r + (hs ? 0u : dc.in_r);
                        const uint32_t rdp = rd - base_d + (hs ? 0u : dc.in_d);
                        const uint64_t slot = dc.out0 + n_emit;
                        if (slot < p.cap) {
                            p.out.aln[slot] = aln_cur;
                            p.out.ref_pos[slot] = ref + rs_cur;
                            p.out.read_pos[slot] = rdp;
                            p.out.len[slot] = len;
                            p.out.type[slot] = (uint8_t)((op == 2u) ? SVX_SIG_DEL : SVX_SIG_INS);
                        }
                    }
                    ++n_emit;
                }
                qn += __popcll(eb);
            }
            // ops advancing the reference cursor: M(0) D(2) =(7) X(8); the query cursor:
            // M(0) I(1) S(4) =(7) X(8)   (SVIM_intra.py:14-29; N,H,P,B and codes >= 10: nothing)
            if (FAST24) {
                rr += __umul24(len, __builtin_amdgcn_ubfe(0x01850185u, wv[t], 1));
                rd += __umul24(len, __builtin_amdgcn_ubfe(0x01930193u, wv[t], 1));
            } else {
                const uint32_t sop = SOA ? (op < 16u ? op : 15u) : op;
                rr += len & (uint32_t)__builtin_amdgcn_sbfe(0x185, sop, 1);
                rd += len & (uint32_t)__builtin_amdgcn_sbfe(0x193, sop, 1);
            }
        }
    }
    WalkOut o;
    o.tot_r = rr;
    o.tot_d = rd;
    o.tail_r = rr - base_r;
    o.tail_d = rd - base_d;
    o.n_emit = n_emit;
    o.n_queued = qn;
    return o;
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= dpp0<kDppShr1, 0xF>(v); v |= dpp0<kDppShr2, 0xF>(v); v |= dpp0<kDppShr4, 0xF>(v);
    v |= dpp0<kDppShr8, 0xF>(v); v |= dpp0<kDppBcast15, 0xA>(v); v |= dpp0<kDppBcast31, 0xC>(v);
    return __builtin_amdgcn_readlane(v, 63);
}


#ifdef SVX_EXP_PROF  // perf experiment only (tools/prof_phases.py): per-tile phase clocks
constexpr uint32_t kProfTiles = 32768;
__device__ uint32_t g_prof[kProfTiles * 8];  // 5 phase sums (shader clocks), lifetime and begin in 100 MHz ticks
#define SVX_PROF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define SVX_PROF_ADD(i, d) prof_acc[i] += (uint32_t)(d)
#else
#define SVX_PROF_T(v)
#define SVX_PROF_ADD(i, d)
#endif

// What a tile needs from outside.  MODE_STAGE: where a_lo (alignments that start before the tile)
// comes from; MODE_DIRECT: a_lo, the carry-in and the output base, all known to the caller.
enum { ALO_TABLE = 0, ALO_SEARCH = 1, ALO_GIVEN = 2 };
struct TileIn {
    uint32_t a_lo, carry_r, carry_d, obase;
};

// One tile (TILE_OPS <= 4096 ops) processed by one wave.  MODE_STAGE: signatures go to the tile's slab
// with tile-local cursors; MODE_DIRECT: carry-in and output base are known, signatures are final.
template <int MODE, bool SOA, int TILE_OPS, int ALO>
__device__ __forceinline__ uint4 process_tile(const CigarArgs& p, const uint32_t tile, const int lane, uint4* xp,
                                             uint32_t* hmask, uint4* queue, uint4* stage, const TileIn& in) {
    static_assert(TILE_OPS % kRoundOps == 0 && TILE_OPS <= kTileOps, "a tile is 1..kRounds whole rounds");
    uint4* lcarry = xp;  // per-lane carry-ins reuse the transpose buffer once the walk has consumed it
    const uint64_t g0 = (uint64_t)tile * TILE_OPS;
    const uint64_t tile_end = (g0 + TILE_OPS < p.n_ops) ? g0 + TILE_OPS : p.n_ops;

    // first round's loads go out before the (latency-bound) alignment-start lookup
    const uint32_t tile_len = (uint32_t)(tile_end - g0);
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(p.cigar + g0, tile_len * 4u);
    // SoA op bytes: the hardware range check works per DWORD, so the resource covers the tile's bytes
    // rounded up to 4 (the op array is read at most 3 bytes past n_ops, inside its last dword; those
    // bytes are masked off in load_round — they would otherwise read as ops of length 0)
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(SOA ? (const void*)(p.op + g0) : (const void*)p.cigar,
                                                  SOA ? ((tile_len + 3u) & ~3u) : 0u);
#ifdef SVX_EXP_PROF
    uint32_t prof_acc[8] = {};
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    SVX_PROF_T(t_begin);
    uint4 q[kLU];
    uint32_t qo[kLU] = {};
    load_round<SOA>(rs_c, rs_o, tile_len, 0u, lane, q, qo);

    // ---- alignment starts inside this tile → 4096-bit mask in LDS; `dup` = two alignments start
    // at the same op (empty alignments), which disables the popcount shortcut for the index ----
    // wave-uniform: one scalar load (start table of k_tile_alo), a 64-ary search (single-launch path), or given
    const uint32_t a_lo = ALO == ALO_TABLE ? p.tile_alo[tile]
                        : ALO == ALO_SEARCH ? (g0 ? wave_lower_bound(p.aln_off, p.n_aln, g0, lane) : 0u) : in.a_lo;
#pragma unroll
    for (int i = lane; i < kTileOps / 32; i += 64) hmask[i] = 0;
    wave_lds_sync();
    bool dup = false;
    // hu: for every round, the op slots (0..kLaneOps-1) at which ANY lane starts an alignment —
    // gathered here once per tile (kLaneOps == 16: four 16-bit fields in 64 bits) instead of a
    // wave OR-reduction of the lanes' masks in every round
    uint32_t hu_lo = 0, hu_hi = 0;
    for (uint64_t a = (uint64_t)a_lo + lane;; a += 64) {
        bool in = false, twice = false;
        if (a < p.n_aln) {
            uint64_t off = p.aln_off[a];
            if (off < tile_end) {
                in = true;
                const uint32_t bit = (uint32_t)(off - g0);
                twice = (atomicOr(&hmask[bit >> 5], 1u << (bit & 31)) >> (bit & 31)) & 1u;
                if (kLaneOps == 16 && kRounds <= 4) {
                    const uint32_t f = ((bit / kRoundOps) & 1u) * 16u + (bit & 15u);
                    if ((bit / kRoundOps) & 2u) hu_hi |= 1u << f; else hu_lo |= 1u << f;
                }
            }
        }
        dup = dup || __any(twice);
        if (!__all(in)) break;
    }
    if (kLaneOps == 16 && kRounds <= 4) {
        hu_lo = wave_or_u32(hu_lo);
        hu_hi = wave_or_u32(hu_hi);
    }
    wave_lds_sync();
    // this lane's start bits of all rounds move to two registers; the mask's LDS then serves as the
    // per-round signature queue (1 KiB at 64 entries, 1.5 KiB at the one-round tiles' 96) — six 4-wave workgroups
    // stay on a CU either way
    static_assert(kLaneOps == 16 && kRounds <= 4, "the queue aliases the start mask: 16 ops per lane, at most 4 rounds");
    constexpr int kQ = queue_cap<TILE_OPS>();
    static_assert(kQ * sizeof(uint4) <= head_words<TILE_OPS>() * sizeof(uint32_t), "queue must fit the start mask's LDS");
    uint32_t hm01 = 0, hm23 = 0;
    if (kLaneOps == 16 && kRounds <= 4) {
        const uint32_t sh = ((uint32_t)lane & 1u) * 16u, wi = (uint32_t)lane >> 1;
        const uint32_t h0 = (hmask[wi] >> sh) & 0xFFFFu;
        const uint32_t h1 = kRounds > 1 ? (hmask[32 + wi] >> sh) & 0xFFFFu : 0u;
        const uint32_t h2 = kRounds > 2 ? (hmask[64 + wi] >> sh) & 0xFFFFu : 0u;
        const uint32_t h3 = kRounds > 3 ? (hmask[96 + wi] >> sh) & 0xFFFFu : 0u;
        hm01 = h0 | (h1 << 16);
        hm23 = h2 | (h3 << 16);
        wave_lds_sync();
    }

    SVX_PROF_T(t_pro);
    SVX_PROF_ADD(0, t_pro - t_begin);
    // ---- tile state carried across rounds (wave-uniform) ----
    uint32_t carry_r = 0, carry_d = 0;
    bool seen = false, overflow = false;
    uint32_t tile_cnt = 0, heads_before = 0;
    // Finished records wait in LDS (`stage`, kStage entries = slab ranks stage_base ..) and go to the
    // tile's slab in one contiguous burst — at the end of the tile for all but the densest ones.
    // Four small scattered 16-byte stores per round cost 10-15 % of the kernel (every one opens
    // another DRAM row in the middle of the read stream).
    uint32_t stage_base = 0;
    auto drain = [&](uint32_t upto) {  // slab ranks [stage_base, upto) leave LDS
        const uint32_t n = upto - stage_base;
        if ((uint32_t)lane < n && stage_base + (uint32_t)lane < (uint32_t)kSlab) {
            // streaming store: the slab is written once and read once by k_cigar_finish
            const uint4 v = stage[lane];
            u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p.slab + (uint64_t)tile * kSlab + stage_base + lane));
        }
        stage_base = upto;
    };
    uint32_t obase = 0;
    if (MODE == MODE_DIRECT) {
        carry_r = in.carry_r;
        carry_d = in.carry_d;
        seen = true;  // carry-in already holds "since the last start before the tile"
        obase = in.obase;
    }

    for (int round = 0; round < kRounds; ++round) {
        const uint32_t ro = (uint32_t)round * kRoundOps;
        if (ro >= tile_len) break;  // wave-uniform

        SVX_PROF_T(t_r0);
        // ---- transpose through wave-private LDS.  uint4 #i (= lane's k-th load) belongs to lane
        // c = i / kLU as its p = i % kLU -th group; it is stored at c*kLU + (p ^ xswz(c)), which keeps
        // both the ds_write_b128 and the per-group ds_read_b128 bank-conflict free ----
        // every length of the round below 2^24 (any real CIGAR): the walk may use 24-bit multiply-adds
        bool fast24 = false;
        if (!SOA) {
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < kLU; ++k) any |= q[k].x | q[k].y | q[k].z | q[k].w;
            fast24 = __builtin_amdgcn_ballot_w64((any >> 28) != 0u) == 0ull;
        }
#pragma unroll
        for (int k = 0; k < kLU; ++k) {
            const int i = k * 64 + lane;
            const int c = i / kLU;
            xp[c * kLU + ((i & (kLU - 1)) ^ xswz(c))] = q[k];
        }
        uint32_t opw[kLU];
#pragma unroll
        for (int k = 0; k < kLU; ++k) opw[k] = qo[k];
        // software pipeline: next round's global loads are in flight during this round's math
        if (round + 1 < kRounds && ro + kRoundOps < tile_len)
            load_round<SOA>(rs_c, rs_o, tile_len, ro + kRoundOps, lane, q, qo);
        wave_lds_sync();
        SVX_PROF_T(t_r1);
        SVX_PROF_ADD(1, t_r1 - t_r0);  // wait for the round's data + transpose
        const uint32_t lbase = round * kRoundOps + lane * kLaneOps;  // tile-local index of op 0
        const uint32_t hm = (kLaneOps == 16 && kRounds <= 4)
                                ? (((round & 2) ? hm23 : hm01) >> ((round & 1) * 16)) & 0xFFFFu
                                : (kLaneOps == 32 ? hmask[lbase >> 5]
                                                  : (hmask[lbase >> 5] >> (lbase & 31)) & ((1u << (kLaneOps & 31)) - 1u));
        // slots where ANY lane starts an alignment (SGPR)
        const uint32_t HU = (kLaneOps == 16 && kRounds <= 4) ? (((round & 2) ? hu_hi : hu_lo) >> ((round & 1) * 16)) & 0xFFFFu
                                                              : wave_or_u32(hm);
        const uint4* myx = xp + lane * kLU;   // this lane's consecutive ops, 4 per uint4 (swizzled)
        const int swz = xswz(lane);

        DirectCtx dc;
        dc.in_r = 0; dc.in_d = 0; dc.out0 = 0; dc.a_lo = a_lo; dc.g_lane0 = g0 + lbase;
        dc.aln0 = 0; dc.rs0 = 0; dc.dup = dup;
        constexpr int kWalk1 = (MODE == MODE_STAGE) ? WALK_QUEUE : WALK_TOTALS;
        const WalkOut wo = (!SOA && fast24) ? walk16<kWalk1, SOA, !SOA, kQ>(p, myx, swz, opw, hm, HU, lane, queue, dc)
                                            : walk16<kWalk1, SOA, false, kQ>(p, myx, swz, opw, hm, HU, lane, queue, dc);

        SVX_PROF_T(t_r2);
        SVX_PROF_ADD(2, t_r2 - t_r1);  // walk
        // ---- wave scans (DPP).  Plain inclusive sums of the lane totals, signature counts and
        // alignment-start counts; the segmentation is applied afterwards: the carry-in of lane l is
        // its exclusive sum plus Q(h) = tail(h) - P(h) of the last lane h < l that holds an
        // alignment start (ballot + one ds_bpermute per cursor), or plus the round's carry. ----
        uint32_t pr = wo.tot_r, pd = wo.tot_d, sc = wo.n_emit | ((uint32_t)__popc(hm) << 16);
#define SVX_ADD3_STEP(CTRL, RM) \
        pr += dpp0<CTRL, RM>(pr); pd += dpp0<CTRL, RM>(pd); sc += dpp0<CTRL, RM>(sc);
        SVX_ADD3_STEP(kDppShr1, 0xF) SVX_ADD3_STEP(kDppShr2, 0xF) SVX_ADD3_STEP(kDppShr4, 0xF)
        SVX_ADD3_STEP(kDppShr8, 0xF) SVX_ADD3_STEP(kDppBcast15, 0xA) SVX_ADD3_STEP(kDppBcast31, 0xC)
#undef SVX_ADD3_STEP
        // sc packs two counters: signatures (low 16 bits, <= 1024) and alignment starts (high 16 bits)
        const uint32_t xr = dpp0<kDppWaveShr1, 0xF>(pr), xd = dpp0<kDppWaveShr1, 0xF>(pd),
                       xch = dpp0<kDppWaveShr1, 0xF>(sc);
        const uint32_t xc = xch & 0xFFFFu;
        const uint32_t CH = __builtin_amdgcn_readlane(sc, 63);
        const uint32_t C = CH & 0xFFFFu;
        const uint64_t H = __builtin_amdgcn_ballot_w64(hm != 0);
        const uint64_t hl = H & ((1ull << lane) - 1ull);
        const uint32_t qr = wo.tail_r - pr, qd = wo.tail_d - pd;
        const bool xf = hl != 0;  // a start in an earlier lane of this round
        const int hsrc = xf ? 63 - __clzll((long long)hl) : 0;
        const uint32_t gq_r = (uint32_t)__builtin_amdgcn_ds_bpermute(hsrc << 2, (int)qr);
        const uint32_t gq_d = (uint32_t)__builtin_amdgcn_ds_bpermute(hsrc << 2, (int)qd);
        const uint32_t in_r = xr + (xf ? gq_r : carry_r);   // lane carry-in
        const uint32_t in_d = xd + (xf ? gq_d : carry_d);

        if (MODE == MODE_STAGE) {
            if (C) {  // wave-uniform
                if (tile_cnt + C - stage_base > (uint32_t)kStage) {  // no room for this round's records
                    drain(tile_cnt);
                    wave_lds_sync();
                }
                // .z: signatures before this lane | alignment starts before this lane in the round << 12
                //     | "still lacks the tile's carry-in" << 31;   .w: this lane's start mask
                lcarry[lane] = make_uint4(in_r, in_d, xc | ((xch >> 16) << 12) | ((!seen && !xf) ? 0x80000000u : 0u), hm);
                wave_lds_sync();
                const uint32_t n_here = wo.n_queued < (uint32_t)kQ ? wo.n_queued : (uint32_t)kQ;
                for (uint32_t qb = 0; qb < n_here; qb += 64u)  // (a second pass only for a round with more than 64)
                if (qb + (uint32_t)lane < n_here) {
                    const uint4 e = queue[qb + lane];
                    const uint32_t L = e.w & 63u, slot = (e.w >> 6) & 31u, own = (e.w >> 11) & 1u,
                                   li = (e.w >> 12) & 63u;
                    const uint4 cin = lcarry[L];
                    const uint32_t ref = own ? e.x : e.x + cin.x;
                    const uint32_t rdp = own ? e.y : e.y + cin.y;
                    const uint32_t prec = own ? 0u : (cin.z >> 31);
                    uint32_t op, len;
                    if (SOA) { op = (e.w >> 18) & 0xFFu; len = e.z; }
                    else { op = e.z & 15u; len = e.z >> 4; }
                    const uint32_t type = (op == 2u) ? SVX_SIG_DEL : SVX_SIG_INS;
                    const uint32_t rank = tile_cnt + (cin.z & 0xFFFu) + li;
                    // alignment index: a_lo - 1 + (alignment starts at or before the op inside the tile)
                    const uint32_t m = heads_before + ((cin.z >> 12) & 0xFFFu) + __popc(cin.w & ((slot == 31u) ? 0xFFFFFFFFu : ((2u << slot) - 1u)));
                    uint32_t aln = a_lo + m - 1u;
                    if (dup) aln = find_aln(p.aln_off, p.n_aln, a_lo, g0 + round * kRoundOps + L * kLaneOps + slot);
                    const uint4 rec = make_uint4(aln, ref, rdp, len | (type << 28) | (prec << 29));
                    if (rank - stage_base < (uint32_t)kStage) stage[rank - stage_base] = rec;
                    else if (kQ > kStage && rank < (uint32_t)kSlab)  // the round's records past the stage buffer
                        p.slab[(uint64_t)tile * kSlab + rank] = rec;
                }
                if (wo.n_queued > (uint32_t)kQ) overflow = true;
            }
        } else {
            if (C) {  // dense tile: second walk finishes each signature on the spot
                dc.in_r = in_r; dc.in_d = in_d;
                dc.out0 = (uint64_t)obase + tile_cnt + xc;
                // the alignment the lane's first op continues: a_lo - 1 + the starts before the lane inside the tile
                // (none before the batch's first op: that op is a start itself)
                dc.aln0 = !dup ? a_lo + heads_before + (xch >> 16) - 1u
                               : (dc.g_lane0 ? find_aln(p.aln_off, p.n_aln, a_lo, dc.g_lane0 - 1u) : 0xFFFFFFFFu);
                dc.rs0 = (p.ref_start && dc.aln0 < p.n_aln) ? (uint32_t)p.ref_start[dc.aln0] : 0u;
                (void)walk16<WALK_DIRECT, SOA, false>(p, myx, swz, opw, hm, HU, lane, queue, dc);
            }
        }
        wave_lds_sync();  // queue / lcarry / xp are rewritten by the next round
        SVX_PROF_T(t_r3);
        SVX_PROF_ADD(3, t_r3 - t_r2);  // scans + flush

        // ---- carry to the next round (wave-uniform) ----
        const uint32_t R = __builtin_amdgcn_readlane(pr, 63), D = __builtin_amdgcn_readlane(pd, 63);
        if (H) {
            const int hlast = 63 - __clzll((long long)H);
            carry_r = R + __builtin_amdgcn_readlane(qr, hlast);
            carry_d = D + __builtin_amdgcn_readlane(qd, hlast);
            seen = true;
        } else {
            carry_r += R;
            carry_d += D;
        }
        tile_cnt += C;
        heads_before += CH >> 16;
    }

#ifdef SVX_EXP_PROF
    SVX_PROF_T(t_end);
    SVX_PROF_ADD(4, t_end - t_begin);
    prof_acc[5] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt_begin);
    prof_acc[6] = (uint32_t)rt_begin;
    if (MODE == MODE_STAGE && lane < 8) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v = (lane == i) ? prof_acc[i] : v;
        g_prof[(tile % kProfTiles) * 8 + lane] = v;
    }
#endif
    if (MODE == MODE_STAGE) {
        const uint32_t hi = tile_cnt - stage_base < (uint32_t)kStage ? tile_cnt : stage_base + (uint32_t)kStage;
        drain(hi);
    }
    const uint4 dsc = make_uint4(tile_cnt | (overflow ? kDescForceDense : 0u) | ((seen ? 1u : 0u) << 31), carry_r, carry_d, a_lo);
    if (MODE == MODE_STAGE && lane == 0) p.desc[tile] = dsc;
    return dsc;  // wave-uniform
}

// Small-batch path: the four tiles of a workgroup fold their descriptors into one (segmented sum in tile order),
// so that the scan every workgroup of k_cigar_finish_small runs covers a quarter of the entries.
__device__ __forceinline__ void fold_group_desc(const CigarArgs& p, uint4* s_agg, const int wave, const int lane, const uint4 dsc,
                                                const uint32_t group) {
    if (lane == 0) s_agg[wave] = dsc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t f = 0, r = 0, d = 0, c = 0, dense = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            const uint4 v = s_agg[k];
            if (v.x >> 31) { f = 1; r = v.y; d = v.z; }
            else { r += v.y; d += v.z; }
            c += v.x & 0x3FFFFFFFu;
            // .w: which of the four tiles k_cigar_finish_small has to walk again (beyond the slab, or a round beyond the queue)
            if ((v.x & 0x3FFFFFFFu) > (uint32_t)kSlab || (v.x & kDescForceDense)) dense |= 1u << k;
        }
        p.desc4[group] = make_uint4(c | (f << 31), r, d, dense);
    }
}

// ---- A0: per tile, the number of alignments that start before it (= lower bound of the tile's
// first op in aln_off).  One thread per alignment: alignment a is the last one starting before
// tile t exactly when aln_off[a] < t*kTileOps <= aln_off[a+1], so every tile t >= 1 has exactly one
// writer and the streaming kernel's prologue needs no search (four dependent loads per tile). ----
__global__ __launch_bounds__(256) void k_tile_alo(const uint64_t* __restrict__ aln_off, uint32_t n_aln,
                                                  uint32_t n_tiles, uint32_t* __restrict__ tile_alo) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n_aln) return;
    if (a == 0) tile_alo[0] = 0;
    const uint64_t lo = aln_off[a], hi = aln_off[a + 1];
    uint64_t t = lo / kTileOps + 1, t_hi = hi / kTileOps;
    if (t_hi >= n_tiles) t_hi = n_tiles - 1;
    for (; t <= t_hi; ++t) tile_alo[t] = a + 1;
}

// ---- A: stream every tile once ----
template <bool SOA, int TILE_OPS, int ALO>
__global__ __launch_bounds__(64 * kWaves, SVX_TILE_MIN_WAVES) void k_cigar_tiles(CigarArgs p) {
    __shared__ uint4 s_xpose[kWaves][kXposeU4];
    __shared__ __attribute__((aligned(16))) uint32_t s_head[kWaves][head_words<TILE_OPS>()];  // start mask, then the queue
    __shared__ uint4 s_stage[kWaves][kStage];
    // the wave index is wave-uniform: tell the compiler so that tile indices, loop bounds and
    // carries live in SGPRs and the tile/round loops are scalar branches
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (TILE_OPS == kRoundOps) {  // small-batch path (one round per tile): the grid covers every tile once
        __shared__ uint4 s_agg[kWaves];
        const uint32_t tile = blockIdx.x * kWaves + wave;
        uint4 dsc = make_uint4(0, 0, 0, 0);
        if (tile < p.n_tiles)
            dsc = process_tile<MODE_STAGE, SOA, TILE_OPS, ALO>(p, tile, lane, s_xpose[wave], s_head[wave],
                                                              reinterpret_cast<uint4*>(s_head[wave]), s_stage[wave], TileIn());
        fold_group_desc(p, s_agg, wave, lane, dsc, blockIdx.x);
        return;
    }
    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < p.n_tiles; tile += gridDim.x * kWaves)
        (void)process_tile<MODE_STAGE, SOA, TILE_OPS, ALO>(p, tile, lane, s_xpose[wave], s_head[wave],
                                                                  reinterpret_cast<uint4*>(s_head[wave]), s_stage[wave], TileIn());
}

// ---- B: segmented exclusive scan over tile descriptors ----
// Each workgroup scans kScanBlock descriptors locally and publishes its aggregate; the LAST
// workgroup to arrive (ticket counter, agent-scope fences, no spinning) scans the block
// aggregates into blk_prefix.  Consumers combine local values with blk_prefix[tile / kScanBlock].
// Per tile: carry_ref/carry_read = cursor sums since the last alignment start inside the scan
// block (bit 31 of out_base set) or since the block start (bit clear → add the block prefix).
__global__ __launch_bounds__(kScanBlock) void k_desc_scan(const uint4* __restrict__ desc, uint32_t n_tiles,
                                                          uint32_t* __restrict__ out_base,
                                                          uint32_t* __restrict__ carry_ref,
                                                          uint32_t* __restrict__ carry_read,
                                                          uint32_t* __restrict__ dense_list,
                                                          uint32_t* __restrict__ n_dense,
                                                          uint4* __restrict__ blk_agg,
                                                          uint4* __restrict__ blk_prefix,
                                                          uint32_t* __restrict__ ticket,
                                                          uint64_t* __restrict__ n_out) {
    __shared__ uint32_t s_f[16], s_r[16], s_d[16], s_c[16];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_blocks = gridDim.x;
    {
        const uint32_t t = blockIdx.x * kScanBlock + tid;
        uint32_t f = 0, sr = 0, sd = 0, sc = 0;
        if (t < n_tiles) {
            const uint4 v = desc[t];
            sc = v.x & 0x3FFFFFFFu;
            f = v.x >> 31;
            sr = v.y;
            sd = v.z;
            if (sc > (uint32_t)kSlab || (v.x & kDescForceDense)) dense_list[atomicAdd(n_dense, 1u)] = t;
        }
        SVX_SEG_SCAN()
        if (lane == 63) { s_f[wave] = f; s_r[wave] = sr; s_d[wave] = sd; s_c[wave] = sc; }
        __syncthreads();
        uint32_t pf = 0, pr_ = 0, pd_ = 0, pc = 0;  // prefix over preceding waves
        uint32_t af = 0, ar = 0, ad = 0, ac = 0;    // aggregate over all waves
        for (int w2 = 0; w2 < kScanBlock / 64; ++w2) {
            if (w2 == wave) { pf = af; pr_ = ar; pd_ = ad; pc = ac; }
            if (s_f[w2]) { af = 1; ar = s_r[w2]; ad = s_d[w2]; }
            else { ar += s_r[w2]; ad += s_d[w2]; }
            ac += s_c[w2];
        }
        const uint32_t xf = dpp0<kDppWaveShr1, 0xF>(f), xr = dpp0<kDppWaveShr1, 0xF>(sr),
                       xd = dpp0<kDppWaveShr1, 0xF>(sd), xc = dpp0<kDppWaveShr1, 0xF>(sc);
        if (t < n_tiles) {
            carry_ref[t] = xf ? xr : pr_ + xr;
            carry_read[t] = xf ? xd : pd_ + xd;
            out_base[t] = (pc + xc) | ((xf | pf) << 31);
        }
        // the aggregate is the only thing another workgroup of THIS launch reads: stored write-through
        // (sc0 sc1) and read back with sc1 loads by the last workgroup, so no cache-wide release /
        // acquire fence is needed (each costs microseconds: MI355X_MICROARCH.md, inter-workgroup visibility)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its dense-tile atomics have been performed
        __syncthreads();
        if (tid == 0) {
            u32x4 w; w.x = af; w.y = ar; w.z = ad; w.w = ac;
            __builtin_amdgcn_raw_buffer_store_b128(w, make_rsrc(blk_agg, n_blocks * 16u), (int)(blockIdx.x * 16u), 0, 17);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = (atomicAdd(ticket, 1u) == n_blocks - 1) ? 1u : 0u;
        }
    }
    // ---- last-arriving workgroup scans the block aggregates ----
    __syncthreads();
    if (!s_last) return;
    uint32_t cf = 0, cr = 0, cd = 0;
    uint64_t cc = 0;
    for (uint32_t base = 0; base < n_blocks; base += kScanBlock) {
        const uint32_t b = base + tid;
        uint32_t f = 0, sr = 0, sd = 0, sc = 0;
        if (b < n_blocks) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(blk_agg, n_blocks * 16u), (int)(b * 16u), 0, 17);
            f = v.x; sr = v.y; sd = v.z; sc = v.w;
        }
        SVX_SEG_SCAN()
        __syncthreads();
        if (lane == 63) { s_f[wave] = f; s_r[wave] = sr; s_d[wave] = sd; s_c[wave] = sc; }
        __syncthreads();
        uint32_t pf = cf, pr_ = cr, pd_ = cd;
        uint64_t pc = cc;
        uint32_t af = cf, ar = cr, ad = cd;
        uint64_t ac = cc;
        for (int w2 = 0; w2 < kScanBlock / 64; ++w2) {
            if (w2 == wave) { pf = af; pr_ = ar; pd_ = ad; pc = ac; }
            if (s_f[w2]) { af = 1; ar = s_r[w2]; ad = s_d[w2]; }
            else { ar += s_r[w2]; ad += s_d[w2]; }
            ac += s_c[w2];
        }
        const uint32_t xf = dpp0<kDppWaveShr1, 0xF>(f), xr = dpp0<kDppWaveShr1, 0xF>(sr),
                       xd = dpp0<kDppWaveShr1, 0xF>(sd), xc = dpp0<kDppWaveShr1, 0xF>(sc);
        if (b < n_blocks)
            blk_prefix[b] = make_uint4(xf | pf, xf ? xr : pr_ + xr, xf ? xd : pd_ + xd, (uint32_t)(pc + xc));
        cf = af; cr = ar; cd = ad; cc = ac;
    }
    if (tid == 0) {
        *n_out = cc;
        // publish the dense-tile count and leave the counters zeroed for the next call
        n_dense[2] = atomicExch(&n_dense[0], 0u);  // (read where the other workgroups' atomics were performed)
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- C: finish, sparse tiles: 16 lanes per tile copy the staged records into the final SoA at the
// scanned output base (a tile of 4096 ops carries ~30 signatures: lane = signature), adding the
// tile carry-in where the record lacks it and ref_start of its alignment.  No LDS, few registers:
// the kernel is a chain of two dependent loads per tile and lives on occupancy. ----
constexpr int kFinLanes = 16;
#ifndef SVX_FINSPEC
#define SVX_FINSPEC 3
#endif
constexpr int kFinSpec = SVX_FINSPEC;  // records per lane requested together with the descriptor (48 per tile)
__device__ __forceinline__ void cigar_finish_block(const CigarArgs& p, const uint32_t block) {
    const uint32_t tile = block * (256u / kFinLanes) + threadIdx.x / kFinLanes;
    const uint32_t l = threadIdx.x % kFinLanes;
    if (tile >= p.n_tiles) return;
    // speculative: slots past the tile's count hold stale bytes and are never used
    uint4 spec[kFinSpec];
#pragma unroll
    for (int k = 0; k < kFinSpec; ++k) spec[k] = p.slab[(uint64_t)tile * kSlab + l + k * kFinLanes];
    const uint4 dsc = p.desc[tile];
    const uint4 bp = p.blk_prefix[tile / kScanBlock];
    const uint32_t lb = p.out_base[tile];
    const uint32_t lcr = p.carry_ref[tile], lcd = p.carry_read[tile];
    const uint32_t cnt = dsc.x & 0x3FFFFFFFu;
    if (cnt == 0 || cnt > (uint32_t)kSlab || (dsc.x & kDescForceDense)) return;  // dense: k_cigar_dense
    const bool local_head = (lb >> 31) != 0;
    const uint32_t cr = lcr + (local_head ? 0u : bp.y);
    const uint32_t cd = lcd + (local_head ? 0u : bp.z);
    const uint64_t ob = (uint64_t)(lb & 0x7FFFFFFFu) + bp.w;
    // the ref_start gathers of both speculative records go out together
    uint32_t rs[kFinSpec];
#pragma unroll
    for (int k = 0; k < kFinSpec; ++k)
        rs[k] = (l + k * kFinLanes < cnt && p.ref_start) ? (uint32_t)p.ref_start[spec[k].x] : 0u;
#pragma unroll
    for (int k = 0; k < kFinSpec; ++k) {
        const uint32_t r = l + k * kFinLanes;
        if (r < cnt && ob + r < p.cap) {
            const uint4 rec = spec[k];
            const uint32_t len = rec.w & 0x0FFFFFFFu, type = (rec.w >> 28) & 1u, prec = (rec.w >> 29) & 1u;
            p.out.aln[ob + r] = rec.x;
            p.out.ref_pos[ob + r] = rec.y + (prec ? cr : 0u) + rs[k];
            p.out.read_pos[ob + r] = rec.z + (prec ? cd : 0u);
            p.out.len[ob + r] = len;
            p.out.type[ob + r] = (uint8_t)type;
        }
    }
    for (uint32_t r = l + kFinSpec * kFinLanes; r < cnt; r += kFinLanes) {
        const uint4 rec = p.slab[(uint64_t)tile * kSlab + r];
        const uint32_t len = rec.w & 0x0FFFFFFFu, type = (rec.w >> 28) & 1u, prec = (rec.w >> 29) & 1u;
        store_final(p, ob + r, rec.x, rec.y + (prec ? cr : 0u), rec.z + (prec ? cd : 0u), len, type);
    }
}

__global__ __launch_bounds__(256) void k_cigar_finish(CigarArgs p) { cigar_finish_block(p, blockIdx.x); }

// ---- D: dense tiles (more than kSlab signatures, or a round that overflowed the queue: SV-dense stretches of
// an assembly, satellite arrays, a tiny min_len) are re-walked with carry-in and output base known.  Always
// launched — the host cannot know — and empty in the common case (every workgroup reads the count and leaves).
// One WORKGROUP per dense tile, one wave per round of 1024 ops: the four rounds are walked side by side (totals
// walk + wave scan), exchange their totals through LDS — the carry chain across the rounds is four scalar steps —
// and emit side by side.  (Round 3 gave a tile to one wave, which walked the rounds one after the other and
// searched aln_off once per signature: 25 us per tile, 73 us for the product's full-size sample.) ----
struct RoundTotals {
    uint32_t r, d, tail_r, tail_d, cnt, heads, seen, pad;
};

template <bool SOA>
__device__ __forceinline__ void dense_tile_wg(const CigarArgs& p, const uint32_t tile, const int wave, const int lane,
                                              uint4* xp, uint32_t* hmask, uint32_t* s_dup, RoundTotals* s_round,
                                              const TileIn& in) {
    static_assert(kRounds == kWaves && kLaneOps == 16, "one wave per round");
    const uint64_t g0 = (uint64_t)tile * kTileOps;
    const uint64_t tile_end = (g0 + kTileOps < p.n_ops) ? g0 + kTileOps : p.n_ops;
    const uint32_t tile_len = (uint32_t)(tile_end - g0);
    const uint32_t ro = (uint32_t)wave * kRoundOps;
    const bool live = ro < tile_len;  // wave-uniform: the ragged last tile may not reach this wave's round
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(p.cigar + g0, tile_len * 4u);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(SOA ? (const void*)(p.op + g0) : (const void*)p.cigar,
                                                  SOA ? ((tile_len + 3u) & ~3u) : 0u);
    uint4 q[kLU];
    uint32_t qo[kLU] = {};
    if (live) load_round<SOA>(rs_c, rs_o, tile_len, ro, lane, q, qo);
    // ---- the tile's start mask, built by the whole workgroup
    const int tid = wave * 64 + lane;
    if (tid < kTileOps / 32) hmask[tid] = 0;
    if (tid == 0) *s_dup = 0;
    __syncthreads();
    for (uint64_t a = (uint64_t)in.a_lo + tid;; a += 64 * kWaves) {
        bool inside = false;
        if (a < p.n_aln) {
            const uint64_t off = p.aln_off[a];
            if (off < tile_end) {
                inside = true;
                const uint32_t bit = (uint32_t)(off - g0);
                if ((atomicOr(&hmask[bit >> 5], 1u << (bit & 31)) >> (bit & 31)) & 1u) *s_dup = 1u;
            }
        }
        if (!__syncthreads_and(inside ? 1 : 0)) break;  // offsets ascend: a thread past the tile ends the sweep
    }
    __syncthreads();
    const bool dup = *s_dup != 0;
    const uint32_t lbase = ro + (uint32_t)lane * kLaneOps;
    const uint32_t hm = (hmask[lbase >> 5] >> (lbase & 31)) & 0xFFFFu;
    const uint32_t HU = wave_or_u32(hm);
    uint32_t pr = 0, pd = 0, sc = 0, qr = 0, qd = 0;
    uint64_t H = 0;
    uint32_t opw[kLU];
#pragma unroll
    for (int k = 0; k < kLU; ++k) opw[k] = qo[k];
    const uint4* myx = xp + lane * kLU;
    const int swz = xswz(lane);
    DirectCtx dc;
    dc.in_r = 0; dc.in_d = 0; dc.out0 = 0; dc.a_lo = in.a_lo; dc.g_lane0 = g0 + lbase; dc.aln0 = 0; dc.rs0 = 0; dc.dup = dup;
    if (live) {
#pragma unroll
        for (int k = 0; k < kLU; ++k) {
            const int i = k * 64 + lane;
            const int c = i / kLU;
            xp[c * kLU + ((i & (kLU - 1)) ^ xswz(c))] = q[k];
        }
        wave_lds_sync();
        const WalkOut wo = walk16<WALK_TOTALS, SOA, false>(p, myx, swz, opw, hm, HU, lane, nullptr, dc);
        pr = wo.tot_r; pd = wo.tot_d; sc = wo.n_emit | ((uint32_t)__popc(hm) << 16);
#define SVX_ADD3_STEP(CTRL, RM) \
        pr += dpp0<CTRL, RM>(pr); pd += dpp0<CTRL, RM>(pd); sc += dpp0<CTRL, RM>(sc);
        SVX_ADD3_STEP(kDppShr1, 0xF) SVX_ADD3_STEP(kDppShr2, 0xF) SVX_ADD3_STEP(kDppShr4, 0xF)
        SVX_ADD3_STEP(kDppShr8, 0xF) SVX_ADD3_STEP(kDppBcast15, 0xA) SVX_ADD3_STEP(kDppBcast31, 0xC)
#undef SVX_ADD3_STEP
        H = __builtin_amdgcn_ballot_w64(hm != 0);
        qr = wo.tail_r - pr; qd = wo.tail_d - pd;
    }
    if (lane == 0) {
        RoundTotals t;
        const uint32_t R = __builtin_amdgcn_readlane(pr, 63), D = __builtin_amdgcn_readlane(pd, 63);
        const uint32_t CH = __builtin_amdgcn_readlane(sc, 63);
        const int hlast = H ? 63 - __clzll((long long)H) : 0;
        t.r = R; t.d = D; t.cnt = CH & 0xFFFFu; t.heads = CH >> 16; t.seen = H ? 1u : 0u; t.pad = 0;
        t.tail_r = R + __builtin_amdgcn_readlane(qr, hlast);
        t.tail_d = D + __builtin_amdgcn_readlane(qd, hlast);
        s_round[wave] = t;
    }
    __syncthreads();
    // ---- carry into this wave's round: fold the rounds before it (wave-uniform scalars)
    uint32_t carry_r = in.carry_r, carry_d = in.carry_d, cnt_before = 0, heads_before = 0;
    for (int k = 0; k < wave; ++k) {
        const RoundTotals t = s_round[k];
        if (t.seen) { carry_r = t.tail_r; carry_d = t.tail_d; }
        else { carry_r += t.r; carry_d += t.d; }
        cnt_before += t.cnt;
        heads_before += t.heads;
    }
    if (live && (__builtin_amdgcn_readlane(sc, 63) & 0xFFFFu)) {
        const uint32_t xr = dpp0<kDppWaveShr1, 0xF>(pr), xd = dpp0<kDppWaveShr1, 0xF>(pd), xch = dpp0<kDppWaveShr1, 0xF>(sc);
        const uint64_t hl = H & ((1ull << lane) - 1ull);
        const bool xf = hl != 0;
        const int hsrc = xf ? 63 - __clzll((long long)hl) : 0;
        const uint32_t gq_r = (uint32_t)__builtin_amdgcn_ds_bpermute(hsrc << 2, (int)qr);
        const uint32_t gq_d = (uint32_t)__builtin_amdgcn_ds_bpermute(hsrc << 2, (int)qd);
        dc.in_r = xr + (xf ? gq_r : carry_r);
        dc.in_d = xd + (xf ? gq_d : carry_d);
        dc.out0 = (uint64_t)in.obase + cnt_before + (xch & 0xFFFFu);
        dc.aln0 = !dup ? in.a_lo + heads_before + (xch >> 16) - 1u
                       : (dc.g_lane0 ? find_aln(p.aln_off, p.n_aln, in.a_lo, dc.g_lane0 - 1u) : 0xFFFFFFFFu);
        dc.rs0 = (p.ref_start && dc.aln0 < p.n_aln) ? (uint32_t)p.ref_start[dc.aln0] : 0u;
        (void)walk16<WALK_DIRECT, SOA, false>(p, myx, swz, opw, hm, HU, lane, nullptr, dc);
    }
    __syncthreads();  // the mask, the totals and the transpose buffers are rewritten for the next tile
}

// ---- per-alignment CIGAR statistics: one wave per alignment ----
struct StatsArgs {
    const uint32_t* cigar;
    const uint64_t* aln_off;
    uint32_t n_aln;
    svx_aln_stats out;
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

struct AlnStats {
    uint32_t lead, ref, qal, rl, hard;
};
constexpr int kStatLoads = 8;  // CIGAR words a lane requests before it uses the first

// Statistics of the alignment whose ops are cigar[b .. e), computed by one wave (every lane returns the same):
// lead = Σ leading S (skipping H), ref = Σ{M,D,N,=,X}, qal = Σ{M,I,=,X}, rl = Σ{M,I,S,=,X,H}, hard = Σ H.
__device__ __forceinline__ AlnStats wave_alignment_stats(const uint32_t* cigar, uint64_t b, uint64_t e, int lane) {
    // lead: S ops before the first op that is neither S nor H (pysam getQueryStart) — taken from the
    // words of the main loop's rounds while the prefix is still open (in practice: the first round), so that it costs
    // no round trip of its own
    uint32_t lead = 0, ref = 0, qal = 0, rl = 0, hard = 0;
    bool open = true;  // wave-uniform
    for (uint64_t i0 = b; i0 < e; i0 += kStatLoads * 64) {  // kStatLoads loads in flight per lane: the loop is a chain of round trips
        uint32_t w[kStatLoads];
#pragma unroll
        for (int k = 0; k < kStatLoads; ++k) {
            const uint64_t i = i0 + (uint64_t)(k * 64 + lane);
            w[k] = i < e ? cigar[i] : 0xFu;  // op 15: counts for nothing, ends the prefix
        }
#pragma unroll
        for (int k = 0; k < kStatLoads; ++k) {
            const uint32_t op = w[k] & 15u, len = w[k] >> 4;
            if ((0x18Du >> op) & 1u) ref += len;   // M D N = X  (htslib bam_endpos)
            if ((0x183u >> op) & 1u) qal += len;   // M I = X
            if ((0x1B3u >> op) & 1u) rl += len;    // M I S H = X (infer_read_length)
            if (op == 5u) hard += len;
        }
        if (open) {
#pragma unroll
            for (int k = 0; k < kStatLoads; ++k) {
                const uint32_t op = w[k] & 15u;
                const uint64_t nb = __ballot(!(op == 4u || op == 5u));
                const int first = nb ? __ffsll((unsigned long long)nb) - 1 : 64;
                if (open && lane < first && op == 4u) lead += w[k] >> 4;
                open = open && nb == 0;
            }
        }
    }
    AlnStats r;
    r.lead = wave_sum(lead);
    r.ref = wave_sum(ref); r.qal = wave_sum(qal); r.rl = wave_sum(rl); r.hard = wave_sum(hard);
    return r;
}

// ... and for an alignment of at most kTinyOps ops by ONE lane (an SA-derived segment is S M S): 64 alignments per
// wave, every load of a lane issued before the first is used.
constexpr int kTinyOps = 8;
// (two halves, so that a caller can put other work between the loads and their use)
__device__ __forceinline__ void lane_alignment_load(const uint32_t* cigar, const uint64_t b, const uint32_t n, uint32_t (&w)[kTinyOps]) {
#pragma unroll
    for (int k = 0; k < kTinyOps; ++k) w[k] = (uint32_t)k < n ? cigar[b + k] : 0xFu;
}
__device__ __forceinline__ AlnStats lane_alignment_stats(const uint32_t (&w)[kTinyOps], const uint32_t n) {
    AlnStats r;
    r.lead = r.ref = r.qal = r.rl = r.hard = 0;
    bool prefix = true;
#pragma unroll
    for (int k = 0; k < kTinyOps; ++k) {
        const uint32_t op = w[k] & 15u, len = w[k] >> 4;
        const bool real = (uint32_t)k < n;
        prefix = prefix && real && (op == 4u || op == 5u);
        if (prefix && op == 4u) r.lead += len;
        if ((0x18Du >> op) & 1u) r.ref += len;
        if ((0x183u >> op) & 1u) r.qal += len;
        if ((0x1B3u >> op) & 1u) r.rl += len;
        if (op == 5u) r.hard += len;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_cigar_stats(StatsArgs p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t a = blockIdx.x * 4 + wave; a < p.n_aln; a += gridDim.x * 4) {
        const AlnStats st = wave_alignment_stats(p.cigar, p.aln_off[a], p.aln_off[a + 1], lane);
        if (lane == 0) {
            if (p.out.ref_len) p.out.ref_len[a] = st.ref;
            if (p.out.q_start) p.out.q_start[a] = st.lead;
            if (p.out.q_end) p.out.q_end[a] = st.lead + st.qal;
            if (p.out.read_len) p.out.read_len[a] = st.rl;
            if (p.out.n_hard) p.out.n_hard[a] = st.hard;
        }
    }
}

// ---- segment rows of the chimeric reads (SVIM_inter.py:66-81): one wave per segment ----
struct SegRowArgs {
    const uint32_t* cigar;
    const uint64_t* aln_off;
    const uint32_t* seg_src;   // alignment (index into aln_off) each segment is
    const int32_t* seg_tid;
    const int32_t* seg_pos;
    const uint8_t* seg_rev;
    const int32_t* seg_qend;   // >= 0: query_alignment_end taken from the stored sequence (pysam), else from the CIGAR
    uint32_t n_segs;
    const uint32_t* read_off;  // n_reads + 1
    uint32_t n_reads;
    svx_seg* segs;
    int32_t* read_len;         // per read: infer_read_length() of its first segment (the primary)
};

// the row of SVIM_inter.py:66-81 for segment j from its alignment's CIGAR statistics
__device__ __forceinline__ svx_seg segment_row(const SegRowArgs& p, const uint32_t j, const AlnStats& st) {
    const int32_t q_start = (int32_t)st.lead;
    const int32_t over = p.seg_qend[j];
    const int32_t q_end = over >= 0 ? over : (int32_t)(st.lead + st.qal);
    const int32_t rl = (int32_t)st.rl;
    const bool rev = p.seg_rev[j] != 0;
    svx_seg s;
    s.q_start = rev ? rl - q_end : q_start;      // :68-73 (query coordinates flipped for reverse records)
    s.q_end = rev ? rl - q_start : q_end;
    s.ref_id = p.seg_tid[j];
    s.ref_start = p.seg_pos[j];
    s.ref_end = p.seg_pos[j] + (int32_t)(st.ref ? st.ref : 1u);  // htslib bam_endpos
    s.is_reverse = rev ? 1 : 0;
    return s;
}

__global__ __launch_bounds__(256) void k_segment_rows(SegRowArgs p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t j = blockIdx.x * 4 + wave; j < p.n_segs; j += gridDim.x * 4) {
        const uint32_t a = p.seg_src[j];
        const AlnStats st = wave_alignment_stats(p.cigar, p.aln_off[a], p.aln_off[a + 1], lane);
        if (lane == 0) p.segs[j] = segment_row(p, j, st);
    }
    // read_len[r] = read length of read r's primary = its first segment; recomputed by the wave that owns it
    for (uint32_t r = blockIdx.x * 4 + wave; r < p.n_reads; r += gridDim.x * 4) {
        const uint32_t j = p.read_off[r];
        if (j >= p.n_segs || p.read_off[r + 1] == j) {
            if (lane == 0) p.read_len[r] = 0;
            continue;
        }
        const uint32_t a = p.seg_src[j];
        const AlnStats st = wave_alignment_stats(p.cigar, p.aln_off[a], p.aln_off[a + 1], lane);
        if (lane == 0) p.read_len[r] = (int32_t)st.rl;
    }
}

// ---- the split-segment chain of one submission (SVIM_inter.py:62-340) inside the launches of the CIGAR path: a
// workgroup owns consecutive chimeric reads — the range the caller's table gives it (`deal`: equal CIGAR op counts,
// svx_chain_deal), else `reads_per_block` of them.  Stage A (in the tile launch of the two-launch path, in the finish
// launch of the streaming path): the segment rows (CIGAR statistics -> svx_seg, :66-81: a lane per tiny alignment,
// chunks of 128 ops dealt to the workgroup's sixteen 16-lane groups for all others), a workgroup barrier, the
// adjacent-pair decision tree (eight lanes per read, :83-258).  Stage B (in the last launch of the
// path): the three post-passes (one lane per read, :260-338).  The rows travel through HBM
// and are read back by the workgroup that wrote them, i.e. from the same CU's cache, behind the barrier; the raw
// records are an output anyway.  Round 3 ran the chain as three launches of 4-7 us each behind the CIGAR path.
struct A3Args {
    int32_t* seg_rl;  // per segment: infer_read_length() of its alignment (the tree takes a read's length from its first segment)
    SegRowArgs rows;
    svx_seg_dev::SegArgs tree;
    svx_post_dev::PostArgs post;
    uint32_t reads_per_block;
    const uint32_t* deal;  // nullable: {first read, first segment} per workgroup of the rows + tree stage, n_deal + 1 entries
    uint32_t n_deal;
};

enum { A3_ROWS_TREE = 1, A3_POST = 2 };

// Rows of the alignments beyond kTinyOps: the workgroup cuts them into CHUNKS of kChunkOps ops (kChunkX4 16-byte loads
// per lane of a 16-lane group), numbers the chunks of its batch of 256 segments by an exclusive scan and deals the
// numbers out to its sixteen groups in contiguous, equally long ranges — every group is busy for the same number of
// steps whatever the sizes are, and a long alignment (of ANY length: there is no size class above this one) is shared
// by several groups.  A group finds the segment of its first chunk by a binary search of the scan in LDS and walks on
// from there; it keeps its sums in registers while consecutive chunks belong to one alignment and adds them to the
// alignment's LDS accumulators when that changes.
#ifndef SVX_CHUNK_X4
#define SVX_CHUNK_X4 2
#endif
#ifndef SVX_A3_NT
#define SVX_A3_NT 0
#endif
constexpr int kChunkX4 = SVX_CHUNK_X4;       // 16-byte loads per lane and step
constexpr int kChunkOps = 64 * kChunkX4;     // 16 lanes x 4 words x kChunkX4 = 128
// (Round 4: one word per load, 14 VALU per word, a materialised chunk list in LDS and alignments beyond 2048 ops walked
// by the whole workgroup one at a time — 17.6 of the chain's 44.5 us on the cohort of 256 samples went to those 1.2 %
// of the primaries, because reads were dealt 32 per workgroup whatever their sizes.  Before that, size classes:
// k_finish_a3 74.8 us by classes, 53.5 us by chunks; profiles/r04_ab_chain_rows.txt.)
struct A3Lds {
    uint32_t wsum[4];
    uint32_t open[8];              // bit per slot: the whole first chunk was clips (the owner then walks the prefix itself)
    uint32_t cb_lo[256], cb_hi[256], n_ops[256];   // per slot = thread of the batch of 256 segments
    uint32_t cpre[257 + 3];        // exclusive scan of the slots' chunk counts; [256] = their number
    uint32_t acc[4][256];          // lead, ref, qal, rl
};

// sum over an aligned row of 16 lanes, every lane gets it (four DPP adds)
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp0<kDppShr1, 0xF>(v); v += dpp0<kDppShr2, 0xF>(v); v += dpp0<kDppShr4, 0xF>(v); v += dpp0<kDppShr8, 0xF>(v);
    v += dpp0<kDppBcast15, 0xA>(v); v += dpp0<kDppBcast31, 0xC>(v);
    return v;
}

// four consecutive CIGAR words at any word address (a global 16-byte load only needs dword alignment)
typedef uint32_t u32x4_dw __attribute__((ext_vector_type(4), aligned(4)));

template <int STAGES>
__device__ __forceinline__ void a3_chain_block(const A3Args& a, const uint32_t blk, A3Lds* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_reads = a.rows.n_reads;
    uint32_t r_lo, r_hi, j_lo = 0, j_hi = 0;
    const bool dealt = (STAGES & A3_ROWS_TREE) && a.deal != nullptr;
    if (dealt) {  // the caller's table: {first read, first segment} of every workgroup, equal op counts (svx_chain_deal)
        const uint2 d0 = *reinterpret_cast<const uint2*>(a.deal + 2 * (size_t)blk);
        const uint2 d1 = *reinterpret_cast<const uint2*>(a.deal + 2 * (size_t)blk + 2);
        r_lo = d0.x; j_lo = d0.y; r_hi = d1.x; j_hi = d1.y;
        if (r_hi > n_reads) r_hi = n_reads;  // (a table is the caller's: nothing beyond the submission is indexed)
        if (j_hi > a.rows.n_segs) j_hi = a.rows.n_segs;
        if (r_lo >= r_hi || j_lo > j_hi) return;
    } else {
        const uint32_t per = (STAGES & A3_ROWS_TREE) ? a.reads_per_block : 256u;  // the post-passes: one lane per read
        r_lo = blk * per;
        if (r_lo >= n_reads) return;
        r_hi = r_lo + per < n_reads ? r_lo + per : n_reads;
    }
    if (STAGES & A3_ROWS_TREE) {
        // ---- rows.  One THREAD per segment fetches the segment's CIGAR range; tiny alignments (SA-derived: S M S)
        // are finished by that lane, the others go through the chunks (above).
        const SegRowArgs& p = a.rows;
        if (!dealt) { j_lo = p.read_off[r_lo]; j_hi = p.read_off[r_hi]; }
        if (tid < 8) lds->open[tid] = 0;
        // the tree's own read_off words, requested now: one round trip less behind the rows
        const uint32_t r_tree = r_lo + (uint32_t)tid / svx_seg_dev::kGroup;
        uint32_t tree_b = 0, tree_e = 0;
        if (r_tree < r_hi) { tree_b = p.read_off[r_tree]; tree_e = p.read_off[r_tree + 1]; }
        const int gl = lane & 15;
        const int gid = __builtin_amdgcn_readfirstlane(wave) * 4 + (lane >> 4);  // group of sixteen lanes in the workgroup
        const int gshift = lane & ~15;
        for (uint32_t j0 = j_lo; j0 < j_hi; j0 += 256) {
            const uint32_t j = j0 + (uint32_t)tid;
            const bool live = j < j_hi;
            uint64_t cb = 0, ce = 0;
            if (live) {
                const uint32_t src = p.seg_src[j];
                cb = p.aln_off[src];
                ce = p.aln_off[src + 1];
            }
            const uint64_t n = ce - cb;
            const bool tiny = live && n <= (uint64_t)kTinyOps;
            const bool chunked = live && !tiny;
            uint32_t tw[kTinyOps];
            lane_alignment_load(p.cigar, cb, tiny ? (uint32_t)n : 0u, tw);
            const uint32_t nch = chunked ? (uint32_t)((n + (uint64_t)kChunkOps - 1u) / (uint64_t)kChunkOps) : 0u;
            if (chunked) {
                lds->cb_lo[tid] = (uint32_t)cb;
                lds->cb_hi[tid] = (uint32_t)(cb >> 32);
                lds->n_ops[tid] = (uint32_t)n;
#pragma unroll
                for (int q = 0; q < 4; ++q) lds->acc[q][tid] = 0;
            }
            const uint32_t incl = wave_incl_scan(nch);
            if (lane == 63) lds->wsum[wave] = incl;
            if (tiny) {
                const AlnStats st = lane_alignment_stats(tw, (uint32_t)n);
                p.segs[j] = segment_row(p, j, st);
                a.seg_rl[j] = (int32_t)st.rl;
            }
            __syncthreads();
            {
                uint32_t pre = 0;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) pre += w2 < wave ? lds->wsum[w2] : 0u;
                lds->cpre[tid] = pre + incl - nch;
                if (tid == 255) lds->cpre[256] = pre + incl;
            }
            __syncthreads();
            {
#ifdef SVX_EXP_A3_NOCHUNK  // timing ablation: wrong rows
                const uint32_t total = 0;
#else
                const uint32_t total = lds->cpre[256];
#endif
                const uint32_t per_group = (total + 15u) / 16u;
                const uint32_t c_lo = (uint32_t)gid * per_group < total ? (uint32_t)gid * per_group : total;
                const uint32_t c_hi = c_lo + per_group < total ? c_lo + per_group : total;
                if (c_lo < c_hi) {  // group-uniform from here on
                    // the slot of the group's first chunk: the smallest s with cpre[s + 1] > c_lo
                    uint32_t slot = 0;
                    {
                        uint32_t lo = 0, hi = 255;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (lds->cpre[mid + 1] > c_lo) hi = mid; else lo = mid + 1u;
                        }
                        slot = lo;
                    }
                    uint32_t s_beg = lds->cpre[slot], s_end = lds->cpre[slot + 1];
                    uint32_t nv = lds->n_ops[slot];
                    const uint32_t* src = p.cigar + (((uint64_t)lds->cb_hi[slot] << 32) | lds->cb_lo[slot]);
                    uint32_t lead = 0, ref = 0, qal = 0, rl = 0;
                    for (uint32_t c = c_lo;; ++c) {
                        const bool more = c < c_hi;
                        if (!more || c >= s_end) {  // the sums leave for the slot's accumulators
                            lead = row16_sum(lead); ref = row16_sum(ref); qal = row16_sum(qal); rl = row16_sum(rl);
                            if (gl == 0) {
                                if (lead) atomicAdd(&lds->acc[0][slot], lead);
                                atomicAdd(&lds->acc[1][slot], ref);
                                atomicAdd(&lds->acc[2][slot], qal);
                                atomicAdd(&lds->acc[3][slot], rl);
                            }
                            if (!more) break;
                            do { ++slot; s_end = lds->cpre[slot + 1]; } while (c >= s_end);  // (slots without chunks are skipped)
                            s_beg = lds->cpre[slot];
                            nv = lds->n_ops[slot];
                            src = p.cigar + (((uint64_t)lds->cb_hi[slot] << 32) | lds->cb_lo[slot]);
                            lead = ref = qal = rl = 0;
                        }
                        const uint32_t ci = c - s_beg;
                        const uint32_t rel0 = ci * (uint32_t)kChunkOps + (uint32_t)gl * 4u;
                        // position order inside a chunk: load k, lane, word — every load of the group is 256 contiguous bytes
                        uint32_t w[kChunkX4][4];
#pragma unroll
                        for (int k = 0; k < kChunkX4; ++k) {
                            const uint32_t rel = rel0 + (uint32_t)(k * 64);
                            w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0xFu;  // op 15: counts for nothing, ends the clip prefix
                            if (rel < nv) {
                                const uint32_t rem = nv - rel;
                                if (rem >= 4u) {
#if SVX_A3_NT
                                    const u32x4_dw v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_dw*>(src + rel));
#else
                                    const u32x4_dw v = *reinterpret_cast<const u32x4_dw*>(src + rel);
#endif
                                    w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
                                } else {  // the alignment's last words (nothing behind them is read)
                                    w[k][0] = src[rel];
                                    if (rem > 1u) w[k][1] = src[rel + 1];
                                    if (rem > 2u) w[k][2] = src[rel + 2];
                                }
                            }
                        }
                        // one bit of a 16-entry table per sum and word (v_bfe takes its offset from the word's low five
                        // bits: the table twice, the lowest length bit picks either copy), one 24-bit multiply-add each
                        uint32_t t_ref = 0, t_qal = 0, t_rl = 0, any = 0;
#pragma unroll
                        for (int k = 0; k < kChunkX4; ++k)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const uint32_t x = w[k][t], len = x >> 4;
                                any |= x;
                                t_ref += __umul24(len, __builtin_amdgcn_ubfe(0x018D018Du, x, 1));   // M D N = X
                                t_qal += __umul24(len, __builtin_amdgcn_ubfe(0x01830183u, x, 1));   // M I = X
                                t_rl += __umul24(len, __builtin_amdgcn_ubfe(0x01B301B3u, x, 1));    // M I S H = X
                            }
                        if (__builtin_expect(__ballot((any >> 28) != 0u) != 0ull, 0)) {  // a length of 2^24 or more: no multiply
                            t_ref = t_qal = t_rl = 0;
#pragma unroll
                            for (int k = 0; k < kChunkX4; ++k)
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const uint32_t x = w[k][t], len = x >> 4;
                                    t_ref += len & (uint32_t)__builtin_amdgcn_sbfe(0x018D018D, x, 1);
                                    t_qal += len & (uint32_t)__builtin_amdgcn_sbfe(0x01830183, x, 1);
                                    t_rl += len & (uint32_t)__builtin_amdgcn_sbfe(0x01B301B3, x, 1);
                                }
                        }
                        ref += t_ref; qal += t_qal; rl += t_rl;
                        if (ci == 0) {  // the prefix of clips (S ops before the first op that is neither S nor H)
                            bool open = true;
#pragma unroll
                            for (int k = 0; k < kChunkX4; ++k) {
                                if (open) {
                                    // inside the lane's four words: the clips in front, and the S lengths among them
                                    uint32_t pfx = 1u, spre = 0;
#pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        const uint32_t x = w[k][t];
                                        pfx &= __builtin_amdgcn_ubfe(0x00300030u, x, 1);                          // S H
                                        spre += (x >> 4) & (0u - (pfx & __builtin_amdgcn_ubfe(0x00100010u, x, 1)));  // S
                                    }
                                    const uint32_t nb = (uint32_t)(__ballot(pfx == 0u) >> gshift) & 0xFFFFu;  // lanes whose four words end the prefix
                                    const int first = nb ? __ffs((int)nb) - 1 : 16;
                                    if (gl <= first) lead += spre;
                                    open = nb == 0;
                                }
                            }
                            // (words behind the end read as op 15 and close the prefix: still open = a whole chunk of clips)
                            if (open && gl == 0) atomicOr(&lds->open[slot >> 5], 1u << (slot & 31u));
                        }
                    }
                }
            }
            __syncthreads();
            if (chunked) {
                AlnStats st;
                st.lead = lds->acc[0][tid]; st.ref = lds->acc[1][tid]; st.qal = lds->acc[2][tid]; st.rl = lds->acc[3][tid];
                st.hard = 0;
                if ((lds->open[tid >> 5] >> (tid & 31)) & 1u) {  // S ops before the first op that is neither S nor H, by this lane
                    st.lead = 0;
                    const uint64_t b2 = ((uint64_t)lds->cb_hi[tid] << 32) | lds->cb_lo[tid], e2 = b2 + lds->n_ops[tid];
                    for (uint64_t i = b2; i < e2; ++i) {
                        const uint32_t x = p.cigar[i], op = x & 15u;
                        if (op != 4u && op != 5u) break;
                        if (op == 4u) st.lead += x >> 4;
                    }
                }
                p.segs[j] = segment_row(p, j, st);
                a.seg_rl[j] = (int32_t)st.rl;
            }
            if (j0 + 256 < j_hi) {  // workgroup-uniform: another batch of segments
                __syncthreads();
                if (tid < 8) lds->open[tid] = 0;
            }
        }
        __syncthreads();
        // ---- decision tree: eight lanes per read
#ifdef SVX_EXP_A3_NOTREE
        return;
#endif
        const int tl = lane & (svx_seg_dev::kGroup - 1), gbase = lane & ~(svx_seg_dev::kGroup - 1);
        for (uint32_t r0 = r_lo; r0 < r_hi; r0 += 256 / svx_seg_dev::kGroup) {
            const uint32_t r = r0 + tid / svx_seg_dev::kGroup;
            if (r0 == r_lo) svx_seg_dev::segments_group(a.tree, r, r < r_hi, tl, gbase, tree_b, tree_e);
            else svx_seg_dev::segments_group(a.tree, r, r < r_hi, tl, gbase);
        }
    }
    // ---- post-passes: one lane per read
    if (STAGES & A3_POST)
        for (uint32_t r = r_lo + tid; r < r_hi; r += 256) svx_post_dev::post_one_read(a.post, r);
}

// The dense-tile launch of the streaming path (dense_tile_wg above); with WITH_POST its first n_a3_blocks workgroups
// run the post-passes of the split-segment chain instead — the launch is empty in the common case anyway.
template <bool SOA, bool WITH_POST>
__global__ __launch_bounds__(64 * kWaves) void k_cigar_dense(CigarArgs p, A3Args a3, uint32_t n_a3_blocks) {
    if (WITH_POST && blockIdx.x < n_a3_blocks) {  // workgroup-uniform
        a3_chain_block<A3_POST>(a3, blockIdx.x, nullptr);
        return;
    }
    __shared__ uint4 s_xpose[kWaves][kXposeU4];
    __shared__ uint32_t s_mask[kTileOps / 32];
    __shared__ RoundTotals s_round[kWaves];
    __shared__ uint32_t s_dup;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t n_dense = p.n_dense[2];
    const uint32_t first = WITH_POST ? n_a3_blocks : 0u;
    for (uint32_t work = blockIdx.x - first; work < n_dense; work += gridDim.x - first) {
        const uint32_t tile = p.dense_list[work];
        const uint4 bp = p.blk_prefix[tile / kScanBlock];
        const uint32_t lr = p.carry_ref[tile], ld = p.carry_read[tile], lb = p.out_base[tile];
        const bool local_head = (lb >> 31) != 0;  // a start precedes the tile inside its scan block
        TileIn in;
        in.a_lo = p.tile_alo[tile];
        in.carry_r = local_head ? lr : lr + bp.y;
        in.carry_d = local_head ? ld : ld + bp.z;
        in.obase = (lb & 0x7FFFFFFFu) + bp.w;
        dense_tile_wg<SOA>(p, tile, wave, lane, s_xpose[wave], s_mask, &s_dup, s_round, in);
    }
}

// ---- two-launch path for small batches (one BAM of a human assembly: ~1.5 M ops) ----
// Five dependent launches cost more than the work itself below a chip-load of tiles, and a tile
// of 4096 ops per wave leaves most of the chip idle.  Here: k_cigar_tiles with tiles of 1024 ops
// (one round per wave: 4x the waves for the same batch) and the tile's start index from a 64-ary
// search instead of a table kernel; then ONE kernel in which every workgroup scans ALL tile
// descriptors itself (at most 2048 x 16 B, L2-resident: cheaper than a scan kernel plus a launch
// gap, and no tickets, fences or waiting), copies the records of its own 16 tiles to the final SoA
// and re-walks those of them that are dense.
constexpr int kSmallTileOps = kRoundOps;            // 1024 ops
constexpr int kSmallPer = 8;                        // folded descriptors per thread in the scan
constexpr uint32_t kSmallMaxGroups = 256u * kSmallPer;   // groups of kWaves tiles (one per workgroup of the tile kernel)
constexpr uint32_t kSmallMaxTiles = kSmallMaxGroups * kWaves;  // batches up to 8 M ops take this path
#ifndef SVX_FIN_GROUPS
#define SVX_FIN_GROUPS 4
#endif
constexpr uint32_t kFinGroups = SVX_FIN_GROUPS;     // groups of four tiles a workgroup finishes: one per wave, at most kWaves
constexpr uint32_t kFinTiles = 4u * kFinGroups;     // (16 lanes per tile; with fewer than four groups the other waves only help
                                                    //  with the workgroup's dense tiles)
static_assert(kFinGroups >= 1 && kFinGroups <= (uint32_t)kWaves && kWaves == 4, "a wave of the finish kernel owns one group of four tiles");

template <bool SOA, bool WITH_POST>
__global__ __launch_bounds__(256) void k_cigar_finish_small(CigarArgs p, uint64_t* __restrict__ n_out, A3Args a3, uint32_t n_a3_blocks) {
    if (WITH_POST && blockIdx.x < n_a3_blocks) {  // workgroup-uniform: the post-passes of the split-segment chain
        a3_chain_block<A3_POST>(a3, blockIdx.x, nullptr);
        return;
    }
    const uint32_t block = WITH_POST ? blockIdx.x - n_a3_blocks : blockIdx.x;
    __shared__ uint32_t s_cr[kWaves], s_cd[kWaves], s_ob[kWaves];  // exclusive prefix of this workgroup's groups
    __shared__ uint4 s_dense[kFinTiles];  // tile, and its group's exclusive prefix: carry_ref, carry_read, output base
    __shared__ uint32_t s_n_dense;
    __shared__ uint4 s_xpose[kWaves][kXposeU4];
    __shared__ __attribute__((aligned(16))) uint32_t s_head[kWaves][kHeadWords];
    __shared__ uint32_t s_f[4], s_r[4], s_d[4], s_c[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the tile this thread's 16-lane group finishes: its records are requested before the scan
    const uint32_t t0 = block * kFinTiles, g0 = block * kFinGroups;
    const uint32_t tile = t0 + tid / kFinLanes;
    const uint32_t l = tid % kFinLanes;
    const bool mine = (uint32_t)tid < kFinTiles * kFinLanes && tile < p.n_tiles;
    uint4 spec[kFinSpec];
#pragma unroll
    for (int k = 0; k < kFinSpec; ++k)
        spec[k] = mine ? p.slab[(uint64_t)tile * kSlab + l + k * kFinLanes] : make_uint4(0, 0, 0, 0);
    const uint4 dsc = mine ? p.desc[tile] : make_uint4(0, 0, 0, 0);

    // ---- segmented exclusive scan over the folded descriptors of ALL groups (one per four tiles, written by the
    // tile kernel's workgroups): kSmallPer consecutive ones per thread, DPP scan across the workgroup; only the
    // prefixes of this workgroup's own four groups are kept
    const uint32_t n_groups = (p.n_tiles + kWaves - 1) / kWaves;
    uint32_t f = 0, sr = 0, sd = 0, sc = 0;
    uint4 d[kSmallPer];
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        const uint32_t g = (uint32_t)tid * kSmallPer + i;
        d[i] = g < n_groups ? p.desc4[g] : make_uint4(0, 0, 0, 0);
    }
    uint32_t lr[kSmallPer], ld[kSmallPer], lc[kSmallPer], lf = 0;  // exclusive inside the thread; lf: bit i = a start before item i
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        lr[i] = sr; ld[i] = sd; lc[i] = sc;
        lf |= f << i;
        if (d[i].x >> 31) { f = 1; sr = d[i].y; sd = d[i].z; }
        else { sr += d[i].y; sd += d[i].z; }
        sc += d[i].x & 0x3FFFFFFFu;
    }
    SVX_SEG_SCAN()
    if (lane == 63) { s_f[wave] = f; s_r[wave] = sr; s_d[wave] = sd; s_c[wave] = sc; }
    if (tid == 0) s_n_dense = 0;
    __syncthreads();
    uint32_t pr_ = 0, pd_ = 0, pc = 0, ar = 0, ad = 0, ac = 0;
    for (int w2 = 0; w2 < 4; ++w2) {
        if (w2 == wave) { pr_ = ar; pd_ = ad; pc = ac; }
        if (s_f[w2]) { ar = s_r[w2]; ad = s_d[w2]; }
        else { ar += s_r[w2]; ad += s_d[w2]; }
        ac += s_c[w2];
    }
    const uint32_t xf = dpp0<kDppWaveShr1, 0xF>(f), xr = dpp0<kDppWaveShr1, 0xF>(sr),
                   xd = dpp0<kDppWaveShr1, 0xF>(sd), xc = dpp0<kDppWaveShr1, 0xF>(sc);
    const uint32_t Tr = xf ? xr : pr_ + xr, Td = xf ? xd : pd_ + xd, Tc = pc + xc;  // exclusive over the threads before
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        const uint32_t g = (uint32_t)tid * kSmallPer + i;
        if (g - g0 < kFinGroups) {
            const bool own = (lf >> i) & 1u;  // a start inside this thread's earlier items
            s_cr[g - g0] = own ? lr[i] : Tr + lr[i];
            s_cd[g - g0] = own ? ld[i] : Td + ld[i];
            s_ob[g - g0] = Tc + lc[i];
        }
    }
    if (block == 0 && tid == 255) *n_out = (uint64_t)ac;
    // ---- dense tiles (a round that overflowed the queue, a tile beyond the slab) are walked again with carry-in and
    // output base known.  They come in stretches (an SV-dense contig), so they are dealt out round-robin over ALL
    // finishing workgroups — tile t to workgroup t mod n — instead of staying with the workgroup that finishes
    // their neighbours: every workgroup has scanned every group's prefix anyway.  (Until this change a workgroup
    // walked its own sixteen tiles' dense ones, up to four per wave one after the other: 29 us for this kernel in
    // the full-size run of the command line against 5 on sparse input.)
    const uint32_t n_fin = WITH_POST ? gridDim.x - n_a3_blocks : gridDim.x;
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        uint32_t m = d[i].w & 0xFu;
        const uint32_t g = (uint32_t)tid * kSmallPer + i;
        while (m) {  // rare
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            const uint32_t t = g * kWaves + k;
            if (t % n_fin == block) {
                const bool own = (lf >> i) & 1u;
                const uint32_t at = atomicAdd(&s_n_dense, 1u);
                s_dense[at] = make_uint4(t, own ? lr[i] : Tr + lr[i], own ? ld[i] : Td + ld[i], Tc + lc[i]);
            }
        }
    }
    __syncthreads();
    // ---- inside the group (= this wave's four tiles, 16 lanes each): fold the tiles before mine
    uint32_t t_cr = s_cr[wave], t_cd = s_cd[wave], t_ob = s_ob[wave];
    {
        const int k_mine = lane / kFinLanes;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t vx = __builtin_amdgcn_readlane(dsc.x, k * kFinLanes), vy = __builtin_amdgcn_readlane(dsc.y, k * kFinLanes),
                           vz = __builtin_amdgcn_readlane(dsc.z, k * kFinLanes);
            if (k < k_mine) {
                if (vx >> 31) { t_cr = vy; t_cd = vz; }
                else { t_cr += vy; t_cd += vz; }
                t_ob += vx & 0x3FFFFFFFu;
            }
        }
    }

    // ---- finish, sparse tiles (as k_cigar_finish): 16 lanes per tile
    const uint32_t cnt = dsc.x & 0x3FFFFFFFu;
    const bool dense = cnt > (uint32_t)kSlab || (dsc.x & kDescForceDense);
    if (mine && cnt && !dense) {
        const uint32_t cr = t_cr, cd = t_cd;
        const uint64_t ob = t_ob;
        uint32_t rs[kFinSpec];
#pragma unroll
        for (int k = 0; k < kFinSpec; ++k)
            rs[k] = (l + k * kFinLanes < cnt && p.ref_start) ? (uint32_t)p.ref_start[spec[k].x] : 0u;
#pragma unroll
        for (int k = 0; k < kFinSpec; ++k) {
            const uint32_t r = l + k * kFinLanes;
            if (r < cnt && ob + r < p.cap) {
                const uint4 rec = spec[k];
                const uint32_t len = rec.w & 0x0FFFFFFFu, type = (rec.w >> 28) & 1u, prec = (rec.w >> 29) & 1u;
                p.out.aln[ob + r] = rec.x;
                p.out.ref_pos[ob + r] = rec.y + (prec ? cr : 0u) + rs[k];
                p.out.read_pos[ob + r] = rec.z + (prec ? cd : 0u);
                p.out.len[ob + r] = len;
                p.out.type[ob + r] = (uint8_t)type;
            }
        }
        for (uint32_t r = l + kFinSpec * kFinLanes; r < cnt; r += kFinLanes) {
            const uint4 rec = p.slab[(uint64_t)tile * kSlab + r];
            const uint32_t len = rec.w & 0x0FFFFFFFu, type = (rec.w >> 28) & 1u, prec = (rec.w >> 29) & 1u;
            store_final(p, ob + r, rec.x, rec.y + (prec ? cr : 0u), rec.z + (prec ? cd : 0u), len, type);
        }
    }
    // ---- this workgroup's share of the dense tiles, one wave each, the waves taking turns: fold the tiles in front of
    // it inside its group of four (scalar loads of their descriptors), then the walk
    const uint32_t n_dense = s_n_dense;
    for (uint32_t i = wave; i < n_dense; i += kWaves) {
        const uint4 e = s_dense[i];
        TileIn in;
        in.carry_r = e.y;
        in.carry_d = e.z;
        in.obase = e.w;
        const uint32_t first = e.x & ~(uint32_t)(kWaves - 1);
        for (uint32_t t = first; t < e.x; ++t) {  // wave-uniform
            const uint4 v = p.desc[t];
            if (v.x >> 31) { in.carry_r = v.y; in.carry_d = v.z; }
            else { in.carry_r += v.y; in.carry_d += v.z; }
            in.obase += v.x & 0x3FFFFFFFu;
        }
        in.a_lo = p.desc[e.x].w;
        (void)process_tile<MODE_DIRECT, SOA, kSmallTileOps, ALO_GIVEN>(p, e.x, lane, s_xpose[wave], s_head[wave],
                                                                              reinterpret_cast<uint4*>(s_head[wave]), nullptr, in);
    }
}

// The small-batch path's two launches carry the chain inside them: the first one the rows and the decision tree
// beside the tile workgroups of k_cigar_tiles, the second one (k_cigar_finish_small) the post-passes beside the
// finishing workgroups — the chain's workgroups first in each grid, they are the longer dependent sequences.  So
// a1+a2 and a3 of a sample overlap without a second stream and its cross-stream events, and the register-hungry
// post-pass code (float64 linkage) stays out of the kernel whose occupancy matters.
template <bool SOA, int TILE_OPS, int ALO>
__global__ __launch_bounds__(64 * kWaves, SVX_TILE_MIN_WAVES) void k_tiles_a3(CigarArgs p, A3Args a, uint32_t n_a3_blocks) {
    // a workgroup is either a chain workgroup or a tile workgroup: one LDS block, two layouts
    struct TileLds {
        uint4 xpose[kWaves][kXposeU4];
        __attribute__((aligned(16))) uint32_t head[kWaves][head_words<TILE_OPS>()];
        uint4 stage[kWaves][kStage];
    };
    constexpr size_t kLdsBytes = sizeof(TileLds) > sizeof(A3Lds) ? sizeof(TileLds) : sizeof(A3Lds);
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kLdsBytes];
    TileLds& tl = *reinterpret_cast<TileLds*>(s_raw);
    // the first n_a3_blocks workgroups carry the chain: it is the longer dependent sequence
    if (blockIdx.x < n_a3_blocks) {  // workgroup-uniform
        a3_chain_block<A3_ROWS_TREE>(a, blockIdx.x, reinterpret_cast<A3Lds*>(s_raw));
        return;
    }
    const uint32_t tile_block = blockIdx.x - n_a3_blocks;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t tile = tile_block * kWaves + wave;
    uint4 dsc = make_uint4(0, 0, 0, 0);
    if (tile < p.n_tiles)
        dsc = process_tile<MODE_STAGE, SOA, TILE_OPS, ALO>(p, tile, lane, tl.xpose[wave], tl.head[wave],
                                                          reinterpret_cast<uint4*>(tl.head[wave]), tl.stage[wave], TileIn());
    if (TILE_OPS == kSmallTileOps) {
        __shared__ uint4 s_agg[kWaves];
        fold_group_desc(p, s_agg, wave, lane, dsc, tile_block);
    }
}

// Streaming path: the chain's stage A rides inside the FINISH launch (the chain's workgroups first: they are the longer
// dependent sequences), its stage B inside the dense-tile launch; the bandwidth-bound streaming launch stays pure.
// (Measured and not kept, profiles/r04_ab_chain_placement.txt: stage A among the tile workgroups of the streaming launch
// — the chain workgroups then hold slots of the tile workgroups, step 0.3936 vs 0.3877 ms —; the chain's workgroups
// dealt out between the finishing ones, 62 us instead of 54, or behind them, 63.)
#ifndef SVX_FIN_A3_WAVES
#define SVX_FIN_A3_WAVES 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SVX_FIN_A3_WAVES, 8)))
void k_finish_a3(CigarArgs p, A3Args a, uint32_t n_a3_blocks) {
    if (blockIdx.x < n_a3_blocks) {  // workgroup-uniform
#ifndef SVX_EXP_FIN_ONLY
        __shared__ A3Lds lds;
        a3_chain_block<A3_ROWS_TREE>(a, blockIdx.x, &lds);
#endif
        return;
    }
#ifndef SVX_EXP_A3_ONLY
    cigar_finish_block(p, blockIdx.x - n_a3_blocks);
#endif
}

// Fills the chain's arguments from a plan; takes the tree's and the post-passes' scratch from the workspace
// (reserved by the caller together with the CIGAR path's).
void a3_fill(svx_ctx* ctx, const uint32_t* d_cigar, const uint64_t* d_aln_off, const svx_a3_plan& q, uint64_t post_stride,
             A3Args* a) {
    a->rows = SegRowArgs{d_cigar, d_aln_off, q.d_seg_src, q.d_seg_tid, q.d_seg_pos, q.d_seg_rev, q.d_seg_qend, q.n_segs,
                         q.d_read_off, q.n_reads, q.d_segs, q.d_read_len};
    a->seg_rl = svx_ws_take<int32_t>(ctx, q.n_segs ? q.n_segs : 1);
    a->tree.segs = q.d_segs;
    a->tree.seg_rl = a->seg_rl;
    a->tree.read_len_out = q.d_read_len;
    a->tree.sorted = svx_ws_take<svx_seg>(ctx, q.n_segs ? q.n_segs : 1);
    a->tree.read_off = q.d_read_off;
    a->tree.read_len = q.d_read_len;
    a->tree.n_reads = q.n_reads;
    a->tree.o = q.params;
    a->tree.out = q.d_raw;
    svx_post_dev::PostArgs& o = a->post;
    o.raw = q.d_raw; o.read_off = q.d_read_off; o.n_reads = q.n_reads; o.contig_rank = q.d_contig_rank; o.n_contigs = q.n_contigs;
    o.min_sv = q.params.min_sv_size; o.max_sv = q.params.max_sv_size;
    o.out = q.d_post; o.out_off = q.d_post_off; o.out_cnt = q.d_post_cnt;
    o.scratch = svx_ws_take<char>(ctx, (size_t)q.n_reads * post_stride);
    o.scratch_off = nullptr;
    o.scratch_stride = post_stride;
    // reads per workgroup of the rows + tree stage when the caller hands over no table: one or two while the grid stays
    // small (the shortest dependent sequence per workgroup), up to 32 for a cohort's reads, so that its ~2 000
    // workgroups are all resident at once (8 per CU).  Measured on the cohort of 256 samples with the round-4 rows loop,
    // k_finish_a3: 8 reads 67.7 us, 16: 55.3, 24: 56.4, 32: 53.5, 48: 57.1, 64: 59.1 (profiles/r04_ab_chain_rows.txt);
    // svx_chain_deal (svx_collect.hip) uses the same count of workgroups for its table
#ifndef SVX_A3_READS_DIV
#define SVX_A3_READS_DIV 1024u
#endif
#ifndef SVX_A3_READS_MAX
#define SVX_A3_READS_MAX 32u
#endif
    const uint32_t per = q.n_reads / SVX_A3_READS_DIV;
    // (one read per workgroup while that leaves at most a workgroup and a half per CU: one sample 13.1 -> 11.5 us per
    //  step; the 533 reads of a diploid submission are better off with two: 25.9 vs 26.7 us)
    const uint32_t least = q.n_reads < 384u ? 1u : 2u;
    a->reads_per_block = per < least ? least : (per > SVX_A3_READS_MAX ? SVX_A3_READS_MAX : per);
    a->deal = q.n_deal_blocks ? q.d_deal : nullptr;
    a->n_deal = q.n_deal_blocks;
}

// a3 != nullptr: the split-segment chain of the same submission goes out with the CIGAR path — inside the tile and
// finish launches of the small-batch path, inside the finish and dense-tile launches of the streaming path
// (svx_collect_batch_dev).
template <bool SOA>
int cigar_extract_dev_impl(svx_ctx* ctx, const uint32_t* d_cigar_or_len, const uint8_t* d_op,
                           uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                           const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out,
                           uint64_t cap, uint64_t* d_n_out, const svx_a3_plan* a3 = nullptr) {
    if (!ctx || !d_n_out) return SVX_E_INVALID;
    if (n_ops >= (1ull << 32)) {
        SVX_SET_ERR(ctx, "n_ops=%llu exceeds the 2^32-1 per-call limit; split the batch",
                    (unsigned long long)n_ops);
        return SVX_E_TOO_LARGE;
    }
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    if (n_ops == 0 || n_aln == 0) {
        SVX_HIP(ctx, hipMemsetAsync(d_n_out, 0, sizeof(uint64_t), ctx->stream));
        return a3 ? SVX_E_INVALID : SVX_OK;
    }
    if (!d_cigar_or_len || !d_aln_off || (SOA && !d_op)) return SVX_E_INVALID;
    if (cap > 0 && (!d_out.aln || !d_out.ref_pos || !d_out.read_pos || !d_out.len || !d_out.type))
        return SVX_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(d_cigar_or_len) & 15u) ||
        (SOA && (reinterpret_cast<uintptr_t>(d_op) & 15u))) {
        SVX_SET_ERR(ctx, "device CIGAR buffers must be 16-byte aligned");
        return SVX_E_INVALID;
    }
    const bool small = n_ops <= (uint64_t)kSmallMaxTiles * kSmallTileOps && n_ops <= ctx->small_batch_ops;
    const uint32_t n_tiles = small ? (uint32_t)((n_ops + kSmallTileOps - 1) / kSmallTileOps)
                                   : (uint32_t)((n_ops + kTileOps - 1) / kTileOps);
    size_t need = svx_cigar_extract_ws_need(ctx, n_ops);
    if (a3) need += svx_take_bytes(a3->n_segs ? a3->n_segs : 1, sizeof(svx_seg)) + svx_take_bytes(a3->n_segs ? a3->n_segs : 1, 4) +
                    svx_take_bytes((size_t)a3->n_reads * a3->post_stride, 1);
    int rc = svx_ws_reserve(ctx, need);
    if (rc != SVX_OK) return rc;

    CigarArgs a;
    a.cigar = d_cigar_or_len;
    a.op = d_op;
    a.aln_off = d_aln_off;
    a.ref_start = d_ref_start;
    a.n_ops = n_ops;
    a.n_aln = n_aln;
    a.n_tiles = n_tiles;
    a.min_len = min_len;
    a.n_dense = reinterpret_cast<uint32_t*>(ctx->ws);  // workspace header: zero at allocation, self-cleaning
    a.desc = svx_ws_take<uint4>(ctx, n_tiles);
    a.desc4 = svx_ws_take<uint4>(ctx, (n_tiles + kWaves - 1) / kWaves);
    a.slab = svx_ws_take<uint4>(ctx, (size_t)n_tiles * kSlab);
    a.out_base = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.carry_ref = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.carry_read = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.dense_list = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.tile_alo = svx_ws_take<uint32_t>(ctx, n_tiles);
    const uint32_t n_scan_blocks = (n_tiles + kScanBlock - 1) / kScanBlock;
    a.blk_agg = svx_ws_take<uint4>(ctx, n_scan_blocks);
    a.blk_prefix = svx_ws_take<uint4>(ctx, n_scan_blocks);
    a.out = d_out;
    a.cap = cap;
    A3Args c;
    memset(&c, 0, sizeof(c));
    uint32_t n_a3_blocks = 0, n_post_blocks = 0;
    if (a3) {
        a3_fill(ctx, d_cigar_or_len, d_aln_off, *a3, a3->post_stride, &c);
        n_a3_blocks = c.deal ? c.n_deal : (a3->n_reads + c.reads_per_block - 1) / c.reads_per_block;
        n_post_blocks = (a3->n_reads + 255u) / 256u;
    }

    if (small) {  // two launches: tiles of 1024 ops (+ the split-segment chain); scan + finish + dense tiles
        rc = svx_timing_begin(ctx);
        if (rc != SVX_OK) return rc;
        rc = svx_timing_mark(ctx, 1);
        if (rc != SVX_OK) return rc;
        const uint32_t tile_blocks = (n_tiles + kWaves - 1) / kWaves;
        if (a3)
            hipLaunchKernelGGL((k_tiles_a3<SOA, kSmallTileOps, ALO_SEARCH>), dim3(n_a3_blocks + tile_blocks), dim3(64 * kWaves), 0,
                               ctx->stream, a, c, n_a3_blocks);
        else
            hipLaunchKernelGGL((k_cigar_tiles<SOA, kSmallTileOps, ALO_SEARCH>), dim3(tile_blocks), dim3(64 * kWaves), 0,
                               ctx->stream, a);
        rc = svx_timing_mark(ctx, 2);
        if (rc != SVX_OK) return rc;
        if (ctx->want_dom) {
            if (!ctx->ev_dom) SVX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_dom, hipEventDisableTiming));
            SVX_HIP(ctx, hipEventRecord(ctx->ev_dom, ctx->stream));
            ctx->ev_dom_recorded = true;
        }
        // (at least SVX_FIN_MIN_BLOCKS workgroups: the ones beyond the tiles' own have no records to copy and only take
        //  their share of the dense tiles — a batch of a few hundred tiles that are ALL dense is a satellite array)
#ifndef SVX_FIN_MIN_BLOCKS
#define SVX_FIN_MIN_BLOCKS 128u  // (tools/dense_probe.py `satellite`, two-launch path: 48.4 -> 28.2 us; 256: 28.0)
#endif
        uint32_t fin_blocks = (n_tiles + kFinTiles - 1) / kFinTiles;
        if (fin_blocks < SVX_FIN_MIN_BLOCKS) fin_blocks = SVX_FIN_MIN_BLOCKS;
        if (a3)
            hipLaunchKernelGGL((k_cigar_finish_small<SOA, true>), dim3(n_post_blocks + fin_blocks), dim3(256), 0, ctx->stream, a, d_n_out,
                               c, n_post_blocks);
        else
            hipLaunchKernelGGL((k_cigar_finish_small<SOA, false>), dim3(fin_blocks), dim3(256), 0, ctx->stream, a, d_n_out, c, 0u);
        SVX_HIP(ctx, hipGetLastError());
        return svx_timing_end(ctx);
    }
    const uint32_t blocks_all = (n_tiles + kWaves - 1) / kWaves;
    const uint32_t blocks_cap = (uint32_t)ctx->n_cu * 8u;
    const uint32_t finish_blocks = (n_tiles + 256 / kFinLanes - 1) / (256 / kFinLanes);
    const uint32_t dense_blocks = n_tiles < blocks_cap ? n_tiles : blocks_cap;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_tile_alo, dim3((n_aln + 255) / 256), dim3(256), 0, ctx->stream, d_aln_off, n_aln, n_tiles,
                       a.tile_alo);
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL((k_cigar_tiles<SOA, kTileOps, ALO_TABLE>), dim3(blocks_all), dim3(64 * kWaves), 0, ctx->stream, a);
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    if (ctx->want_dom) {  // somebody pipelines against this context (svx_ctx_wait_dominant)
        if (!ctx->ev_dom) SVX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_dom, hipEventDisableTiming));
        SVX_HIP(ctx, hipEventRecord(ctx->ev_dom, ctx->stream));
        ctx->ev_dom_recorded = true;
    }
    hipLaunchKernelGGL(k_desc_scan, dim3(n_scan_blocks), dim3(kScanBlock), 0, ctx->stream, a.desc, n_tiles,
                       a.out_base, a.carry_ref, a.carry_read, a.dense_list, a.n_dense, a.blk_agg,
                       a.blk_prefix, a.n_dense + 1, d_n_out);
    if (a3) {  // the chain's rows and decision tree inside the finish launch, its post-passes inside the dense-tile launch
        hipLaunchKernelGGL(k_finish_a3, dim3(n_a3_blocks + finish_blocks), dim3(256), 0, ctx->stream, a, c, n_a3_blocks);
        hipLaunchKernelGGL((k_cigar_dense<SOA, true>), dim3(n_post_blocks + dense_blocks), dim3(64 * kWaves), 0, ctx->stream, a, c,
                           n_post_blocks);
    } else {
        hipLaunchKernelGGL(k_cigar_finish, dim3(finish_blocks), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL((k_cigar_dense<SOA, false>), dim3(dense_blocks), dim3(64 * kWaves), 0, ctx->stream, a, c, 0u);
    }
    SVX_HIP(ctx, hipGetLastError());
    return svx_timing_end(ctx);
}

int validate_offsets(svx_ctx* ctx, const uint64_t* aln_off, uint32_t n_aln) {
    if (n_aln == 0) return SVX_OK;
    if (!aln_off) return SVX_E_INVALID;
    if (aln_off[0] != 0) {
        SVX_SET_ERR(ctx, "aln_off[0] must be 0");
        return SVX_E_INVALID;
    }
    for (uint32_t i = 0; i < n_aln; ++i)
        if (aln_off[i + 1] < aln_off[i]) {
            SVX_SET_ERR(ctx, "aln_off must be non-decreasing (index %u)", i);
            return SVX_E_INVALID;
        }
    return SVX_OK;
}

template <bool SOA>
int cigar_extract_host_impl(svx_ctx* ctx, const uint32_t* cigar_or_len, const uint8_t* op,
                            const uint64_t* aln_off, uint32_t n_aln, const int32_t* ref_start,
                            uint32_t min_len, svx_sig_soa out, uint64_t cap, uint64_t* n_out) {
    if (!ctx || !n_out) return SVX_E_INVALID;
    *n_out = 0;
    int rc = validate_offsets(ctx, aln_off, n_aln);
    if (rc != SVX_OK) return rc;
    const uint64_t n_ops = n_aln ? aln_off[n_aln] : 0;
    if (n_ops == 0) return SVX_OK;
    if (!cigar_or_len || (SOA && !op)) return SVX_E_INVALID;
    if (n_ops >= (1ull << 32)) return SVX_E_TOO_LARGE;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n_ops, 4) + (SOA ? svx_take_bytes(n_ops, 1) : 0) +
                  svx_take_bytes((size_t)n_aln + 1, 8) + svx_take_bytes(n_aln, 4) +
                  4 * svx_take_bytes(cap, 4) + svx_take_bytes(cap, 1) + svx_take_bytes(1, 8);
    rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    uint32_t* d_c = svx_stage_take<uint32_t>(ctx, n_ops);
    uint8_t* d_op = SOA ? svx_stage_take<uint8_t>(ctx, n_ops) : nullptr;
    uint64_t* d_off = svx_stage_take<uint64_t>(ctx, (size_t)n_aln + 1);
    int32_t* d_rs = ref_start ? svx_stage_take<int32_t>(ctx, n_aln) : nullptr;
    svx_sig_soa d_out;
    d_out.aln = svx_stage_take<uint32_t>(ctx, cap);
    d_out.ref_pos = svx_stage_take<uint32_t>(ctx, cap);
    d_out.read_pos = svx_stage_take<uint32_t>(ctx, cap);
    d_out.len = svx_stage_take<uint32_t>(ctx, cap);
    d_out.type = svx_stage_take<uint8_t>(ctx, cap);
    uint64_t* d_n = svx_stage_take<uint64_t>(ctx, 1);
    SVX_HIP(ctx, hipMemcpyAsync(d_c, cigar_or_len, n_ops * 4, hipMemcpyHostToDevice, ctx->stream));
    if (SOA) SVX_HIP(ctx, hipMemcpyAsync(d_op, op, n_ops, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_off, aln_off, ((size_t)n_aln + 1) * 8, hipMemcpyHostToDevice,
                                ctx->stream));
    if (d_rs)
        SVX_HIP(ctx, hipMemcpyAsync(d_rs, ref_start, (size_t)n_aln * 4, hipMemcpyHostToDevice,
                                    ctx->stream));
    rc = cigar_extract_dev_impl<SOA>(ctx, d_c, d_op, n_ops, d_off, n_aln, d_rs, min_len, d_out, cap,
                                     d_n);
    if (rc != SVX_OK) return rc;
    uint64_t n = 0;
    SVX_HIP(ctx, hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    const uint64_t m = n < cap ? n : cap;
    if (m) {
        SVX_HIP(ctx, hipMemcpyAsync(out.aln, d_out.aln, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipMemcpyAsync(out.ref_pos, d_out.ref_pos, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipMemcpyAsync(out.read_pos, d_out.read_pos, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipMemcpyAsync(out.len, d_out.len, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipMemcpyAsync(out.type, d_out.type, m, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n > cap) {
        SVX_SET_ERR(ctx, "output capacity %llu < %llu signatures", (unsigned long long)cap,
                    (unsigned long long)n);
        return SVX_E_CAPACITY;
    }
    return SVX_OK;
}

}  // namespace

extern "C" int svx_cigar_extract_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops,
                                     const uint64_t* d_aln_off, uint32_t n_aln,
                                     const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out,
                                     uint64_t cap, uint64_t* d_n_out) {
    return cigar_extract_dev_impl<false>(ctx, d_cigar, nullptr, n_ops, d_aln_off, n_aln, d_ref_start,
                                         min_len, d_out, cap, d_n_out);
}

int svx_cigar_extract_chain_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                                const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out, uint64_t cap,
                                uint64_t* d_n_out, const svx_a3_plan* a3) {
    return cigar_extract_dev_impl<false>(ctx, d_cigar, nullptr, n_ops, d_aln_off, n_aln, d_ref_start, min_len, d_out, cap,
                                         d_n_out, a3);
}

extern "C" int svx_cigar_extract_soa_dev(svx_ctx* ctx, const uint8_t* d_op, const uint32_t* d_len,
                                         uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                                         const int32_t* d_ref_start, uint32_t min_len,
                                         svx_sig_soa d_out, uint64_t cap, uint64_t* d_n_out) {
    return cigar_extract_dev_impl<true>(ctx, d_len, d_op, n_ops, d_aln_off, n_aln, d_ref_start,
                                        min_len, d_out, cap, d_n_out);
}

extern "C" int svx_cigar_extract(svx_ctx* ctx, const uint32_t* cigar, const uint64_t* aln_off,
                                 uint32_t n_aln, const int32_t* ref_start, uint32_t min_len,
                                 svx_sig_soa out, uint64_t cap, uint64_t* n_out) {
    return cigar_extract_host_impl<false>(ctx, cigar, nullptr, aln_off, n_aln, ref_start, min_len,
                                          out, cap, n_out);
}

extern "C" int svx_cigar_extract_soa(svx_ctx* ctx, const uint8_t* op, const uint32_t* len,
                                     const uint64_t* aln_off, uint32_t n_aln,
                                     const int32_t* ref_start, uint32_t min_len, svx_sig_soa out,
                                     uint64_t cap, uint64_t* n_out) {
    return cigar_extract_host_impl<true>(ctx, len, op, aln_off, n_aln, ref_start, min_len, out, cap,
                                         n_out);
}

extern "C" int svx_cigar_stats_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops,
                                   const uint64_t* d_aln_off, uint32_t n_aln, svx_aln_stats d_out) {
    if (!ctx) return SVX_E_INVALID;
    if (n_aln == 0) return SVX_OK;
    if (!d_aln_off || (n_ops && !d_cigar)) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    StatsArgs a{d_cigar, d_aln_off, n_aln, d_out};
    uint32_t blocks = (n_aln + 3) / 4;
    uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    hipLaunchKernelGGL(k_cigar_stats, dim3(blocks < cap ? blocks : cap), dim3(256), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    return SVX_OK;
}

extern "C" int svx_cigar_stats(svx_ctx* ctx, const uint32_t* cigar, const uint64_t* aln_off,
                               uint32_t n_aln, svx_aln_stats out) {
    if (!ctx) return SVX_E_INVALID;
    int rc = validate_offsets(ctx, aln_off, n_aln);
    if (rc != SVX_OK) return rc;
    if (n_aln == 0) return SVX_OK;
    const uint64_t n_ops = aln_off[n_aln];
    if (n_ops && !cigar) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n_ops, 4) + svx_take_bytes((size_t)n_aln + 1, 8) +
                  5 * svx_take_bytes(n_aln, 4);
    rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    uint32_t* d_c = svx_stage_take<uint32_t>(ctx, n_ops ? n_ops : 1);
    uint64_t* d_off = svx_stage_take<uint64_t>(ctx, (size_t)n_aln + 1);
    svx_aln_stats d;
    d.ref_len = svx_stage_take<uint32_t>(ctx, n_aln);
    d.q_start = svx_stage_take<uint32_t>(ctx, n_aln);
    d.q_end = svx_stage_take<uint32_t>(ctx, n_aln);
    d.read_len = svx_stage_take<uint32_t>(ctx, n_aln);
    d.n_hard = svx_stage_take<uint32_t>(ctx, n_aln);
    if (n_ops) SVX_HIP(ctx, hipMemcpyAsync(d_c, cigar, n_ops * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_off, aln_off, ((size_t)n_aln + 1) * 8, hipMemcpyHostToDevice,
                                ctx->stream));
    rc = svx_cigar_stats_dev(ctx, d_c, n_ops, d_off, n_aln, d);
    if (rc != SVX_OK) return rc;
    const size_t b = (size_t)n_aln * 4;
    if (out.ref_len) SVX_HIP(ctx, hipMemcpyAsync(out.ref_len, d.ref_len, b, hipMemcpyDeviceToHost, ctx->stream));
    if (out.q_start) SVX_HIP(ctx, hipMemcpyAsync(out.q_start, d.q_start, b, hipMemcpyDeviceToHost, ctx->stream));
    if (out.q_end) SVX_HIP(ctx, hipMemcpyAsync(out.q_end, d.q_end, b, hipMemcpyDeviceToHost, ctx->stream));
    if (out.read_len) SVX_HIP(ctx, hipMemcpyAsync(out.read_len, d.read_len, b, hipMemcpyDeviceToHost, ctx->stream));
    if (out.n_hard) SVX_HIP(ctx, hipMemcpyAsync(out.n_hard, d.n_hard, b, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

size_t svx_cigar_extract_ws_need(const svx_ctx* ctx, uint64_t n_ops) {
    const bool small = n_ops <= (uint64_t)kSmallMaxTiles * kSmallTileOps && n_ops <= ctx->small_batch_ops;
    const uint32_t n_tiles = small ? (uint32_t)((n_ops + kSmallTileOps - 1) / kSmallTileOps)
                                   : (uint32_t)((n_ops + kTileOps - 1) / kTileOps);
    return svx_take_bytes(n_tiles, sizeof(uint4)) + svx_take_bytes((n_tiles + kWaves - 1) / kWaves, sizeof(uint4)) +
           svx_take_bytes((size_t)n_tiles * kSlab, sizeof(uint4)) +
           5 * svx_take_bytes(n_tiles, sizeof(uint32_t)) + svx_take_bytes(4, sizeof(uint32_t)) +
           2 * svx_take_bytes((n_tiles + kScanBlock - 1) / kScanBlock, sizeof(uint4));
}

extern "C" int svx_segments_rows_dev(svx_ctx* ctx, const uint32_t* d_cigar, const uint64_t* d_aln_off,
                                     const uint32_t* d_seg_src, const int32_t* d_seg_tid, const int32_t* d_seg_pos,
                                     const uint8_t* d_seg_rev, const int32_t* d_seg_qend, uint32_t n_segs,
                                     const uint32_t* d_read_off, uint32_t n_reads, svx_seg* d_segs, int32_t* d_read_len) {
    if (!ctx) return SVX_E_INVALID;
    if (n_segs == 0 && n_reads == 0) return SVX_OK;
    if (!d_aln_off || !d_seg_src || !d_seg_tid || !d_seg_pos || !d_seg_rev || !d_seg_qend || !d_read_off || !d_segs ||
        !d_read_len)
        return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    SegRowArgs a{d_cigar, d_aln_off, d_seg_src, d_seg_tid, d_seg_pos, d_seg_rev, d_seg_qend, n_segs, d_read_off, n_reads,
                 d_segs, d_read_len};
    const uint32_t work = n_segs > n_reads ? n_segs : n_reads;
    uint32_t blocks = (work + 3) / 4;
    const uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    hipLaunchKernelGGL(k_segment_rows, dim3(blocks < cap ? blocks : cap), dim3(256), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    return SVX_OK;
}

#ifdef SVX_EXP_PROF
extern "C" int svx_debug_prof(uint32_t* out, uint32_t n_tiles) {
    if (n_tiles > kProfTiles) n_tiles = kProfTiles;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(uint32_t) * 8 * n_tiles) == hipSuccess ? 0 : -1;
}
#endif
