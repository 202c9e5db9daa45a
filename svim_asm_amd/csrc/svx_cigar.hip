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
// Pipeline (no inter-workgroup waiting anywhere, so no dispatch-order assumption):
//   A  k_cigar_tiles<STAGE>   one WAVE per tile of 4096 ops; 4 rounds of 1024 ops; per round:
//                             coalesced dwordx4 loads (1 KiB/wave-instr) → wave-private padded LDS
//                             transpose → 16 consecutive ops per lane → lane-local segmented walk →
//                             wave segmented scan (__shfl_up) → signatures staged into the tile's
//                             slab (256 x 16 B) with tile-local cursors; tile descriptor
//                             {count, seen_head, ref_tail, read_tail, a_lo} written at the end.
//   B  k_desc_scan            segmented exclusive scan over the tile descriptors: per-tile carry-in
//                             (cursor sums since the last alignment start before the tile), output
//                             base (exclusive signature count) and the list of dense tiles.
//   C  k_cigar_gather         one wave per sparse tile: slab → final SoA at out_base, adding the
//                             carry to signatures that precede the tile's first alignment start
//                             and resolving the alignment index by bounded search in aln_off.
//   A2 k_cigar_tiles<DIRECT>  dense tiles (> 256 signatures, e.g. adversarial all-indel CIGARs) are
//                             re-walked with carry-in and output base known, writing final SoA.
// Output order = (alignment, op) order by construction (prefix sums, no atomically-ordered appends).
#include "svx_internal.h"

namespace {

constexpr int kLaneOps = 16;
constexpr int kRoundOps = 64 * kLaneOps;       // 1024 ops per wave round (4 KiB)
constexpr int kRounds = 4;
constexpr int kTileOps = kRoundOps * kRounds;  // 4096 ops per tile (16 KiB)
constexpr int kSlab = 256;                     // staged signatures per tile
constexpr int kWaves = 4;                      // waves (= tiles) per workgroup
constexpr int kXposeU4 = 64 * 5;               // padded transpose buffer: 5 uint4 per lane

enum { MODE_STAGE = 0, MODE_DIRECT = 1 };

struct CigarArgs {
    const uint32_t* cigar;  // packed words, or len[] in SoA mode
    const uint8_t* op;      // SoA mode only
    const uint64_t* aln_off;
    const int32_t* ref_start;  // nullable
    uint64_t n_ops;
    uint32_t n_aln;
    uint32_t n_tiles;
    uint32_t min_len;
    uint4* desc;
    uint4* slab;
    uint32_t* out_base;
    uint32_t* carry_ref;
    uint32_t* carry_read;
    uint32_t* dense_list;
    uint32_t* n_dense;
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
    while (hi - lo > 64) {
        uint32_t step = (hi - lo + 63) / 64;
        uint64_t idx = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
        bool less = (idx < hi) ? (a[idx] < key) : false;
        uint32_t p = __popcll(__ballot(less));
        uint64_t nlo = (uint64_t)lo + (uint64_t)p * step;
        uint64_t nhi = (uint64_t)lo + (uint64_t)(p + 1) * step - 1;
        lo = (uint32_t)(nlo < hi ? nlo : hi);
        hi = (uint32_t)(nhi < hi ? nhi : hi);
    }
    uint32_t idx = lo + lane;
    bool less = (idx < hi) ? (a[idx] < key) : false;
    return lo + __popcll(__ballot(less));
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
        p.out.aln[slot] = aln;
        p.out.ref_pos[slot] = ref + rs;
        p.out.read_pos[slot] = read;
        p.out.len[slot] = len;
        p.out.type[slot] = (uint8_t)type;
    }
}

template <int MODE, bool SOA>
__global__ __launch_bounds__(64 * kWaves) void k_cigar_tiles(CigarArgs p) {
    __shared__ uint4 s_xpose[kWaves][kXposeU4];
    __shared__ uint32_t s_head[kWaves][kTileOps / 32];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint4* xp = s_xpose[wave];
    uint32_t* hmask = s_head[wave];

    uint32_t work = blockIdx.x * kWaves + wave;
    const uint32_t work_stride = gridDim.x * kWaves;
    const uint32_t n_work = (MODE == MODE_DIRECT) ? *p.n_dense : p.n_tiles;

    for (; work < n_work; work += work_stride) {
        const uint32_t tile = (MODE == MODE_DIRECT) ? p.dense_list[work] : work;
        const uint64_t g0 = (uint64_t)tile * kTileOps;
        const uint64_t tile_end = (g0 + kTileOps < p.n_ops) ? g0 + kTileOps : p.n_ops;

        // ---- alignment starts inside this tile → 4096-bit mask in LDS ----
        const uint32_t a_lo = wave_lower_bound(p.aln_off, p.n_aln + 1, g0, lane);
        hmask[lane] = 0;
        hmask[lane + 64] = 0;
        wave_lds_sync();
        for (uint64_t a = (uint64_t)a_lo + lane;; a += 64) {
            bool in = false;
            if (a < p.n_aln) {
                uint64_t off = p.aln_off[a];
                if (off < tile_end) {
                    in = true;
                    uint32_t bit = (uint32_t)(off - g0);
                    atomicOr(&hmask[bit >> 5], 1u << (bit & 31));
                }
            }
            if (!__all(in)) break;
        }
        wave_lds_sync();

        // ---- tile state carried across rounds (wave-uniform) ----
        uint32_t carry_r = 0, carry_d = 0;
        bool seen = false;
        uint32_t tile_cnt = 0;
        uint32_t obase = 0;
        if (MODE == MODE_DIRECT) {
            carry_r = p.carry_ref[tile];
            carry_d = p.carry_read[tile];
            seen = true;  // carry-in already holds "since the last start before the tile"
            obase = p.out_base[tile];
        }

        for (int round = 0; round < kRounds; ++round) {
            const uint64_t r0 = g0 + (uint64_t)round * kRoundOps;
            if (r0 >= tile_end) break;  // wave-uniform

            // ---- coalesced load: lane reads uint4 #(k*64+lane) of this round ----
            uint4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t e0 = r0 + (uint64_t)(k * 64 + lane) * 4;
                if (e0 + 4 <= tile_end) {
                    q[k] = *reinterpret_cast<const uint4*>(p.cigar + e0);
                } else {
                    uint32_t t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = (e0 + j < tile_end) ? p.cigar[e0 + j] : 0u;
                    q[k] = make_uint4(t[0], t[1], t[2], t[3]);
                }
            }
            uint32_t opw[4] = {0, 0, 0, 0};  // SoA: 16 op codes of this lane's 16 consecutive ops
            if (SOA) {
                const uint64_t e0 = r0 + (uint64_t)lane * 16;
                if (e0 + 16 <= tile_end) {
                    uint4 o = *reinterpret_cast<const uint4*>(p.op + e0);
                    opw[0] = o.x; opw[1] = o.y; opw[2] = o.z; opw[3] = o.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        uint32_t b = (e0 + j < tile_end) ? p.op[e0 + j] : 0u;
                        opw[j >> 2] |= b << ((j & 3) * 8);
                    }
                }
            }
            // ---- transpose through wave-private LDS: 5-uint4 stride per lane is conflict-free ----
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = k * 64 + lane;
                xp[(i >> 2) * 5 + (i & 3)] = q[k];
            }
            wave_lds_sync();
            uint32_t w[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint4 v = xp[lane * 5 + j];
                w[4 * j + 0] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
            }
            const uint32_t lbase = round * kRoundOps + lane * kLaneOps;  // tile-local index of op 0
            const uint32_t hm = (hmask[lbase >> 5] >> (lbase & 31)) & 0xFFFFu;
            wave_lds_sync();  // xp is rewritten next round

            // ---- lane-local segmented walk over 16 consecutive ops ----
            uint32_t rr = 0, rd = 0, emask = 0;
            uint32_t pr[16], pd[16], ln[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t op, len;
                if (SOA) {
                    op = (opw[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
                    len = w[i];
                } else {
                    op = w[i] & 15u;
                    len = w[i] >> 4;
                }
                ln[i] = len;
                if ((hm >> i) & 1u) { rr = 0; rd = 0; }
                pr[i] = rr;
                pd[i] = rd;
                // ops advancing the reference cursor: M(0) D(2) =(7) X(8); the query cursor:
                // M(0) I(1) S(4) =(7) X(8)   (SVIM_intra.py:14-29; N,H,P,B: nothing)
                const bool valid = SOA ? (op < 16u) : true;
                const uint32_t aref = (valid && ((0x185u >> op) & 1u)) ? len : 0u;
                const uint32_t ard = (valid && ((0x193u >> op) & 1u)) ? len : 0u;
                rr += aref;
                rd += ard;
                const bool em = (op - 1u) < 2u && len >= p.min_len;  // I or D, inclusive threshold
                emask |= (em ? 1u : 0u) << i;
                // keep the type in ln's spare top bit? no: len may use 28 bits; type re-derived below
                w[i] = op;
            }
            const uint32_t cnt = __popc(emask);

            // ---- wave segmented inclusive scan of (flag, ref, read) + plain scan of cnt ----
            uint32_t f = hm != 0 ? 1u : 0u, sr = rr, sd = rd, sc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t f2 = __shfl_up(f, d);
                uint32_t r2 = __shfl_up(sr, d);
                uint32_t d2 = __shfl_up(sd, d);
                uint32_t c2 = __shfl_up(sc, d);
                if (lane >= d) {
                    if (!f) { sr += r2; sd += d2; }
                    f |= f2;
                    sc += c2;
                }
            }
            // exclusive values for this lane
            uint32_t xf = __shfl_up(f, 1), xr = __shfl_up(sr, 1), xd = __shfl_up(sd, 1),
                     xc = __shfl_up(sc, 1);
            if (lane == 0) { xf = 0; xr = 0; xd = 0; xc = 0; }
            uint32_t in_r, in_d;
            bool pre;
            if (xf) {
                in_r = xr; in_d = xd; pre = false;
            } else {
                in_r = xr + carry_r; in_d = xd + carry_d; pre = !seen;
            }

            // ---- emit ----
            if (__any(emask != 0)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if ((emask >> i) & 1u) {
                        const bool own = (hm & ((2u << i) - 1u)) != 0;  // a start at or before op i in-lane
                        const uint32_t ref = own ? pr[i] : pr[i] + in_r;
                        const uint32_t rdp = own ? pd[i] : pd[i] + in_d;
                        const bool prec = own ? false : pre;
                        const uint32_t rank = tile_cnt + xc + __popc(emask & ((1u << i) - 1u));
                        const uint32_t type = (w[i] == 2u) ? SVX_SIG_DEL : SVX_SIG_INS;
                        if (MODE == MODE_STAGE) {
                            if (rank < kSlab) {
                                uint32_t w0 = (lbase + i) | (type << 12) | ((prec ? 1u : 0u) << 13);
                                p.slab[(uint64_t)tile * kSlab + rank] = make_uint4(w0, ref, rdp, ln[i]);
                            }
                        } else {
                            const uint64_t g = g0 + lbase + i;
                            const uint32_t aln = find_aln(p.aln_off, p.n_aln, a_lo, g);
                            store_final(p, (uint64_t)obase + rank, aln, ref, rdp, ln[i], type);
                        }
                    }
                }
            }

            // ---- carry to the next round (wave-uniform via lane 63's inclusive values) ----
            const uint32_t F = __shfl(f, 63), R = __shfl(sr, 63), D = __shfl(sd, 63),
                           C = __shfl(sc, 63);
            if (F) { carry_r = R; carry_d = D; seen = true; }
            else { carry_r += R; carry_d += D; }
            tile_cnt += C;
        }

        if (MODE == MODE_STAGE && lane == 0) {
            p.desc[tile] = make_uint4(tile_cnt | ((seen ? 1u : 0u) << 31), carry_r, carry_d, a_lo);
        }
    }
}

// ---- B: segmented exclusive scan over tile descriptors (single workgroup, chunked) ----
__global__ __launch_bounds__(1024) void k_desc_scan(const uint4* __restrict__ desc, uint32_t n_tiles,
                                                    uint32_t* __restrict__ out_base,
                                                    uint32_t* __restrict__ carry_ref,
                                                    uint32_t* __restrict__ carry_read,
                                                    uint32_t* __restrict__ dense_list,
                                                    uint32_t* __restrict__ n_dense,
                                                    uint64_t* __restrict__ n_out) {
    __shared__ uint32_t s_f[16], s_r[16], s_d[16], s_c[16];
    __shared__ uint32_t s_ndense;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ndense = 0;
    uint32_t cf = 0, cr = 0, cd = 0;  // running carry (uniform)
    uint64_t cc = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t t = base + tid;
        uint32_t f = 0, r = 0, d = 0, c = 0;
        if (t < n_tiles) {
            uint4 v = desc[t];
            c = v.x & 0x7FFFFFFFu;
            f = v.x >> 31;
            r = v.y;
            d = v.z;
            if (c > (uint32_t)kSlab) dense_list[atomicAdd(&s_ndense, 1u)] = t;
        }
        uint32_t sf = f, sr = r, sd = d, sc = c;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            uint32_t f2 = __shfl_up(sf, k), r2 = __shfl_up(sr, k), d2 = __shfl_up(sd, k),
                     c2 = __shfl_up(sc, k);
            if (lane >= k) {
                if (!sf) { sr += r2; sd += d2; }
                sf |= f2;
                sc += c2;
            }
        }
        if (lane == 63) { s_f[wave] = sf; s_r[wave] = sr; s_d[wave] = sd; s_c[wave] = sc; }
        __syncthreads();
        // prefix over preceding waves, seeded with the running carry
        uint32_t pf = cf, pr_ = cr, pd_ = cd;
        uint64_t pc = cc;
        for (int w2 = 0; w2 < wave; ++w2) {
            if (s_f[w2]) { pf = 1; pr_ = s_r[w2]; pd_ = s_d[w2]; }
            else { pr_ += s_r[w2]; pd_ += s_d[w2]; }
            pc += s_c[w2];
        }
        // exclusive within the wave
        uint32_t xf = __shfl_up(sf, 1), xr = __shfl_up(sr, 1), xd = __shfl_up(sd, 1),
                 xc = __shfl_up(sc, 1);
        if (lane == 0) { xf = 0; xr = 0; xd = 0; xc = 0; }
        uint32_t er = xf ? xr : pr_ + xr;
        uint32_t ed = xf ? xd : pd_ + xd;
        if (t < n_tiles) {
            carry_ref[t] = er;
            carry_read[t] = ed;
            out_base[t] = (uint32_t)(pc + xc);
        }
        // new running carry = prefix through the last wave
        uint32_t nf = cf, nr = cr, nd = cd;
        uint64_t nc = cc;
        for (int w2 = 0; w2 < 16; ++w2) {
            if (s_f[w2]) { nf = 1; nr = s_r[w2]; nd = s_d[w2]; }
            else { nr += s_r[w2]; nd += s_d[w2]; }
            nc += s_c[w2];
        }
        cf = nf; cr = nr; cd = nd; cc = nc;
        __syncthreads();
    }
    if (tid == 0) {
        *n_dense = s_ndense;
        *n_out = cc;
    }
}

// ---- C: gather sparse tiles' staged signatures into the final SoA ----
__global__ __launch_bounds__(64 * kWaves) void k_cigar_gather(CigarArgs p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < p.n_tiles; tile += gridDim.x * kWaves) {
        const uint4 dsc = p.desc[tile];
        const uint32_t cnt = dsc.x & 0x7FFFFFFFu;
        if (cnt == 0 || cnt > (uint32_t)kSlab) continue;
        const uint32_t a_lo = dsc.w;
        const uint32_t cr = p.carry_ref[tile], cd = p.carry_read[tile];
        const uint64_t ob = p.out_base[tile];
        const uint64_t g0 = (uint64_t)tile * kTileOps;
        for (uint32_t r = lane; r < cnt; r += 64) {
            const uint4 rec = p.slab[(uint64_t)tile * kSlab + r];
            const uint32_t loc = rec.x & 0xFFFu, type = (rec.x >> 12) & 1u, prec = (rec.x >> 13) & 1u;
            const uint32_t aln = find_aln(p.aln_off, p.n_aln, a_lo, g0 + loc);
            store_final(p, ob + r, aln, rec.y + (prec ? cr : 0u), rec.z + (prec ? cd : 0u), rec.w, type);
        }
    }
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

__global__ __launch_bounds__(256) void k_cigar_stats(StatsArgs p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t a = blockIdx.x * 4 + wave; a < p.n_aln; a += gridDim.x * 4) {
        const uint64_t b = p.aln_off[a], e = p.aln_off[a + 1];
        // leading soft clips: S ops before the first op that is neither S nor H
        // (pysam getQueryStart)
        uint32_t lead = 0;
        {
            uint64_t i = b;
            bool done = false;
            while (!done && i < e) {
                uint64_t j = i + lane;
                uint32_t w = (j < e) ? p.cigar[j] : 0u;  // op 0 (M) terminates the prefix
                uint32_t op = w & 15u;
                bool clip = (j < e) && (op == 4u || op == 5u);
                uint64_t nb = __ballot(!clip);
                int first = nb ? __ffsll((unsigned long long)nb) - 1 : 64;
                uint32_t s = (lane < first && op == 4u && j < e) ? (w >> 4) : 0u;
                lead += wave_sum(s);
                done = first < 64;
                i += 64;
            }
        }
        uint32_t ref = 0, qal = 0, rl = 0, hard = 0;
        for (uint64_t i = b + lane; i < e; i += 64) {
            const uint32_t w = p.cigar[i], op = w & 15u, len = w >> 4;
            if ((0x18Du >> op) & 1u) ref += len;   // M D N = X  (htslib bam_endpos)
            if ((0x183u >> op) & 1u) qal += len;   // M I = X
            if ((0x1B3u >> op) & 1u) rl += len;    // M I S H = X (infer_read_length)
            if (op == 5u) hard += len;
        }
        ref = wave_sum(ref); qal = wave_sum(qal); rl = wave_sum(rl); hard = wave_sum(hard);
        if (lane == 0) {
            if (p.out.ref_len) p.out.ref_len[a] = ref;
            if (p.out.q_start) p.out.q_start[a] = lead;
            if (p.out.q_end) p.out.q_end[a] = lead + qal;
            if (p.out.read_len) p.out.read_len[a] = rl;
            if (p.out.n_hard) p.out.n_hard[a] = hard;
        }
    }
}

template <bool SOA>
int cigar_extract_dev_impl(svx_ctx* ctx, const uint32_t* d_cigar_or_len, const uint8_t* d_op,
                           uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                           const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out,
                           uint64_t cap, uint64_t* d_n_out) {
    if (!ctx || !d_n_out) return SVX_E_INVALID;
    if (n_ops >= (1ull << 32)) {
        SVX_SET_ERR(ctx, "n_ops=%llu exceeds the 2^32-1 per-call limit; split the batch",
                    (unsigned long long)n_ops);
        return SVX_E_TOO_LARGE;
    }
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    if (n_ops == 0 || n_aln == 0) {
        SVX_HIP(ctx, hipMemsetAsync(d_n_out, 0, sizeof(uint64_t), ctx->stream));
        return SVX_OK;
    }
    if (!d_cigar_or_len || !d_aln_off || (SOA && !d_op)) return SVX_E_INVALID;
    if (cap > 0 && (!d_out.aln || !d_out.ref_pos || !d_out.read_pos || !d_out.len || !d_out.type))
        return SVX_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(d_cigar_or_len) & 15u) ||
        (SOA && (reinterpret_cast<uintptr_t>(d_op) & 15u))) {
        SVX_SET_ERR(ctx, "device CIGAR buffers must be 16-byte aligned");
        return SVX_E_INVALID;
    }
    const uint32_t n_tiles = (uint32_t)((n_ops + kTileOps - 1) / kTileOps);
    size_t need = svx_take_bytes(n_tiles, sizeof(uint4)) +
                  svx_take_bytes((size_t)n_tiles * kSlab, sizeof(uint4)) +
                  4 * svx_take_bytes(n_tiles, sizeof(uint32_t)) + svx_take_bytes(4, sizeof(uint32_t));
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
    a.desc = svx_ws_take<uint4>(ctx, n_tiles);
    a.slab = svx_ws_take<uint4>(ctx, (size_t)n_tiles * kSlab);
    a.out_base = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.carry_ref = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.carry_read = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.dense_list = svx_ws_take<uint32_t>(ctx, n_tiles);
    a.n_dense = svx_ws_take<uint32_t>(ctx, 4);
    a.out = d_out;
    a.cap = cap;

    const uint32_t blocks_all = (n_tiles + kWaves - 1) / kWaves;
    const uint32_t blocks_cap = (uint32_t)ctx->n_cu * 8u;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL((k_cigar_tiles<MODE_STAGE, SOA>), dim3(blocks_all), dim3(64 * kWaves), 0,
                       ctx->stream, a);
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_desc_scan, dim3(1), dim3(1024), 0, ctx->stream, a.desc, n_tiles, a.out_base,
                       a.carry_ref, a.carry_read, a.dense_list, a.n_dense, d_n_out);
    hipLaunchKernelGGL(k_cigar_gather, dim3(blocks_all < blocks_cap ? blocks_all : blocks_cap),
                       dim3(64 * kWaves), 0, ctx->stream, a);
    hipLaunchKernelGGL((k_cigar_tiles<MODE_DIRECT, SOA>),
                       dim3(blocks_all < blocks_cap ? blocks_all : blocks_cap), dim3(64 * kWaves), 0,
                       ctx->stream, a);
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
