// svx_editdist.hip — batched global unit-cost edit distance on gfx950, exact at any length.
//
// Arithmetic of edlib.align(a, b)["editDistance"] (default mode NW) as called by
// compute_distance (reference SVIM_COMBINE.py:50,64,76,88,100).  The algorithm is the
// published bit-vector one edlib itself implements (Myers 1999, block formulation with
// horizontal carries of Hyyrö 2003, Ukkonen's band cut-off); the mapping is for a 64-lane wave:
//
//   * one wave per haplotype pair.  The longer sequence is the pattern (rows), cut into
//     64-row blocks; lane l owns block l of the current strip of 64 blocks (4096 rows) and keeps
//     the block's vertical delta vectors Pv / Mv in two 64-bit registers;
//   * the wave is a systolic array over anti-diagonals of (block, column): at step t lane l
//     processes column t - l.  The horizontal delta leaving the bottom of a block (-1/0/+1) and the
//     text symbol of the column move down one lane per step with one DPP `wave_shr:1` each —
//     no LDS traffic and no barrier on the recurrence; text symbols enter at lane 0 from a
//     64-symbol register chunk (one coalesced load per 64 steps, v_readlane per step);
//   * match vectors Peq[symbol][block] live in LDS, one 8-byte column per lane (conflict-free
//     ds_read_b64), loaded one step ahead of their use.  Symbols are the distinct byte values of
//     the pattern, ranked on the fly (bytes compare exactly: the reference does not fold case of
//     INS alleles); text bytes that do not occur in the pattern share an all-zero row.  The fast
//     instantiation holds 16 symbols (any DNA/IUPAC data); pairs with a richer alphabet are
//     re-run by a 256-symbol instantiation (128 KiB of LDS, one wave per CU);
//   * patterns longer than 4096 rows are processed strip after strip; the horizontal deltas
//     along a strip's bottom row are packed 2 bits per column into an HBM scratch stream that
//     feeds lane 0 of the next strip;
//   * Ukkonen band at strip granularity: strip s only visits columns within k of its rows.
//     Cells outside are replaced by valid path costs (deltas of +1), so the result D' is always
//     an upper bound and equals the distance when D' <= k (or when nothing was cut).
//
// Result contract (include/svx.h): exact distance when <= k_max, 0xFFFFFFFF otherwise;
// k_max = 0xFFFFFFFF requests exact distances: unresolved pairs are re-run with a 4x wider band
// until the band covers the whole matrix.
//
// VALU-bound integer work (≈ 50 VALU per 64 cells); no MFMA, HBM traffic is negligible.
#include <cmath>
#include <type_traits>
#include <cstdlib>

#include "svx_internal.h"

#include <algorithm>
#include <numeric>

namespace {

#ifndef SVX_ED_PREFETCH
#define SVX_ED_PREFETCH 0  // 1: text chunks requested a whole chunk ahead — measured slower (5.76 vs 5.51 ms, profiles/README.md)
#endif
constexpr uint32_t kStripRows = 64 * 64;
constexpr uint32_t kUnproven = 0x80000000u;   // flag: D' > band and cells were cut (upper bound only)
constexpr uint32_t kAlphabet = 0xFFFFFFFEu;   // marker: pattern alphabet exceeds this instantiation
constexpr int kDppWaveShr1 = 0x138;

struct EdArgs {
    const uint8_t* seq;
    const uint64_t* a_off;
    const uint32_t* a_len;
    const uint64_t* b_off;
    const uint32_t* b_len;
    const uint32_t* order;        // pair processed by workgroup w (longest first)
    const uint64_t* stream_off;   // per launch slot w: offset (in u64 words) of its 2 delta streams
    uint64_t* stream;             // HBM scratch for strip boundaries
    const uint32_t* band;         // per launch slot w: Ukkonen band k of that pair
    uint32_t n;
    uint32_t* dist;
};

__device__ __forceinline__ uint32_t shr1_in(uint32_t fresh, uint32_t v) {
    // lane l receives v of lane l-1; lane 0 receives `fresh`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fresh, (int)v, kDppWaveShr1, 0xF, 0xF, false);
}

__device__ __forceinline__ uint32_t shr1_zero(uint32_t v) {
    // lane l receives v of lane l-1; lane 0 receives 0 (bound_ctrl) — no move in front of the DPP instruction, and an OR
    // with it folds into one v_or_b32_dpp
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kDppWaveShr1, 0xF, 0xF, true);
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CAP>
__global__ __launch_bounds__(64) void k_edit_myers(EdArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    uint64_t* peq = lds;                                                  // [(CAP + 1) * 64]
    uint8_t* cmap = reinterpret_cast<uint8_t*>(peq + (CAP + 1) * 64);     // byte value -> symbol (CAP: absent)
    uint32_t* present = reinterpret_cast<uint32_t*>(cmap + 256);          // 256-bit set of pattern bytes
    const uint32_t lane = threadIdx.x;
    const uint32_t w = blockIdx.x;
    const uint32_t pi = p.order ? p.order[w] : w;
    uint32_t la = p.a_len[pi], lb = p.b_len[pi];
    const uint8_t* A = p.seq + p.a_off[pi];
    const uint8_t* B = p.seq + p.b_off[pi];
    // pattern = longer sequence (more lanes busy, fewer steps); the distance is symmetric
    const uint8_t* P = la >= lb ? A : B;
    const uint8_t* T = la >= lb ? B : A;
    const uint32_t m = la >= lb ? la : lb, n = la >= lb ? lb : la;
    if (n == 0) {
        if (lane == 0) p.dist[pi] = m;
        return;
    }
    const uint32_t k = p.band[w];
    const uint32_t n_strips = (m + kStripRows - 1) / kStripRows;
    if (n_strips > 1 && m - n > k) {  // the end cell lies outside the band: only "> k" is known
        if (lane == 0) p.dist[pi] = m | kUnproven;
        return;
    }
    // ---- alphabet of the pattern
    if (lane < 8) present[lane] = 0;
    wave_lds_fence();
    for (uint32_t i = lane; i < m; i += 64) {
        const uint32_t c = P[i];
        atomicOr(&present[c >> 5], 1u << (c & 31));
    }
    wave_lds_fence();
    uint32_t sigma = 0;
    {
        uint32_t below[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            below[q] = sigma;
            sigma += __popc(present[q]);
        }
        if (sigma > (uint32_t)CAP) {
            if (lane == 0) p.dist[pi] = kAlphabet;
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = lane + 64 * q;
            const uint32_t word = present[v >> 5], bit = v & 31;
            uint32_t rank = __popc(word & ((1u << bit) - 1u));
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += ((uint32_t)j == (v >> 5)) ? below[j] : 0u;
            cmap[v] = ((word >> bit) & 1u) ? (uint8_t)rank : (uint8_t)CAP;
        }
    }
    wave_lds_fence();

    uint64_t* s_in = nullptr;
    uint64_t* s_out = nullptr;
    if (n_strips > 1) {
        const uint64_t words = ((uint64_t)n + 31) / 32;
        s_in = p.stream + p.stream_off[w];
        s_out = s_in + words;
    }
    if (n_strips == 1) {
        // ---- the pattern fits one strip (<= 4096 rows: every pair of the PAIR step but the contig-sized alleles): no band,
        // no boundary stream, the top row's horizontal delta is +1 everywhere — a leaner step: the score is accumulated by
        // every lane without a branch (only lane `last` holds the matrix's last row), the activity test is one subtract
        // and one compare against a per-lane limit, the horizontal deltas are split with bit operations.
        const uint32_t nblk = (m + 63) / 64, last = nblk - 1, outbit = (m - 1) & 63;
        for (uint32_t c = 0; c <= min(sigma, (uint32_t)CAP); ++c) peq[c * 64 + lane] = 0;
        peq[CAP * 64 + lane] = 0;
        if (lane < nblk) {
            const uint32_t r0 = 64 * lane;
            for (uint32_t r = 0; r < 64 && r0 + r < m; ++r) peq[cmap[P[r0 + r]] * 64 + lane] |= 1ull << r;
        }
        wave_lds_fence();
        const uint32_t nsteps = n + nblk - 1;
        const uint32_t lim = lane < nblk ? n : 0u;          // steps this lane is active for, from step `lane` on
        const uint32_t sh = lane == last ? outbit : 63u;
        uint64_t Pv = ~0ull, Mv = 0;
        // the horizontal delta of a block's bottom row travels to the next lane as two separate bits (+1, -1): nothing to
        // pack and unpack per step; lane 0's +1 (the matrix's top row) is a per-lane constant ORed into the shifted value
        uint32_t val = m, hp_prev = 0, hm_prev = 0, code_cur = CAP, codechunk = CAP;
        const uint32_t top_plus = lane == 0 ? 1u : 0u;
        // text symbols: 64 per register chunk; the chunk after the current one is requested a whole chunk ahead (its
        // bytes at step 0 of a chunk, their symbol codes at step 32) so that no step waits for memory
        auto fetch_raw = [&](uint32_t t0) -> uint32_t {
            const uint32_t col = 1 + t0 + lane;
            return col <= n ? (uint32_t)T[col - 1] : 0x100u;
        };
        auto to_code = [&](uint32_t raw) -> uint32_t { return raw < 0x100u ? (uint32_t)cmap[raw] : (uint32_t)CAP; };
        codechunk = to_code(fetch_raw(0));
        uint32_t raw_ahead = 0x100u, code_ahead = CAP;
        code_cur = shr1_in((uint32_t)__builtin_amdgcn_readlane((int)codechunk, 0), code_cur);
        uint64_t eq_cur = peq[code_cur * 64 + lane];
        // Two steps per trip: the symbol and the match vector of the next step change hands by name, not by moves.
        // Steady steps — every block of the strip has started, none has finished: t in [last, n) — need no activity test:
        // the mask is the loop-invariant `lane < nblk`.
        const bool in_strip = lane < nblk;
        auto one_step = [&](auto steady_c, const uint32_t t, const uint32_t code_cur, const uint64_t eq_cur, uint32_t& code_next,
                            uint64_t& eq_next) {
#if SVX_ED_PREFETCH
            if ((t & 63) == 0) raw_ahead = fetch_raw(t + 64);
            if ((t & 63) == 32) code_ahead = to_code(raw_ahead);
            if ((t & 63) == 63) codechunk = code_ahead;
#else
            if ((t & 63) == 63) { raw_ahead = fetch_raw(t + 1); code_ahead = to_code(raw_ahead); codechunk = code_ahead; }
#endif
            const uint32_t fresh_c = (uint32_t)__builtin_amdgcn_readlane((int)codechunk, (int)((t + 1) & 63));
            code_next = shr1_in(fresh_c, code_cur);
            eq_next = peq[code_next * 64 + lane];
            const bool active = decltype(steady_c)::value ? in_strip : (t - lane) < lim;  // (t < lane wraps to a huge value)
            const uint64_t hneg = shr1_zero(hm_prev), hpos = shr1_zero(hp_prev) | top_plus;
            uint64_t Eq = eq_cur;
            const uint64_t Xv = Eq | Mv;
            Eq |= hneg;
            const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint64_t Ph = Mv | ~(Xh | Pv);
            uint64_t Mh = Pv & Xh;
            const uint32_t hop = (uint32_t)(Ph >> sh) & 1u, hom = (uint32_t)(Mh >> sh) & 1u;
            Ph = (Ph << 1) | hpos;
            Mh = (Mh << 1) | hneg;
            if (active) {
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                hp_prev = hop;
                hm_prev = hom;
                val += hop - hom;
            }
        };
        {
            uint32_t code_b = CAP;
            uint64_t eq_b = 0;
            uint32_t t = 0;
            auto run = [&](const uint32_t t_end, auto steady_c) {
                for (; t + 1 < t_end; t += 2) {
                    one_step(steady_c, t, code_cur, eq_cur, code_b, eq_b);
                    one_step(steady_c, t + 1, code_b, eq_b, code_cur, eq_cur);
                }
                if (t < t_end) {
                    one_step(steady_c, t, code_cur, eq_cur, code_b, eq_b);
                    code_cur = code_b; eq_cur = eq_b;
                    ++t;
                }
            };
            const uint32_t t_s0 = last < nsteps ? last : nsteps, t_s1 = n > t_s0 ? n : t_s0;
            run(t_s0, std::false_type{});   // blocks joining
            run(t_s1, std::true_type{});    // all of them at work
            run(nsteps, std::false_type{}); // blocks leaving
        }
        const uint32_t result1 = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)last);
        if (lane == 0) p.dist[pi] = result1;
        return;
    }
    uint32_t base = 0;           // D'[last row of the strip][c_lo - 1]
    uint32_t prev_c_hi = 0;
    uint32_t carried = 0;        // D'[last row of the previous strip][c_lo of this strip - 1]
    bool cut = false;            // some strip skipped columns
    uint32_t result = 0;
    for (uint32_t s = 0; s < n_strips; ++s) {
        const uint32_t row_lo = s * kStripRows + 1;
        const uint32_t row_hi = min(m, (s + 1) * kStripRows);
        const uint32_t height = row_hi - row_lo + 1;
        const uint32_t nblk = (height + 63) / 64;
        const uint32_t outbit = (s + 1 == n_strips) ? ((height - 1) & 63) : 63;
        uint32_t c_lo = 1, c_hi = n;
        if (n_strips > 1) {
            c_lo = row_lo > (uint64_t)k + 1 ? row_lo - k : 1;
            c_hi = (uint32_t)min((uint64_t)n, (uint64_t)row_hi + k);
            if (c_lo != 1 || c_hi != n) cut = true;
        }
        // where the next strip starts (its left neighbour column is captured on the way)
        uint32_t cap_col = 0xFFFFFFFFu;
        if (s + 1 < n_strips) {
            const uint32_t nlo = row_hi + 1;
            const uint32_t ncl = nlo > (uint64_t)k + 1 ? nlo - k : 1;
            cap_col = ncl - 1;
        }
        base = (s == 0 ? c_lo - 1 : carried) + height;
        // ---- Peq of this strip's blocks (lane-private LDS columns)
        for (uint32_t c = 0; c <= min(sigma, (uint32_t)CAP); ++c) peq[c * 64 + lane] = 0;
        peq[CAP * 64 + lane] = 0;
        if (lane < nblk) {
            const uint32_t r0 = row_lo - 1 + 64 * lane;  // 0-based pattern index of the block's first row
            for (uint32_t r = 0; r < 64 && r0 + r < row_hi; ++r) {
                const uint32_t code = cmap[P[r0 + r]];
                peq[code * 64 + lane] |= 1ull << r;
            }
        }
        wave_lds_fence();

        const uint32_t ncols = c_hi - c_lo + 1;
        const uint32_t nsteps = ncols + nblk - 1;
        const uint32_t last = nblk - 1;
        uint64_t Pv = ~0ull, Mv = 0;
        uint32_t val = base;        // running D' on the strip's last row (meaningful on lane `last`)
        // ... and its minimum over the strip's columns, column 0 included when the strip reaches it (Ukkonen's cut-off
        // test below; the cell left of a band-limited strip lies outside the band)
        uint32_t row_min = c_lo == 1 ? base : 0xFFFFFFFFu;
        uint32_t captured = base;   // value at cap_col (== c_lo - 1 unless seen later)
        uint32_t acc_lo = 0, acc_hi = 0;  // packed outgoing deltas (lane `last`), 16 columns per half
        uint32_t hp_prev = 0, hm_prev = 0, code_cur = CAP;
        const uint32_t lane0 = lane == 0 ? 1u : 0u;
        uint32_t codechunk = CAP, hchunk = 1;
        uint64_t eq_cur = 0;
        // text symbols and incoming deltas: 64 per register chunk, the next chunk requested a whole chunk ahead
        uint32_t raw_ahead = 0x100u, code_ahead = CAP, h_ahead = 1;
        uint64_t hword_ahead = 0;
        auto fetch_raw = [&](uint32_t t0) {
            const uint32_t col = c_lo + t0 + lane;
            raw_ahead = col <= c_hi ? (uint32_t)T[col - 1] : 0x100u;
            hword_ahead = (col <= c_hi && s > 0 && col <= prev_c_hi) ? s_in[(col - 1) >> 5] : ~0ull;
        };
        auto decode_ahead = [&](uint32_t t0) {
            const uint32_t col = c_lo + t0 + lane;
            code_ahead = raw_ahead < 0x100u ? (uint32_t)cmap[raw_ahead] : (uint32_t)CAP;
            // +1: top row of the matrix, or columns the previous strip did not visit
            h_ahead = (col <= c_hi && s > 0 && col <= prev_c_hi) ? (uint32_t)(hword_ahead >> (2 * ((col - 1) & 31))) & 3u : 1u;
        };
        fetch_raw(0);
        decode_ahead(0);
        codechunk = code_ahead;
        hchunk = h_ahead;
        code_cur = shr1_in((uint32_t)__builtin_amdgcn_readlane((int)codechunk, 0), code_cur);
        eq_cur = peq[code_cur * 64 + lane];
        // (the lean step of the single-strip path above, plus what a strip boundary needs: the incoming deltas of the
        //  strip above, the packed outgoing ones — only lane `last`'s are real; every lane shifts by lane `last`'s
        //  column, a scalar — and the score at the column where the next strip starts)
        const uint32_t lim = lane < nblk ? ncols : 0u;
        const bool in_strip = lane < nblk;
        const uint32_t sh = lane == last ? outbit : 63u;
        const uint32_t t_cap = (cap_col != 0xFFFFFFFFu && cap_col >= c_lo) ? cap_col - c_lo + last : 0xFFFFFFFFu;
        // (the pattern's LAST strip hands nothing on: no outgoing stream, no captured score — a tenth of its step)
        auto strip_steps = [&](auto hands_on_c) {
        constexpr bool kHandsOn = decltype(hands_on_c)::value;
        auto one_step = [&](auto steady_c, const uint32_t t, const uint32_t code_cur, const uint64_t eq_cur, uint32_t& code_next,
                            uint64_t& eq_next) {
            const uint32_t tl = t & 63;
            const uint32_t fresh_h = (uint32_t)__builtin_amdgcn_readlane((int)hchunk, (int)tl);
            // next step's symbol and match vector (independent of this step's recurrence)
#if SVX_ED_PREFETCH
            if (tl == 0) fetch_raw(t + 64);
            if (tl == 32) decode_ahead(t + 32);
            if (tl == 63) { codechunk = code_ahead; hchunk = h_ahead; }
#else
            if (tl == 63) { fetch_raw(t + 1); decode_ahead(t + 1); codechunk = code_ahead; hchunk = h_ahead; }
#endif
            const uint32_t fresh_c = (uint32_t)__builtin_amdgcn_readlane((int)codechunk, (int)((t + 1) & 63));
            code_next = shr1_in(fresh_c, code_cur);
            eq_next = peq[code_next * 64 + lane];

            const bool active = decltype(steady_c)::value ? in_strip : (t - lane) < lim;  // (t < lane wraps to a huge value)
            // lane 0 takes the strip above's delta for this column (fresh_h: a scalar, two bits), the others their upper
            // neighbour's, as two separate bits
            const uint64_t hneg = shr1_zero(hm_prev) | (lane0 & (fresh_h >> 1)), hpos = shr1_zero(hp_prev) | (lane0 & fresh_h);
            uint64_t Eq = eq_cur;
            const uint64_t Xv = Eq | Mv;
            Eq |= hneg;
            const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint64_t Ph = Mv | ~(Xh | Pv);
            uint64_t Mh = Pv & Xh;
            const uint32_t hop = (uint32_t)(Ph >> sh) & 1u, hom = (uint32_t)(Mh >> sh) & 1u;
            const uint32_t ho = hop | (hom << 1);  // (only the hand-over strips use the packed form)
            Ph = (Ph << 1) | hpos;
            Mh = (Mh << 1) | hneg;
            if (active) {
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                hp_prev = hop;
                hm_prev = hom;
                val += hop - hom;
                if (kHandsOn) row_min = val < row_min ? val : row_min;
            }
            if (kHandsOn) {
                captured = t == t_cap ? val : captured;
                if (t >= last) {  // wave-uniform: lane `last` is at column col_last — and active from here to the end, so
                                  // its `ho` is the delta that leaves the strip; the other lanes' words are never stored
                    const uint32_t col_last = c_lo + t - last;
                    const uint32_t pos = (col_last - 1) & 31;  // scalar: one fused shift-or on the half it falls into
                    if (pos < 16) acc_lo |= ho << (2 * pos);
                    else acc_hi |= ho << (2 * pos - 32);
                    if (pos == 31 || col_last == c_hi) {
                        if (lane == last) s_out[(col_last - 1) >> 5] = ((uint64_t)acc_hi << 32) | acc_lo;
                        acc_lo = acc_hi = 0;
                    }
                }
            }
        };
        uint32_t code_b = CAP;
        uint64_t eq_b = 0;
        uint32_t t = 0;
        auto run = [&](const uint32_t t_end, auto steady_c) {
            for (; t + 1 < t_end; t += 2) {
                one_step(steady_c, t, code_cur, eq_cur, code_b, eq_b);
                one_step(steady_c, t + 1, code_b, eq_b, code_cur, eq_cur);
            }
            if (t < t_end) {
                one_step(steady_c, t, code_cur, eq_cur, code_b, eq_b);
                code_cur = code_b; eq_cur = eq_b;
                ++t;
            }
        };
        const uint32_t t_s0 = last < nsteps ? last : nsteps, t_s1 = ncols > t_s0 ? ncols : t_s0;
        run(t_s0, std::false_type{});   // blocks joining
        run(t_s1, std::true_type{});    // all of them at work: no activity test (see the single-strip path)
        run(nsteps, std::false_type{}); // blocks leaving
        };
        if (s_out && s + 1 < n_strips) strip_steps(std::true_type{});
        else strip_steps(std::false_type{});
        carried = (uint32_t)__builtin_amdgcn_readlane((int)captured, (int)last);
        result = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)last);
        // Ukkonen's cut-off at strip granularity: every alignment crosses this strip's last row, inside the band if it
        // costs at most k (outside, |row - column| > k already), and a cell whose true value is <= k has D' = D (its
        // best path never left the band).  So when EVERY value on the row exceeds k, so does the distance: the strips
        // below need not be visited — a divergent pair (two unrelated alleles of a crowded partition, asked only
        // "<= max_edit_distance?") ends after its first strip, a too narrow band of an exact request where it fails.
        if (s + 1 < n_strips) {
            const uint32_t rmin = (uint32_t)__builtin_amdgcn_readlane((int)row_min, (int)last);
            if (rmin > k) {
                // (an upper bound for the host's next band: from the strip's last visited cell straight down and right)
                const uint64_t ub = (uint64_t)result + (m - row_hi) + (n > c_hi ? n - c_hi : 0u);
                if (lane == 0) p.dist[pi] = (uint32_t)(ub < 0x7FFFFFFFu ? ub : 0x7FFFFFFFu) | kUnproven;
                return;
            }
        }
        prev_c_hi = c_hi;
        if (s_out) {
            // the next strip reads what lane `last` just wrote (same wave): make it visible
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            uint64_t* tmp = s_in;
            s_in = s_out;
            s_out = tmp;
        }
        wave_lds_fence();
    }
    if (lane == 0) p.dist[pi] = (cut && result > k) ? (result | kUnproven) : result;
}

template <int CAP>
constexpr size_t lds_bytes() { return (size_t)(CAP + 1) * 64 * 8 + 256 + 32; }

// ---- wavefront (diagonal-transition) pass in front of the bit-vector kernel -------------------------
// The haplotypes the PAIR step compares are long and nearly identical (two alleles of one variant plus
// 100 bp flanks): the distance d is tiny next to the lengths.  Ukkonen / Landau-Vishkin / Myers'
// O(n + d^2) scheme — the published algorithm behind WFA — computes exactly the unit-cost global distance
// edlib returns: H_s[k] = furthest row i reachable on diagonal k = j - i with s edits,
//     H_s[k] = extend(max(H_{s-1}[k-1], H_{s-1}[k] + 1, H_{s-1}[k+1] + 1)),   d = min{s : H_s[n - m] = m},
// where extend() slides down the diagonal while the bytes match.  One wave per pair: lane = diagonal
// (64 per iteration), the two live wavefronts in LDS; every lane first compares 8 bytes, runs that are
// longer are finished by the whole wave 512 bytes per step.  Run for at most `cap` edits (the caller's
// threshold, or a work bound); pairs that need more are left to the bit-vector kernel.
struct WfaArgs {
    const uint8_t* seq;
    const uint64_t* a_off;
    const uint32_t* a_len;
    const uint64_t* b_off;
    const uint32_t* b_len;
    const uint32_t* order;
    const uint32_t* cap;   // per launch slot: largest distance to resolve
    uint32_t n;
    uint32_t* dist;        // exact distance, or kUnproven | cap when it exceeds cap
};

constexpr int32_t kWfaNeg = -(1 << 30);

// 8 bytes at an arbitrary address from two aligned 8-byte loads (reads up to 15 bytes past p: the pool
// lives inside the staging block, what lies behind it is masked off by the caller's limit)
__device__ __forceinline__ uint64_t load8_unaligned(const uint8_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint64_t* q = reinterpret_cast<const uint64_t*>(a & ~(uintptr_t)7);
    const uint32_t sh = (uint32_t)(a & 7u) * 8u;
    const uint64_t lo = q[0], hi = q[1];
    return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// number of leading equal bytes of pa[0..lim) and pb[0..lim), lim <= 8
__device__ __forceinline__ uint32_t match8(const uint8_t* pa, const uint8_t* pb, uint32_t lim) {
    const uint64_t x = load8_unaligned(pa) ^ load8_unaligned(pb);
    const uint32_t e = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
    return e < lim ? e : lim;
}

__global__ __launch_bounds__(64) void k_edit_wfa(WfaArgs p, uint32_t lds_cap) {
    extern __shared__ __attribute__((aligned(16))) int32_t wf[];
    const int lane = threadIdx.x;
    const uint32_t w = blockIdx.x;
    const uint32_t pi = p.order[w];
    const uint32_t m = p.a_len[pi], n = p.b_len[pi];
    const uint8_t* A = p.seq + p.a_off[pi];
    const uint8_t* B = p.seq + p.b_off[pi];
    const uint32_t cap = p.cap[w] < lds_cap ? p.cap[w] : lds_cap;
    if (m == 0 || n == 0) {
        if (lane == 0) p.dist[pi] = m > n ? m : n;
        return;
    }
    const int32_t kend = (int32_t)n - (int32_t)m;
    if ((uint32_t)(kend < 0 ? -kend : kend) > cap) {  // at least |n - m| edits
        if (lane == 0) p.dist[pi] = kUnproven | cap;
        return;
    }
    const int32_t width = 2 * (int32_t)lds_cap + 3, zero = (int32_t)lds_cap + 1;
    int32_t* cur = wf;
    int32_t* prev = wf + width;
    for (int32_t i = lane; i < 2 * width; i += 64) wf[i] = kWfaNeg;
    wave_lds_fence();

    // whole-wave extension along one diagonal: equal bytes of pa / pb, at most rem
    auto coop_extend = [&](const uint8_t* pa, const uint8_t* pb, uint32_t rem) -> uint32_t {
        uint32_t done = 0;
        for (;;) {
            const uint32_t off = done + 8u * (uint32_t)lane;
            const uint32_t lim = off < rem ? (rem - off < 8u ? rem - off : 8u) : 0u;
            const uint32_t e = lim ? match8(pa + off, pb + off, lim) : 0u;
            const uint64_t stop = __ballot(e < 8u);
            if (stop) {
                const int f = __ffsll((unsigned long long)stop) - 1;
                return done + 8u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e, f);
            }
            done += 512u;
        }
    };

    {   // s = 0: the common prefix
        const uint32_t i0 = coop_extend(A, B, m < n ? m : n);
        if (kend == 0 && i0 == m) {
            if (lane == 0) p.dist[pi] = 0;
            return;
        }
        if (lane == 0) cur[zero] = (int32_t)i0;
        wave_lds_fence();
    }
    for (uint32_t s = 1; s <= cap; ++s) {
        int32_t* t = cur; cur = prev; prev = t;
        const int32_t lo = -(int32_t)(s < m ? s : m), hi = (int32_t)(s < n ? s : n);
        for (int32_t kbase = lo; kbase <= hi; kbase += 64) {
            const int32_t k = kbase + lane;
            const bool valid = k <= hi;
            int32_t i = kWfaNeg;
            uint32_t rem = 0;
            if (valid) {
                const int32_t x0 = prev[zero + k - 1], x1 = prev[zero + k] + 1, x2 = prev[zero + k + 1] + 1;
                int32_t x = x0 > x1 ? x0 : x1;
                x = x > x2 ? x : x2;
                const int32_t bound = (int32_t)m < (int32_t)n - k ? (int32_t)m : (int32_t)n - k;  // i <= m, i + k <= n
                if (x >= 0) {
                    i = x < bound ? x : bound;
                    rem = (uint32_t)(bound - i);
                }
            }
            // every lane: the next 8 bytes of its diagonal; longer runs are finished by the whole wave
            uint32_t e = 0;
            if (rem) e = match8(A + i, B + i + k, rem < 8u ? rem : 8u);
            i += (int32_t)e;
            uint64_t more = __ballot(rem > 8u && e == 8u);
            while (more) {
                const int l = __ffsll((unsigned long long)more) - 1;
                more &= more - 1;
                const int32_t il = __builtin_amdgcn_readlane(i, l), kl = kbase + l;
                const uint32_t reml = (uint32_t)__builtin_amdgcn_readlane((int)rem, l) - 8u;
                const uint32_t run = coop_extend(A + il, B + il + kl, reml);
                if (lane == l) i += (int32_t)run;
            }
            if (valid) cur[zero + k] = i;
        }
        wave_lds_fence();
        if ((uint32_t)(kend < 0 ? -kend : kend) <= s && cur[zero + kend] >= (int32_t)m) {  // wave-uniform LDS read
            if (lane == 0) p.dist[pi] = s;
            return;
        }
    }
    if (lane == 0) p.dist[pi] = kUnproven | cap;
}

// ---- haplotype strings on device -------------------------------------------------------------------
// compute_distance (SVIM_COMBINE.py:43-100) aligns, for two candidates of one partition,
//     reference[region_start : c.start] + MIDDLE + reference[c.end : region_end]
// with MIDDLE = "" (DEL), the reverse complement of reference[c.start : c.end] (INV), that interval
// repeated copies + 1 times (DUP_TAN), the inserted sequence (INS) or the source interval (DUP_INT);
// every reference slice upper-cased, inserted sequences as they are.  A haplotype is therefore three
// PIECES of a byte pool (reference windows fetched once per partition + the inserted sequences), each
// {offset, length, repeat count, flags}.  k_hap_build writes the strings (one workgroup per string,
// coalesced copies, upper-casing / complementing on the way) into the pool the edit-distance kernel
// reads: the host never slices or concatenates a string.
struct HapArgs {
    const uint8_t* pool;
    const svx_hap_piece* pieces;  // 3 per string, 6 per pair (a then b)
    const uint64_t* str_off;      // [2 * n_pairs] where string w starts in `out`
    uint8_t* out;
    uint32_t n_strings;
};

__device__ __forceinline__ uint8_t hap_byte(uint8_t c, uint32_t flags) {
    if (flags & SVX_PIECE_UPPER) c = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;   // str.upper(), ASCII
    if (flags & SVX_PIECE_REVCOMP) {  // SVIM_COMBINE.py:63: complement of A, C, G, T; anything else stays
        c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
    }
    return c;
}

__global__ __launch_bounds__(256) void k_hap_build(HapArgs p) {
    const uint32_t w = blockIdx.x;
    if (w >= p.n_strings) return;
    uint8_t* dst = p.out + p.str_off[w];
    uint64_t at = 0;
    for (int k = 0; k < 3; ++k) {
        const svx_hap_piece pc = p.pieces[(size_t)w * 3 + k];
        const uint8_t* src = p.pool + pc.off;
        const uint64_t total = (uint64_t)pc.len * pc.repeat;
        for (uint64_t i = threadIdx.x; i < total; i += 256) {
            const uint32_t j = (uint32_t)(i % pc.len);
            const uint8_t c = src[(pc.flags & SVX_PIECE_REVCOMP) ? pc.len - 1 - j : j];
            dst[at + i] = hap_byte(c, pc.flags);
        }
        at += total;
    }
}

// ---- host side of the batched distance: everything after the sequence pool is in HBM ---------------
struct EdPlan {
    std::vector<uint32_t> order;
    uint64_t total_stream = 0;
};

static uint64_t ed_stream_words(uint32_t la, uint32_t lb) {
    const uint64_t m = std::max(la, lb), n = std::min(la, lb);
    return m > kStripRows ? 2 * ((n + 31) / 32) : 0;
}

static int ed_validate_and_plan(svx_ctx* ctx, uint64_t seq_bytes, const uint64_t* a_off, const uint32_t* a_len,
                                const uint64_t* b_off, const uint32_t* b_len, uint32_t n_pairs, EdPlan* plan) {
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (a_off[i] + a_len[i] > seq_bytes || b_off[i] + b_len[i] > seq_bytes) {
            SVX_SET_ERR(ctx, "pair %u reads past the sequence pool", i);
            return SVX_E_INVALID;
        }
        if (a_len[i] >= kUnproven || b_len[i] >= kUnproven) {
            SVX_SET_ERR(ctx, "pair %u: sequences of 2^31 bytes or more are not supported", i);
            return SVX_E_TOO_LARGE;
        }
    }
    // launch order: most expensive pairs first (steps ~ strips * columns), so that the long tail of a
    // few contig-sized alleles overlaps the many small ones
    plan->order.resize(n_pairs);
    std::iota(plan->order.begin(), plan->order.end(), 0u);
    auto cost = [&](uint32_t i) {
        const uint64_t m = std::max(a_len[i], b_len[i]), n = std::min(a_len[i], b_len[i]);
        return ((m + kStripRows - 1) / kStripRows) * (n + 64);
    };
    std::stable_sort(plan->order.begin(), plan->order.end(), [&](uint32_t x, uint32_t y) { return cost(x) > cost(y); });
    plan->total_stream = 0;
    for (uint32_t w = 0; w < n_pairs; ++w) plan->total_stream += ed_stream_words(a_len[plan->order[w]], b_len[plan->order[w]]);
    return SVX_OK;
}

static size_t ed_stage_need(uint32_t n_pairs, const EdPlan& plan) {
    return 3 * svx_take_bytes(n_pairs, 8) + 5 * svx_take_bytes(n_pairs, 4) +
           svx_take_bytes(plan.total_stream ? plan.total_stream : 1, 8);
}

// d_seq: the sequence pool in HBM (already uploaded / built on the context's stream); the stage region
// must have been reserved with ed_stage_need() bytes still free
// k_of != nullptr: per-pair threshold (0xFFFFFFFF = exact) instead of the one k_max for all
static int ed_run(svx_ctx* ctx, const uint8_t* d_seq, const uint64_t* a_off, const uint32_t* a_len,
                  const uint64_t* b_off, const uint32_t* b_len, uint32_t n_pairs, uint32_t k_max, uint32_t* dist,
                  const EdPlan& plan, const uint32_t* k_of = nullptr) {
    auto kmax_of = [&](uint32_t i) { return k_of ? k_of[i] : k_max; };
    auto exact_of = [&](uint32_t i) { return kmax_of(i) == 0xFFFFFFFFu; };
    const std::vector<uint32_t>& order = plan.order;
    int rc;
    EdArgs a;
    uint64_t* d_ao = svx_stage_take<uint64_t>(ctx, n_pairs);
    uint64_t* d_bo = svx_stage_take<uint64_t>(ctx, n_pairs);
    uint64_t* d_so = svx_stage_take<uint64_t>(ctx, n_pairs);
    uint32_t* d_al = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_bl = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_ord = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_band = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint32_t* d_dist = svx_stage_take<uint32_t>(ctx, n_pairs);
    uint64_t* d_stream = svx_stage_take<uint64_t>(ctx, plan.total_stream ? plan.total_stream : 1);
    SVX_HIP(ctx, hipMemcpyAsync(d_ao, a_off, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_bo, b_off, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_al, a_len, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_bl, b_len, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    a.seq = d_seq; a.a_off = d_ao; a.a_len = d_al; a.b_off = d_bo; a.b_len = d_bl;
    a.dist = d_dist;
    a.stream = d_stream;
    a.stream_off = d_so;
    a.order = d_ord;
    a.band = d_band;
    SVX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_edit_myers<256>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<256>()));
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    // band per pair: the threshold itself, or (exact request) 256 widened per pair below
    std::vector<uint32_t> band_of(n_pairs);
    for (uint32_t i = 0; i < n_pairs; ++i) band_of[i] = exact_of(i) ? 256u : kmax_of(i);
    std::vector<uint32_t> todo = order, big, again, lst_band;
    std::vector<uint64_t> lst_soff;
    // ---- wavefront pass: resolves every pair whose distance is small next to its length
    if (ctx->wfa_cap > 0) {
        lst_band.resize(n_pairs);
        uint32_t lds_cap = 1;
        for (uint32_t w = 0; w < n_pairs; ++w) {
            const uint32_t i = order[w];
            // at most the threshold; and no more edits than 1/32 of the lengths: beyond that the d^2 / 64 wave steps
            // of this pass (each with its own round trip to the sequences) cost more than the bit-vector kernel's
            // strips.  Measured on PAIR's exact batch of config 5 (crowded partitions: unrelated events, distances of
            // many hundreds over 2-3 kb): 1/8 -> 1/32 takes the distances of the step from 48 to 40 ms; 1/16 and
            // 1/64 the same within noise, the near-identical batches of the bench unchanged (profiles/README.md)
            uint64_t c = std::max<uint64_t>(64, ((uint64_t)a_len[i] + b_len[i]) / 32);
            c = std::min<uint64_t>(c, ctx->wfa_cap);
            if (!exact_of(i)) c = std::min<uint64_t>(c, kmax_of(i));
            lst_band[w] = (uint32_t)c;
            lds_cap = std::max(lds_cap, (uint32_t)c);
        }
        SVX_HIP(ctx, hipMemcpyAsync(d_ord, order.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
        SVX_HIP(ctx, hipMemcpyAsync(d_band, lst_band.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
        WfaArgs wa;
        wa.seq = d_seq; wa.a_off = d_ao; wa.a_len = d_al; wa.b_off = d_bo; wa.b_len = d_bl;
        wa.order = d_ord; wa.cap = d_band; wa.n = n_pairs; wa.dist = d_dist;
        const size_t lds = (size_t)2 * (2 * (size_t)lds_cap + 3) * sizeof(int32_t);
        if (lds > 32768)  // (the largest cap, 4096 edits, needs 65 560 bytes of dynamic LDS: above the default limit)
            SVX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_edit_wfa),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_edit_wfa, dim3(n_pairs), dim3(64), lds, ctx->stream, wa, lds_cap);
        SVX_HIP(ctx, hipGetLastError());
        SVX_HIP(ctx, hipMemcpyAsync(dist, d_dist, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
        { int wrc = svx_wait_blocking(ctx); if (wrc != SVX_OK) return wrc; }
        todo.clear();
        for (uint32_t w = 0; w < n_pairs; ++w) {
            const uint32_t i = order[w];
            if (!(dist[i] & kUnproven)) continue;                  // exact distance <= cap
            if (!exact_of(i) && lst_band[w] >= kmax_of(i)) continue;          // distance > k_max: all the caller asked
            band_of[i] = exact_of(i) ? std::max<uint32_t>(256u, 2 * lst_band[w]) : kmax_of(i);   // distance > cap is known
            todo.push_back(i);
        }
    }
    // (pairs the rounds below do not visit keep what the wavefront pass wrote: d_dist holds it)
    while (!todo.empty()) {
        // one round: fast instantiation over `todo`, then the 256-symbol one over what it rejected
        for (int pass = 0; pass < 2; ++pass) {
            const std::vector<uint32_t>& lst = pass == 0 ? todo : big;
            if (lst.empty()) continue;
            uint64_t at = 0;
            lst_soff.resize(lst.size());
            lst_band.resize(lst.size());
            for (size_t w2 = 0; w2 < lst.size(); ++w2) {
                lst_soff[w2] = at;
                at += ed_stream_words(a_len[lst[w2]], b_len[lst[w2]]);
                lst_band[w2] = band_of[lst[w2]];
            }
            SVX_HIP(ctx, hipMemcpyAsync(d_ord, lst.data(), lst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            SVX_HIP(ctx, hipMemcpyAsync(d_so, lst_soff.data(), lst.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            SVX_HIP(ctx, hipMemcpyAsync(d_band, lst_band.data(), lst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            a.n = (uint32_t)lst.size();
            if (pass == 0)
                hipLaunchKernelGGL(k_edit_myers<16>, dim3(a.n), dim3(64), lds_bytes<16>(), ctx->stream, a);
            else
                hipLaunchKernelGGL(k_edit_myers<256>, dim3(a.n), dim3(64), lds_bytes<256>(), ctx->stream, a);
            SVX_HIP(ctx, hipGetLastError());
            SVX_HIP(ctx, hipMemcpyAsync(dist, d_dist, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
            { int wrc = svx_wait_blocking(ctx); if (wrc != SVX_OK) return wrc; }
            if (pass == 0) {
                big.clear();
                for (uint32_t i : todo)
                    if (dist[i] == kAlphabet) big.push_back(i);
            }
        }
        for (uint32_t i : big)
            if (dist[i] == kAlphabet) {
                SVX_SET_ERR(ctx, "internal: pair %u rejected by the 256-symbol kernel", i);
                return SVX_E_INVALID;
            }
        again.clear();
        for (uint32_t i : todo)
            if ((dist[i] & kUnproven) && exact_of(i)) again.push_back(i);
        if (again.empty()) break;
        // exact request: widen the band of each unresolved pair.  Its result D' is an upper bound of
        // the distance, so a band of D' is certain to resolve it; 4x the old band is tried first
        // when that is narrower (the bound can be far above the distance)
        for (uint32_t i : again) {
            const uint32_t ub = dist[i] & ~kUnproven;
            const uint32_t k4 = band_of[i] >= 0x10000000u ? 0x7FFFFFFFu : band_of[i] * 4;
            band_of[i] = std::min(ub, k4);
        }
        todo = again;
        big.clear();
    }
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_end(ctx);
    if (rc != SVX_OK) return rc;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (dist[i] & kUnproven) dist[i] = 0xFFFFFFFFu;          // only "> k_max" is known (threshold request)
        else if (!exact_of(i) && dist[i] > kmax_of(i)) dist[i] = 0xFFFFFFFFu;
    }
    return SVX_OK;
}

}  // namespace

extern "C" int svx_edit_distance_batch(svx_ctx* ctx, const uint8_t* seq, uint64_t seq_bytes,
                                       const uint64_t* a_off, const uint32_t* a_len,
                                       const uint64_t* b_off, const uint32_t* b_len, uint32_t n_pairs,
                                       uint32_t k_max, uint32_t* dist) {
    if (!ctx) return SVX_E_INVALID;
    if (n_pairs == 0) return SVX_OK;
    if (!a_off || !a_len || !b_off || !b_len || !dist || (seq_bytes && !seq)) return SVX_E_INVALID;
    EdPlan plan;
    int rc = ed_validate_and_plan(ctx, seq_bytes, a_off, a_len, b_off, b_len, n_pairs, &plan);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    rc = svx_stage_reserve(ctx, svx_take_bytes(seq_bytes ? seq_bytes : 1, 1) + ed_stage_need(n_pairs, plan));
    if (rc != SVX_OK) return rc;
    uint8_t* d_seq = svx_stage_take<uint8_t>(ctx, seq_bytes ? seq_bytes : 1);
    if (seq_bytes) SVX_HIP(ctx, hipMemcpyAsync(d_seq, seq, seq_bytes, hipMemcpyHostToDevice, ctx->stream));
    return ed_run(ctx, d_seq, a_off, a_len, b_off, b_len, n_pairs, k_max, dist, plan);
}

// pool: host bytes staged to HBM by this call, or (d_pool_resident != nullptr) already there
static int hap_distance_impl(svx_ctx* ctx, const uint8_t* pool, const uint8_t* d_pool_resident, bool resident, uint64_t pool_bytes,
                             const svx_hap_piece* pieces, uint32_t n_pairs, uint32_t k_max, uint32_t* dist,
                             const uint32_t* k_of = nullptr) {
    if (!ctx) return SVX_E_INVALID;
    if (n_pairs == 0) return SVX_OK;
    if (!pieces || !dist || (pool_bytes && !(resident ? d_pool_resident : pool))) return SVX_E_INVALID;
    const uint32_t n_strings = 2 * n_pairs;
    std::vector<uint64_t> str_off(n_strings), a_off(n_pairs), b_off(n_pairs);
    std::vector<uint32_t> a_len(n_pairs), b_len(n_pairs);
    uint64_t total = 0;
    for (uint32_t w = 0; w < n_strings; ++w) {
        uint64_t len = 0;
        for (int k = 0; k < 3; ++k) {
            const svx_hap_piece& pc = pieces[(size_t)w * 3 + k];
            if (pc.len && (pc.off > pool_bytes || pc.len > pool_bytes - pc.off)) {
                SVX_SET_ERR(ctx, "pair %u: a piece reads past the byte pool", w / 2);
                return SVX_E_INVALID;
            }
            if (pc.flags & ~(SVX_PIECE_UPPER | SVX_PIECE_REVCOMP)) return SVX_E_INVALID;
            len += (uint64_t)pc.len * pc.repeat;
        }
        if (len >= kUnproven) {
            SVX_SET_ERR(ctx, "pair %u: haplotypes of 2^31 bytes or more are not supported", w / 2);
            return SVX_E_TOO_LARGE;
        }
        str_off[w] = total;
        if (w & 1) { b_off[w / 2] = total; b_len[w / 2] = (uint32_t)len; }
        else { a_off[w / 2] = total; a_len[w / 2] = (uint32_t)len; }
        total += len;
    }
    EdPlan plan;
    int rc = ed_validate_and_plan(ctx, total, a_off.data(), a_len.data(), b_off.data(), b_len.data(), n_pairs, &plan);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    rc = svx_stage_reserve(ctx, (resident ? 0 : svx_take_bytes(pool_bytes ? pool_bytes : 1, 1)) + svx_take_bytes((size_t)n_strings * 3, sizeof(svx_hap_piece)) +
                                    svx_take_bytes(n_strings, 8) + svx_take_bytes(total ? total : 1, 1) + ed_stage_need(n_pairs, plan));
    if (rc != SVX_OK) return rc;
    HapArgs h;
    const uint8_t* d_pool = d_pool_resident;
    if (!resident) {
        uint8_t* staged = svx_stage_take<uint8_t>(ctx, pool_bytes ? pool_bytes : 1);
        if (pool_bytes) SVX_HIP(ctx, hipMemcpyAsync(staged, pool, pool_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_pool = staged;
    }
    svx_hap_piece* d_pc = svx_stage_take<svx_hap_piece>(ctx, (size_t)n_strings * 3);
    uint64_t* d_soff = svx_stage_take<uint64_t>(ctx, n_strings);
    uint8_t* d_str = svx_stage_take<uint8_t>(ctx, total ? total : 1);
    SVX_HIP(ctx, hipMemcpyAsync(d_pc, pieces, (size_t)n_strings * 3 * sizeof(svx_hap_piece), hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_soff, str_off.data(), (size_t)n_strings * 8, hipMemcpyHostToDevice, ctx->stream));
    h.pool = d_pool; h.pieces = d_pc; h.str_off = d_soff; h.out = d_str; h.n_strings = n_strings;
    hipLaunchKernelGGL(k_hap_build, dim3(n_strings), dim3(256), 0, ctx->stream, h);
    SVX_HIP(ctx, hipGetLastError());
    return ed_run(ctx, d_str, a_off.data(), a_len.data(), b_off.data(), b_len.data(), n_pairs, k_max, dist, plan, k_of);
}

extern "C" int svx_haplotype_distance_batch(svx_ctx* ctx, const uint8_t* pool, uint64_t pool_bytes,
                                            const svx_hap_piece* pieces, uint32_t n_pairs, uint32_t k_max,
                                            uint32_t* dist) {
    return hap_distance_impl(ctx, pool, nullptr, false, pool_bytes, pieces, n_pairs, k_max, dist);
}

extern "C" int svx_haplotype_distance_batch_dev(svx_ctx* ctx, const uint8_t* d_pool, uint64_t pool_bytes,
                                                const svx_hap_piece* pieces, uint32_t n_pairs, uint32_t k_max,
                                                uint32_t* dist) {
    return hap_distance_impl(ctx, nullptr, d_pool, true, pool_bytes, pieces, n_pairs, k_max, dist);
}

extern "C" int svx_haplotype_distance_batch_mixed(svx_ctx* ctx, const uint8_t* pool, uint64_t pool_bytes,
                                                  const svx_hap_piece* pieces, uint32_t n_pairs, const uint32_t* k_max,
                                                  uint32_t* dist) {
    if (n_pairs && !k_max) return SVX_E_INVALID;
    return hap_distance_impl(ctx, pool, nullptr, false, pool_bytes, pieces, n_pairs, 0, dist, k_max);
}
