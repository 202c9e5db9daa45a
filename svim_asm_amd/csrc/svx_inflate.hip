// svx_inflate.hip — BGZF members inflated and CRC32-checked on gfx950 ("later parallel/GPU
// inflate").  What it stands in for: htslib's bgzf_read_block under every record the reference reads
// (pysam bam.fetch, SVIM_COLLECT.py:65-68; the inserted sequences of SVIM_intra.py:42) — inflate a member, check
// its CRC32 and ISIZE.  In the product path it is the kernel of the BAM reader's device leg (svx_bam.cpp,
// svx_bam_set_device_inflate): a share of the members under a call's sequence slices, beside the reader's threads.
//
// DEFLATE (RFC 1951) is a serial bit stream per member, so the parallelism is ACROSS members: one LANE per member,
// kActive members per wave (4: below), all members of a call (thousands) in one launch.  Per lane:
//   * bit buffer of 64 bits, refilled 32 bits at a time from the member's compressed bytes in HBM; the next word
//     is requested one refill ahead, so a refill never waits for memory;
//   * canonical Huffman decoding by code length (count / first / index walk, one bit per step — the scheme of
//     zlib's puff.c): no lookup tables to build per block, the per-length counts and the symbols sorted by code
//     sit in LDS, element-major ([entry][lane]) so that lanes reading the same entry hit different banks;
//   * literals and match copies go to the member's own stretch of the output buffer in HBM (a lane reads back
//     its own earlier output for a match: program order per lane), every output byte passes through the running
//     CRC32 (one 256-entry table per workgroup in LDS);
//   * stored, fixed and dynamic blocks; anything malformed (over-subscribed or incomplete code sets — except the
//     one-code distance set zlib allows —, distances beyond the output so far, output beyond ISIZE, input that
//     ends early) ends the lane with a status, never with an access outside the member's input and output.
// Integer / byte work, no MFMA.  Bound by the serial decode chain per lane (LDS and HBM latencies), not by bytes.
#include "svx_internal.h"

namespace {

constexpr int kLL = 288, kDist = 30, kMaxBits = 15;
#ifndef SVX_INFL_LANES
#define SVX_INFL_LANES 16
#endif
constexpr int kLanes = SVX_INFL_LANES;  // members per workgroup (their LDS tables: 2.2 KB each)
#ifndef SVX_INFL_ACTIVE
#define SVX_INFL_ACTIVE 4
#endif
// Lanes of a wave that hold a member.  The lanes of a wave are at different points of their streams, and every distinct
// path is issued for the whole wave: the fewer members share a wave, the shorter each one's decode — and the more waves
// the same number of members in flight needs (110 VGPRs: 16 waves per CU).  7 261 SEQ members, kernel time
// (profiles/r05_inflate_geometry.txt): 32 per wave 58 ms (16 k members in flight on the chip), 4 per wave 49 (16 k),
// 2 per wave 34.5 (8 k), 1 per wave 23 ms up to the 4 096 it holds at once, 48 beyond.  Four per wave keeps the chip's
// capacity of 64 members per CU (4 workgroups of 16 members: 39 KB of LDS each) — a sample's two readers bring 13-14 k.
constexpr int kActive = SVX_INFL_ACTIVE;
static_assert(kActive >= 1 && kActive <= 64 && kLanes % kActive == 0, "a workgroup is kLanes / kActive whole waves");
constexpr int kThreads = kLanes / kActive * 64;
constexpr int kLLBits = 9;    // literal/length codes up to this many bits are resolved by ONE table look-up
constexpr int kDBits = 7;     // ... distance codes

struct InfArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_len;
    const uint32_t* isize;
    const uint32_t* crc;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* status;
    uint32_t n;
};

struct Lds {
    uint32_t crc_table4[4][256];  // slicing-by-4 tables of CRC-32 (reflected 0xEDB88320)
    uint16_t tab_ll[1 << kLLBits][kLanes];  // next kLLBits bits -> symbol << 4 | code length (0: a longer code)
    uint16_t tab_d[1 << kDBits][kLanes];
    uint16_t cnt_ll[kMaxBits + 1][kLanes];
    uint16_t cnt_d[kMaxBits + 1][kLanes];
    uint16_t sym_ll[kLL][kLanes];
    uint16_t sym_d[kDist][kLanes];
    uint8_t lens4[(kLL + kDist + 2) / 2][kLanes];  // code lengths of the block being set up, two per byte
};

__device__ __forceinline__ uint32_t len_get(const Lds& s, int i, int lane) { return (s.lens4[i >> 1][lane] >> (4 * (i & 1))) & 15u; }
__device__ __forceinline__ void len_set(Lds& s, int i, int lane, uint32_t v) {
    const uint32_t sh = 4 * (i & 1);
    s.lens4[i >> 1][lane] = (uint8_t)((s.lens4[i >> 1][lane] & ~(15u << sh)) | ((v & 15u) << sh));
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));

struct Bits {
    const uint8_t* p;
    uint32_t len, pos;   // pos: bytes consumed into buf/next so far
    uint64_t buf;
    uint32_t cnt;
    uint32_t next;       // the word after the ones in buf, requested ahead
    uint32_t next_bytes; // how many real bytes `next` holds (0..4)
};

// The input word at byte position pos, zero-padded behind the member's end — WITHOUT a branch: the refill requests
// it several symbols before it is used, and a load that sits behind control flow is waited for at once (a wait is
// wave-wide: some lane refills at almost every symbol).  Reads the aligned-to-the-end word and shifts; may touch up
// to 3 bytes behind a member's input (the caller pads the buffer).
__device__ __forceinline__ uint32_t load_word(const uint8_t* p, uint32_t pos, uint32_t len, uint32_t* real) {
    const uint32_t q = pos + 4 <= len ? pos : (len >= 4 ? len - 4 : 0u);
    uint32_t w = *reinterpret_cast<const u32_unaligned*>(p + q);
    const uint32_t have = len > q ? (len - q < 4 ? len - q : 4u) : 0u;   // real bytes in w
    w &= have >= 4 ? 0xFFFFFFFFu : ((1u << (8 * have)) - 1u);
    const uint32_t skip = pos - q;                                        // bytes of w that lie before pos
    *real = have > skip ? have - skip : 0u;
    return skip >= 4 ? 0u : w >> (8 * skip);
}

__device__ __forceinline__ void bits_init(Bits& b, const uint8_t* p, uint32_t len) {
    b.p = p; b.len = len; b.pos = 0; b.buf = 0; b.cnt = 0;
    b.next = load_word(p, 0, len, &b.next_bytes);
}

// at least 32 valid (or zero-padded) bits afterwards; `over` counts padding bits handed out beyond the input
__device__ __forceinline__ void bits_refill(Bits& b) {
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.next << b.cnt;
        b.cnt += 32;
        b.pos += 4;
        b.next = load_word(b.p, b.pos, b.len, &b.next_bytes);
    }
}

__device__ __forceinline__ uint32_t bits_get(Bits& b, uint32_t n) {  // n <= 16
    bits_refill(b);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// bits consumed so far; more than 8 * len means the stream ran past its input
__device__ __forceinline__ uint64_t bits_used(const Bits& b) { return (uint64_t)b.pos * 8 - b.cnt; }

// The per-length code counts of the block's two codes, in registers (two 16-bit counts per word) for the whole block:
// the decode walk below is then pure arithmetic plus ONE LDS read for the symbol.
struct Counts {
    uint32_t w[8];  // w[k] = count[2k] | count[2k + 1] << 16
};
template <typename CNT>
__device__ __forceinline__ Counts load_counts(CNT cnt, int lane) {  // cnt[0..15][lane]
    Counts c;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.w[k] = (uint32_t)cnt[2 * k][lane] | ((uint32_t)cnt[2 * k + 1][lane] << 16);
    return c;
}

// puff.c's decode() alone (the code-length code of a dynamic block: 19 symbols, used a few hundred times)
template <typename SYM>
__device__ __forceinline__ int huff_walk(Bits& b, const Counts& c, SYM sym, int lane) {
    bits_refill(b);
    uint32_t bitbuf = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= kMaxBits; ++len) {
        code |= (int)(bitbuf & 1u);
        bitbuf >>= 1;
        const int count = (int)((c.w[len >> 1] >> (16 * (len & 1))) & 0xFFFFu);
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            return sym[index + (code - first)][lane];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// One symbol: the next BITS bits index the block's table (entry = symbol << 4 | length); codes longer than BITS bits
// (entry 0) take puff.c's walk over the code lengths, one bit per step, on the register-resident counts.
template <int BITS, typename TAB, typename SYM>
__device__ __forceinline__ int huff_decode(Bits& b, TAB tab, const Counts& c, SYM sym, int lane) {
    bits_refill(b);
    uint32_t bitbuf = (uint32_t)b.buf;
    const uint32_t e = tab[bitbuf & ((1u << BITS) - 1u)][lane];
    if (e) {
        const uint32_t len = e & 15u;
        b.buf >>= len;
        b.cnt -= len;
        return (int)(e >> 4);
    }
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= kMaxBits; ++len) {
        code |= (int)(bitbuf & 1u);
        bitbuf >>= 1;
        const int count = (int)((c.w[len >> 1] >> (16 * (len & 1))) & 0xFFFFu);
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            return sym[index + (code - first)][lane];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// The look-up table of a code set that huff_construct has laid out (counts + symbols sorted by code): the canonical
// code of the k-th symbol of length l is first(l) + k; its bits arrive LSB first, so the table index is the code
// bit-reversed, repeated for every value of the index bits behind it.
template <int BITS, typename TAB, typename CNT, typename SYM>
__device__ __forceinline__ void huff_table(TAB tab, CNT cnt, SYM sym, int lane) {
    for (int i = 0; i < (1 << BITS); ++i) tab[i][lane] = 0;
    uint32_t code = 0, index = 0;
    for (int l = 1; l <= BITS; ++l) {
        const uint32_t n = cnt[l][lane];
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t rev = __brev(code + k) >> (32 - l);
            const uint16_t e = (uint16_t)(((uint32_t)sym[index + k][lane] << 4) | (uint32_t)l);
            for (uint32_t i = rev; i < (1u << BITS); i += 1u << l) tab[i][lane] = e;
        }
        code = (code + n) << 1;
        index += n;
    }
}

// puff.c's construct(): counts per length, symbols sorted by (length, symbol).  Returns 0 complete, > 0 incomplete,
// < 0 over-subscribed.
template <typename CNT, typename SYM>
__device__ __forceinline__ int huff_construct(CNT cnt, SYM sym, const Lds& lds, int base, int n, int lane) {
    for (int l = 0; l <= kMaxBits; ++l) cnt[l][lane] = 0;
    for (int s = 0; s < n; ++s) cnt[len_get(lds, base + s, lane)][lane] += 1;
    if (cnt[0][lane] == n) return 0;  // no codes: complete, but decoding will fail
    int left = 1;
    for (int l = 1; l <= kMaxBits; ++l) {
        left <<= 1;
        left -= cnt[l][lane];
        if (left < 0) return left;
    }
    uint16_t offs[kMaxBits + 1];
    offs[1] = 0;
    for (int l = 1; l < kMaxBits; ++l) offs[l + 1] = offs[l] + cnt[l][lane];
    for (int s = 0; s < n; ++s) {
        const int l = (int)len_get(lds, base + s, lane);
        if (l) sym[offs[l]++][lane] = (uint16_t)s;
    }
    return left;
}

__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum { ST_OK = 0, ST_BAD_STREAM = 1, ST_SIZE = 2, ST_CRC = 3, ST_INPUT_END = 4 };

#ifdef SVX_INFL_WAVES  // waves per SIMD the register allocator must leave room for (experiments: tools/r05_infl_geom.sh)
#define SVX_INFL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SVX_INFL_WAVES, SVX_INFL_WAVES)))
#else
#define SVX_INFL_OCCUPANCY
#endif
__global__ __launch_bounds__(kThreads) SVX_INFL_OCCUPANCY void k_bgzf_inflate(InfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Lds& s = *reinterpret_cast<Lds*>(lds_raw);
    const int lane = (int)(threadIdx.x >> 6) * kActive + (int)(threadIdx.x & 63u);  // this thread's member slot, if it has one
    const bool holds_member = (threadIdx.x & 63u) < (uint32_t)kActive;
    // CRC-32 (IEEE 802.3, reflected 0xEDB88320) table
    for (int i = threadIdx.x; i < 256; i += kThreads) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        s.crc_table4[0][i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kThreads) {
        uint32_t c = s.crc_table4[0][i];
        for (int t = 1; t < 4; ++t) {
            c = s.crc_table4[0][c & 0xFFu] ^ (c >> 8);
            s.crc_table4[t][i] = c;
        }
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * kLanes + lane;
    if (!holds_member || m >= a.n) return;
    const uint8_t* in = a.in + a.in_off[m];
    const uint32_t in_len = a.in_len[m], isize = a.isize[m];
    uint8_t* out = a.out + a.out_off[m];
    Bits b;
    bits_init(b, in, in_len);
    uint32_t produced = 0, st = ST_OK;
    if (isize > 65536u) {  // no BGZF member is longer: a wrong trailer must not move the output limit
        a.status[m] = ST_SIZE;
        return;
    }
    bool last = false;
    while (!last && st == ST_OK) {
        last = bits_get(b, 1) != 0;
        const uint32_t type = bits_get(b, 2);
        if (type == 0) {  // stored: skip to the byte boundary, LEN / NLEN, raw bytes
            const uint32_t drop = b.cnt & 7u;
            b.buf >>= drop;
            b.cnt -= drop;
            const uint32_t len = bits_get(b, 16), nlen = bits_get(b, 16);
            if ((len ^ 0xFFFFu) != nlen) { st = ST_BAD_STREAM; break; }
            if (produced + len > isize) { st = ST_SIZE; break; }
            for (uint32_t i = 0; i < len; ++i) out[produced++] = (uint8_t)bits_get(b, 8);
            if (bits_used(b) > (uint64_t)in_len * 8) st = ST_INPUT_END;
            continue;
        }
        if (type == 3) { st = ST_BAD_STREAM; break; }
        if (type == 1) {  // fixed codes
            for (int i = 0; i < 144; ++i) len_set(s, i, lane, 8);
            for (int i = 144; i < 256; ++i) len_set(s, i, lane, 9);
            for (int i = 256; i < 280; ++i) len_set(s, i, lane, 7);
            for (int i = 280; i < kLL; ++i) len_set(s, i, lane, 8);
            for (int i = 0; i < kDist; ++i) len_set(s, kLL + i, lane, 5);
            huff_construct(s.cnt_ll, s.sym_ll, s, 0, kLL, lane);
            huff_construct(s.cnt_d, s.sym_d, s, kLL, kDist, lane);
        } else {  // dynamic codes
            const uint32_t nlen = bits_get(b, 5) + 257, ndist = bits_get(b, 5) + 1, ncode = bits_get(b, 4) + 4;
            if (nlen > 286 || ndist > 30) { st = ST_BAD_STREAM; break; }
            for (int i = 0; i < 19; ++i) len_set(s, i, lane, 0);
            for (uint32_t i = 0; i < ncode; ++i) len_set(s, kClOrder[i], lane, bits_get(b, 3));
            // the code-length code borrows the distance tables (19 symbols); its own lengths are not needed any more once
            // it is built, so the literal/length + distance lengths it encodes are decoded into the same array from 0
            if (huff_construct(s.cnt_d, s.sym_d, s, 0, 19, lane) != 0) { st = ST_BAD_STREAM; break; }  // (zlib: must be complete)
            const Counts cl_counts = load_counts(s.cnt_d, lane);
            uint32_t idx = 0;
            while (idx < nlen + ndist && st == ST_OK) {
                const int sym = huff_walk(b, cl_counts, s.sym_d, lane);
                if (sym < 0) { st = ST_BAD_STREAM; break; }
                if (sym < 16) {
                    len_set(s, (int)idx++, lane, (uint32_t)sym);
                } else {
                    uint32_t prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { st = ST_BAD_STREAM; break; }
                        prev = len_get(s, (int)idx - 1, lane);
                        rep = 3 + bits_get(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + bits_get(b, 3);
                    } else {
                        rep = 11 + bits_get(b, 7);
                    }
                    if (idx + rep > nlen + ndist) { st = ST_BAD_STREAM; break; }
                    while (rep--) len_set(s, (int)idx++, lane, prev);
                }
            }
            if (st != ST_OK) break;
            if (len_get(s, 256, lane) == 0) { st = ST_BAD_STREAM; break; }  // no end-of-block code
            // the distance lengths follow the literal/length ones: move them to their own region (from the back: the
            // regions may overlap and the destination lies behind the source), pad both with zeros
            for (int i = (int)ndist - 1; i >= 0; --i) len_set(s, kLL + i, lane, len_get(s, (int)nlen + i, lane));
            for (uint32_t i = nlen; i < (uint32_t)kLL; ++i) len_set(s, (int)i, lane, 0);
            for (uint32_t i = ndist; i < (uint32_t)kDist; ++i) len_set(s, kLL + (int)i, lane, 0);
            // incomplete code sets are allowed only when they consist of ONE code of length 1 (zlib inflate_table)
            int err = huff_construct(s.cnt_ll, s.sym_ll, s, 0, kLL, lane);
            if (err < 0 || (err > 0 && kLL != (int)s.cnt_ll[0][lane] + (int)s.cnt_ll[1][lane])) { st = ST_BAD_STREAM; break; }
            err = huff_construct(s.cnt_d, s.sym_d, s, kLL, kDist, lane);
            if (err < 0 || (err > 0 && kDist != (int)s.cnt_d[0][lane] + (int)s.cnt_d[1][lane])) { st = ST_BAD_STREAM; break; }
        }
        // ---- the block's symbols
        huff_table<kLLBits>(s.tab_ll, s.cnt_ll, s.sym_ll, lane);
        huff_table<kDBits>(s.tab_d, s.cnt_d, s.sym_d, lane);
        const Counts c_ll = load_counts(s.cnt_ll, lane), c_d = load_counts(s.cnt_d, lane);
        for (;;) {
            const int sym = huff_decode<kLLBits>(b, s.tab_ll, c_ll, s.sym_ll, lane);
            if (sym < 0) { st = ST_BAD_STREAM; break; }
            if (sym < 256) {
                if (produced >= isize) { st = ST_SIZE; break; }
#ifdef SVX_EXP_INFL_NOLIT  // ablation: what the literal stores cost
                ++produced;
#else
                out[produced++] = (uint8_t)sym;
#endif
                continue;
            }
            if (sym == 256) break;
            const int li = sym - 257;
            if (li >= 29) { st = ST_BAD_STREAM; break; }
            // length and distance bases / extra-bit counts by arithmetic (RFC 1951 §3.2.5: four codes per power of two
            // for lengths, two for distances) instead of table look-ups in memory
            uint32_t len;
            if (li < 8) {
                len = 3u + (uint32_t)li;
            } else if (li == 28) {
                len = 258u;
            } else {
                const uint32_t e = ((uint32_t)li - 4u) >> 2;
                len = 3u + ((4u + ((uint32_t)li & 3u)) << e) + bits_get(b, e);
            }
            const int ds = huff_decode<kDBits>(b, s.tab_d, c_d, s.sym_d, lane);
            if (ds < 0 || ds >= 30) { st = ST_BAD_STREAM; break; }
            uint32_t dist;
            if (ds < 4) {
                dist = 1u + (uint32_t)ds;
            } else {
                const uint32_t ex = ((uint32_t)ds - 2u) >> 1;
                dist = 1u + ((2u + ((uint32_t)ds & 1u)) << ex);
                if (ex > 8) {  // up to 13 extra bits: two reads keep each within the refill guarantee
                    const uint32_t lo = bits_get(b, 8);
                    dist += lo | (bits_get(b, ex - 8) << 8);
                } else {
                    dist += bits_get(b, ex);
                }
            }
            if (dist > produced) { st = ST_BAD_STREAM; break; }
            if (produced + len > isize) { st = ST_SIZE; break; }
            uint8_t* dst = out + produced;
            const uint8_t* src = dst - dist;
#ifdef SVX_EXP_INFL_NOCOPY  // ablation: what the match copies cost (output and CRC wrong)
            if (false) {
#else
            if (dist >= 8) {
#endif
                // source and destination do not overlap within a word: eight bytes per round trip.  The last word may
                // write up to seven bytes past the match — bytes this lane overwrites with its next symbols, or the
                // padding behind the member's stretch (the caller leaves 8 bytes)
                for (uint32_t i = 0; i < len; i += 8) *reinterpret_cast<u64_unaligned*>(dst + i) = *reinterpret_cast<const u64_unaligned*>(src + i);
            } else {
#ifndef SVX_EXP_INFL_NOCOPY
                for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
#endif
            }
            produced += len;
            if (bits_used(b) > (uint64_t)in_len * 8) { st = ST_INPUT_END; break; }
        }
        if (st == ST_OK && bits_used(b) > (uint64_t)in_len * 8) st = ST_INPUT_END;
    }
    // ---- CRC-32 of the member's bytes: a second pass over the lane's own output, four table look-ups per word
    uint32_t crc = 0xFFFFFFFFu;
    if (st == ST_OK && produced == isize) {
        uint32_t i = 0;
        for (; i + 4 <= isize; i += 4) {
            const uint32_t w = *reinterpret_cast<const u32_unaligned*>(out + i) ^ crc;
            crc = s.crc_table4[3][w & 0xFFu] ^ s.crc_table4[2][(w >> 8) & 0xFFu] ^ s.crc_table4[1][(w >> 16) & 0xFFu] ^ s.crc_table4[0][w >> 24];
        }
        for (; i < isize; ++i) crc = s.crc_table4[0][(crc ^ out[i]) & 0xFFu] ^ (crc >> 8);
    }
    if (st == ST_OK && produced != isize) st = ST_SIZE;
    if (st == ST_OK && (crc ^ 0xFFFFFFFFu) != a.crc[m]) st = ST_CRC;
    a.status[m] = st;
}

// pieces of the inflated members → one compact buffer: piece p = src[src_off[p] .. + len[p]) → dst[dst_off[p] ..]; one
// wave per piece (the packed SEQ bytes of one sequence slice inside one member: tens to thousands of bytes)
__global__ __launch_bounds__(256) void k_gather_ranges(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off,
                                                       const uint32_t* __restrict__ len, const uint64_t* __restrict__ dst_off,
                                                       uint32_t n, uint8_t* __restrict__ dst) {
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n) return;
    const uint8_t* s = src + src_off[p];
    uint8_t* d = dst + dst_off[p];
    const uint32_t l = len[p];
    for (uint32_t i = threadIdx.x & 63u; i < l; i += 64u) d[i] = s[i];
}

}  // namespace

// The two launches the BAM reader's device leg needs (svx_bam_seq_slices with svx_bam_set_device_inflate), on a stream of
// the caller's: hipError_t as int.
int svx_bgzf_inflate_on_stream(void* stream, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                               const uint64_t* d_out_off, uint32_t* d_status) {
    if (n_members == 0) return 0;
    InfArgs a{d_in, d_in_off, d_in_len, d_isize, d_crc, d_out, d_out_off, d_status, n_members};
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bgzf_inflate), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(Lds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3((n_members + kLanes - 1) / kLanes), dim3(kThreads), sizeof(Lds),
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int svx_gather_ranges_on_stream(void* stream, const uint8_t* d_src, const uint64_t* d_src_off, const uint32_t* d_len,
                                const uint64_t* d_dst_off, uint32_t n, uint8_t* d_dst) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gather_ranges, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), d_src, d_src_off, d_len,
                       d_dst_off, n, d_dst);
    return (int)hipGetLastError();
}

// svx_bam.cpp reaches the two launches through pointers (it also builds alone, without this file, for the CPU sanitizer tests)
extern "C" void svx_bam_register_device_kernels(
    int (*)(void*, const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint8_t*,
            const uint64_t*, uint32_t*),
    int (*)(void*, const uint8_t*, const uint64_t*, const uint32_t*, const uint64_t*, uint32_t, uint8_t*));
static const int svx_device_kernels_registered =
    (svx_bam_register_device_kernels(&svx_bgzf_inflate_on_stream, &svx_gather_ranges_on_stream), 0);

extern "C" int svx_bgzf_inflate_dev(svx_ctx* ctx, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                    const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                                    const uint64_t* d_out_off, uint32_t* d_status) {
    if (!ctx) return SVX_E_INVALID;
    if (n_members == 0) return SVX_OK;
    if (!d_in || !d_in_off || !d_in_len || !d_isize || !d_crc || !d_out || !d_out_off || !d_status) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, (hipError_t)svx_bgzf_inflate_on_stream(ctx->stream, d_in, d_in_off, d_in_len, d_isize, d_crc, n_members, d_out, d_out_off,
                                                        d_status));
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    return svx_timing_end(ctx);
}
