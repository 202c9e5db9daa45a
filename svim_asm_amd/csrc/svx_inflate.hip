// svx_inflate.hip — BGZF members inflated and CRC32-checked on gfx950 ("later parallel/GPU
// inflate").  What it stands in for: htslib's bgzf_read_block under every record the reference reads
// (pysam bam.fetch, SVIM_COLLECT.py:65-68; the inserted sequences of SVIM_intra.py:42) — inflate a member, check
// its CRC32 and ISIZE.  In the product path it is the kernel of the BAM reader's device leg (svx_bam.cpp,
// svx_bam_set_device_inflate): the members under a call's sequence slices, beside the reader's threads.
//
// Three forms, the same bytes and statuses (tests/test_gpu_inflate.py runs all of them; svx_bgzf_inflate_set_two_pass):
//   3 (shipped)  k_inflate_wparse + k_inflate_parse + k_inflate_resolve.  A WAVE per member parses the bit stream — its
//                64 lanes decode 64 stretches of it at once and resynchronise (a Huffman-coded stream does: see the kernel)
//                —, stores the literals and writes every match as a token; whatever it does not recognise as plain it
//                leaves to the lane-per-member parse behind it, which is also the judge of every malformed stream; a wave
//                per member then applies the tokens in rounds by dependence depth and takes the CRC-32 (1 KiB a lane,
//                folded with crc32_combine's operators).  7 261 sequence members: 6.9 ms; 28 000: 23 ms.
//   2            the same with the lane-per-member parse for every member (16 members a wave, the lanes in step): 36.7 ms.
//   1            k_bgzf_inflate: one launch, a lane per member that decodes, copies its matches from its own earlier
//                output and takes the CRC (rounds 4-5): 49 ms.  DEFLATE (RFC 1951) is a serial bit stream per member; with
//                a lane per member the parallelism is ACROSS members only and a member's ~45 000 symbols are one chain.
// Common to all: canonical Huffman decoding from LDS tables built per block from the code lengths (one look-up for the
// short codes, the long ones by comparison against per-length code limits or puff.c's walk); stored, fixed and dynamic
// blocks; anything malformed (over-subscribed or incomplete code sets — except the one-code set zlib allows —, distances
// beyond the output so far, output beyond ISIZE, input that ends early) ends the member with a status, never with an access
// outside the member's input and output (and their documented padding: include/svx.h).
// Integer / byte work, no MFMA.  Form 3 is bound by instruction issue (~70 VALU per symbol and pass, two passes and a write
// pass per window), forms 1-2 by a member's serial chain.
#include <atomic>

#include "svx_inflate_dev.h"
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

// ==================================================================================================================
// Two-pass form (round 6).  What bounds k_bgzf_inflate above is a member's own serial chain — ≈ 0.7 us per symbol — and
// every step of it ends in a wait for global memory: the refill word, the literal store, a match's read-back of the
// member's own output.  Here the bit stream is PARSED in one launch and the matches are APPLIED in a second one:
//   k_inflate_parse    one wave per sixteen members, the lanes in step (a state per lane, one block of code per state);
//                      the compressed input of each member travels through a 128-byte LDS ring that all 64 lanes of the
//                      wave top up together every eight steps (one 16-byte load per lane: 64 bytes per member), so a step
//                      reads nothing from global memory; a literal is stored where it belongs; a match is NOT copied —
//                      {destination, length, distance} goes to the member's token list (8 bytes) and the output position
//                      moves on.  Decoding never needs the output, so no step waits for it.
//   k_inflate_resolve  one wave per member: the tokens 64 at a time, a lane each.  A match whose source lies in front of
//                      the batch's first destination is independent of the others of the batch (most are: distances reach
//                      32 KiB back) and all of those are copied at once; the others follow in token order.  Then the CRC-32
//                      of the finished bytes, a lane per 1 KiB piece, folded with the x^(8192 * 2^j) operators of
//                      crc32_combine (constants from the host), and the member's status.
// Statuses and bytes are those of k_bgzf_inflate (the tests run both).
struct TwoPassArgs {
    InfArgs a;
    uint2* tok;               // token lists of all members, one behind the other
    uint32_t first, count;    // the members of this launch: [first, first + count) — member m's token list is slice
                              // m - first of `tok`, kTokStride slots each (a member is 65 536 bytes at most, a match 3 at least)
    uint32_t* n_tok;          // per member: tokens written (k_inflate_parse), or 0xFFFFFFFF: the parse ended the member itself
    uint32_t shift[6][32];    // columns of the operators x^(8192 * 2^j) mod P, j = 0..5 (CRC-32, reflected)
    uint32_t redo_only;       // k_inflate_parse: only the members k_inflate_wparse has left to it (n_tok == kTokPending)
};
constexpr uint32_t kTokStride = SVX_INFLATE_TOK_STRIDE;
constexpr uint32_t kTokFinal = 0xFFFFFFFFu;    // n_tok: the parse has ended the member with its status
constexpr uint32_t kTokPending = 0xFFFFFFFEu;  // n_tok: k_inflate_wparse hands the member to k_inflate_parse

constexpr int kRingBytes = 128;  // per member: 2 refills of 64 bytes
struct ParseLds {
    uint16_t tab_ll[1 << kLLBits][kLanes];
    uint16_t tab_d[1 << kDBits][kLanes];
    uint16_t cnt_ll[kMaxBits + 1][kLanes];
    uint16_t cnt_d[kMaxBits + 1][kLanes];
    uint16_t sym_ll[kLL][kLanes];
    uint16_t sym_d[kDist][kLanes];
    uint8_t lens4[(kLL + kDist + 2) / 2][kLanes];
    uint32_t ring[kRingBytes / 4][kLanes];   // [word][member]: lanes reading the same word index hit different banks
    uint32_t in_lo[kLanes], in_hi[kLanes], in_len[kLanes], ring_hi[kLanes], rd_pos[kLanes];  // what the refill lanes need of a member
};
// (the helpers above take the table struct as `Lds`: the same member names)
__device__ __forceinline__ uint32_t len_get(const ParseLds& s, int i, int lane) { return (s.lens4[i >> 1][lane] >> (4 * (i & 1))) & 15u; }
__device__ __forceinline__ void len_set(ParseLds& s, int i, int lane, uint32_t v) {
    const uint32_t sh = 4 * (i & 1);
    s.lens4[i >> 1][lane] = (uint8_t)((s.lens4[i >> 1][lane] & ~(15u << sh)) | ((v & 15u) << sh));
}
template <typename CNT, typename SYM>
__device__ __forceinline__ int huff_construct_p(CNT cnt, SYM sym, const ParseLds& lds, int base, int n, int lane) {
    for (int l = 0; l <= kMaxBits; ++l) cnt[l][lane] = 0;
    for (int s = 0; s < n; ++s) cnt[len_get(lds, base + s, lane)][lane] += 1;
    if (cnt[0][lane] == n) return 0;
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

// The bit buffer of a member, fed from its LDS ring.  pos: input bytes taken into buf / next; the ring holds the input
// bytes [.., ring_hi), zero behind the member's end.
struct RBits {
    uint64_t buf;
    uint32_t cnt, pos, next, ring_hi, in_len;
    const uint8_t* in;
};
// 64 more bytes into THIS lane's ring by the lane itself (block headers take their bits in long runs; the symbol loop
// is topped up by the whole wave, parse_top_up)
// (out of line, like the walk below and the block header: the symbol loop has to stay small enough for the instruction
//  cache — inlined at every refill site the kernel was 150 KB of code and a step took 1 300 clocks)
// (arguments by value: a reference would pin the caller's bit reader to scratch memory)
__device__ __noinline__ void rb_fill_words(const uint8_t* in, uint32_t ring_hi, uint32_t in_len, ParseLds& s, int lane) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { uint32_t real; w[k] = load_word(in, ring_hi + 4u * k, in_len, &real); }
#pragma unroll
    for (int k = 0; k < 16; ++k) s.ring[((ring_hi >> 2) + k) & (kRingBytes / 4 - 1)][lane] = w[k];
}
__device__ __forceinline__ void rb_fill_self(RBits& b, ParseLds& s, int lane) {
    rb_fill_words(b.in, b.ring_hi, b.in_len, s, lane);
    b.ring_hi += 64;
}
__device__ __forceinline__ void rb_refill(RBits& b, ParseLds& s, int lane) {
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.next << b.cnt;
        b.cnt += 32;
        b.pos += 4;
        if (b.pos + 4 > b.ring_hi) rb_fill_self(b, s, lane);
        b.next = s.ring[(b.pos >> 2) & (kRingBytes / 4 - 1)][lane];
    }
}
__device__ __forceinline__ uint32_t rb_get(RBits& b, ParseLds& s, int lane, uint32_t n) {  // n <= 16
    rb_refill(b, s, lane);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
__device__ __forceinline__ uint64_t rb_used(const RBits& b) { return (uint64_t)b.pos * 8 - b.cnt; }

// puff.c's decode() on the register-resident counts: the code-length code of a dynamic block's header, and the codes
// the look-up tables do not hold (longer than 9 / 7 bits).  `sym`: the sorted symbols, [index][member].
// Returns symbol << 8 | bits used, or -1.
__device__ __noinline__ int rb_walk_cold(uint32_t bitbuf, const Counts c, const uint16_t (*sym)[kLanes], int lane) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= kMaxBits; ++len) {
        code |= (int)(bitbuf & 1u);
        bitbuf >>= 1;
        const int count = (int)((c.w[len >> 1] >> (16 * (len & 1))) & 0xFFFFu);
        if (code - count < first) return ((int)sym[index + (code - first)][lane] << 8) | len;
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}
template <typename SYM>
__device__ __forceinline__ int rb_walk(RBits& b, ParseLds& s, const Counts& c, SYM sym, int lane) {
    rb_refill(b, s, lane);
    const int r = rb_walk_cold((uint32_t)b.buf, c, sym, lane);
    if (r < 0) return -1;
    b.buf >>= (uint32_t)r & 255u;
    b.cnt -= (uint32_t)r & 255u;
    return r >> 8;
}
template <int BITS, typename TAB, typename SYM>
__device__ __forceinline__ int rb_decode(RBits& b, ParseLds& s, TAB tab, const Counts& c, SYM sym, int lane) {
    rb_refill(b, s, lane);
    const uint32_t e = tab[(uint32_t)b.buf & ((1u << BITS) - 1u)][lane];
    if (__builtin_expect(e != 0u, 1)) {
        const uint32_t len = e & 15u;
        b.buf >>= len;
        b.cnt -= len;
        return (int)(e >> 4);
    }
    return rb_walk(b, s, c, sym, lane);  // (refills nothing: 32 bits are there)
}

enum { PS_HEADER = 0, PS_SYM = 1, PS_STORED = 2, PS_DONE = 3 };

// what a member's lane carries through the parse
struct ParseLane {
    RBits b;
    Counts c_ll, c_d;
    uint32_t isize, produced, st, stored_left;
    bool last;
};

// A block header: type, (for dynamic codes) the code lengths, the tables of the block.  Out of line — a few calls per
// member, thousands of instructions.  Returns the lane's next state.
__device__ __noinline__ int parse_header(ParseLane& h, ParseLds& s, const int lane) {
    RBits& b = h.b;
    h.last = rb_get(b, s, lane, 1) != 0;
    const uint32_t type = rb_get(b, s, lane, 2);
    if (type == 0) {  // stored: skip to the byte boundary, LEN / NLEN, raw bytes
        const uint32_t drop = b.cnt & 7u;
        b.buf >>= drop;
        b.cnt -= drop;
        const uint32_t len = rb_get(b, s, lane, 16), nlen = rb_get(b, s, lane, 16);
        if ((len ^ 0xFFFFu) != nlen) { h.st = ST_BAD_STREAM; return PS_DONE; }
        if (h.produced + len > h.isize) { h.st = ST_SIZE; return PS_DONE; }
        h.stored_left = len;
        return PS_STORED;
    }
    if (type == 3) { h.st = ST_BAD_STREAM; return PS_DONE; }
    bool good = true;
    if (type == 1) {  // fixed codes
        for (int i = 0; i < 144; ++i) len_set(s, i, lane, 8);
        for (int i = 144; i < 256; ++i) len_set(s, i, lane, 9);
        for (int i = 256; i < 280; ++i) len_set(s, i, lane, 7);
        for (int i = 280; i < kLL; ++i) len_set(s, i, lane, 8);
        for (int i = 0; i < kDist; ++i) len_set(s, kLL + i, lane, 5);
        huff_construct_p(s.cnt_ll, s.sym_ll, s, 0, kLL, lane);
        huff_construct_p(s.cnt_d, s.sym_d, s, kLL, kDist, lane);
    } else {  // dynamic codes (as in k_bgzf_inflate)
        const uint32_t nlen = rb_get(b, s, lane, 5) + 257, ndist = rb_get(b, s, lane, 5) + 1, ncode = rb_get(b, s, lane, 4) + 4;
        good = nlen <= 286 && ndist <= 30;
        if (good) {
            for (int i = 0; i < 19; ++i) len_set(s, i, lane, 0);
            for (uint32_t i = 0; i < ncode; ++i) len_set(s, kClOrder[i], lane, rb_get(b, s, lane, 3));
            good = huff_construct_p(s.cnt_d, s.sym_d, s, 0, 19, lane) == 0;  // (zlib: must be complete)
        }
        if (good) {
            const Counts cl_counts = load_counts(s.cnt_d, lane);
            uint32_t idx = 0;
            while (good && idx < nlen + ndist) {
                const int sym = rb_walk(b, s, cl_counts, s.sym_d, lane);
                if (sym < 0) { good = false; }
                else if (sym < 16) { len_set(s, (int)idx++, lane, (uint32_t)sym); }
                else {
                    uint32_t prev = 0, rep = 0;
                    if (sym == 16) {
                        if (idx == 0) good = false;
                        else { prev = len_get(s, (int)idx - 1, lane); rep = 3 + rb_get(b, s, lane, 2); }
                    } else if (sym == 17) {
                        rep = 3 + rb_get(b, s, lane, 3);
                    } else {
                        rep = 11 + rb_get(b, s, lane, 7);
                    }
                    if (good && idx + rep > nlen + ndist) good = false;
                    while (good && rep--) len_set(s, (int)idx++, lane, prev);
                }
            }
        }
        if (good) good = len_get(s, 256, lane) != 0;  // no end-of-block code
        if (good) {
            for (int i = (int)ndist - 1; i >= 0; --i) len_set(s, kLL + i, lane, len_get(s, (int)nlen + i, lane));
            for (uint32_t i = nlen; i < (uint32_t)kLL; ++i) len_set(s, (int)i, lane, 0);
            for (uint32_t i = ndist; i < (uint32_t)kDist; ++i) len_set(s, kLL + (int)i, lane, 0);
            // incomplete code sets are allowed only when they consist of ONE code of length 1 (zlib inflate_table)
            int err = huff_construct_p(s.cnt_ll, s.sym_ll, s, 0, kLL, lane);
            if (err < 0 || (err > 0 && kLL != (int)s.cnt_ll[0][lane] + (int)s.cnt_ll[1][lane])) good = false;
            if (good) {
                err = huff_construct_p(s.cnt_d, s.sym_d, s, kLL, kDist, lane);
                if (err < 0 || (err > 0 && kDist != (int)s.cnt_d[0][lane] + (int)s.cnt_d[1][lane])) good = false;
            }
        }
    }
    if (!good) { h.st = ST_BAD_STREAM; return PS_DONE; }
    huff_table<kLLBits>(s.tab_ll, s.cnt_ll, s.sym_ll, lane);
    huff_table<kDBits>(s.tab_d, s.cnt_d, s.sym_d, lane);
    h.c_ll = load_counts(s.cnt_ll, lane);
    h.c_d = load_counts(s.cnt_d, lane);
    return PS_SYM;
}

__global__ __launch_bounds__(64) void k_inflate_parse(TwoPassArgs t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ParseLds& s = *reinterpret_cast<ParseLds*>(lds_raw);
    const InfArgs& a = t.a;
    const int lane = (int)threadIdx.x;
    const uint32_t m = t.first + blockIdx.x * kLanes + (uint32_t)lane;
    const bool mine = lane < kLanes && m < t.first + t.count && (!t.redo_only || t.n_tok[m] == kTokPending);
    ParseLane h;
    RBits& b = h.b;
    b.buf = 0; b.cnt = 0; b.pos = 0; b.next = 0; b.ring_hi = 0; b.in_len = 0; b.in = a.in;
    h.isize = 0; h.produced = 0; h.st = ST_OK; h.stored_left = 0; h.last = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) h.c_ll.w[k] = h.c_d.w[k] = 0;
    uint32_t n_tok = 0;
    uint8_t* out = a.out;
    uint2* tok = t.tok;
    int state = PS_DONE;
    if (mine) {
        b.in = a.in + a.in_off[m];
        b.in_len = a.in_len[m];
        h.isize = a.isize[m];
        out = a.out + a.out_off[m];
        tok = t.tok + (uint64_t)(m - t.first) * kTokStride;
        state = PS_HEADER;
        if (h.isize > 65536u) { h.st = ST_SIZE; state = PS_DONE; }  // no BGZF member is longer
        s.in_lo[lane] = (uint32_t)(uintptr_t)b.in;
        s.in_hi[lane] = (uint32_t)((uintptr_t)b.in >> 32);
        s.in_len[lane] = b.in_len;
        rb_fill_self(b, s, lane);
        rb_fill_self(b, s, lane);
        b.next = s.ring[0][lane];
    } else if (lane < kLanes) {
        s.in_len[lane] = 0; s.in_lo[lane] = 0; s.in_hi[lane] = 0;
    }
    // the wave-wide top-up: lane l serves member l & 15 with bytes [16 * (l >> 4), + 16) of the member's next 64
    const int tm = lane & (kLanes - 1), tq = lane >> 4;
    uint32_t step_no = 0;
    for (;;) {
        if (state == PS_HEADER) {  // (through a copy: `h` itself stays in registers)
            ParseLane at_call = h;
            state = parse_header(at_call, s, lane);
            h = at_call;
        }
        if (__ballot(state != PS_DONE) == 0ull) break;
        // ---- every eight steps: all 64 lanes top the sixteen rings up, 64 bytes per member that has room for them
        if ((++step_no & 7u) == 0u) {
            if (lane < kLanes) { s.ring_hi[lane] = b.ring_hi; s.rd_pos[lane] = state == PS_DONE ? 0xFFFFFFFFu : b.pos; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t hi = s.ring_hi[tm], rd = s.rd_pos[tm];
            const bool want = rd != 0xFFFFFFFFu && hi - rd <= 64u && hi < s.in_len[tm] + 8u;
            if (want) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(((uintptr_t)s.in_hi[tm] << 32) | (uintptr_t)s.in_lo[tm]);
                const uint32_t off = hi + 16u * (uint32_t)tq, len_m = s.in_len[tm];
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { uint32_t real; w[k] = load_word(src, off + 4u * k, len_m, &real); }
#pragma unroll
                for (int k = 0; k < 4; ++k) s.ring[((off >> 2) + k) & (kRingBytes / 4 - 1)][tm] = w[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < kLanes && state != PS_DONE && b.ring_hi - b.pos <= 64u && b.ring_hi < b.in_len + 8u) b.ring_hi += 64;
        }
        if (state == PS_STORED) {
            for (int k = 0; k < 4 && h.stored_left; ++k, --h.stored_left) out[h.produced++] = (uint8_t)rb_get(b, s, lane, 8);
            if (h.stored_left == 0) {
                if (rb_used(b) > (uint64_t)b.in_len * 8) { h.st = ST_INPUT_END; state = PS_DONE; }
                else state = h.last ? PS_DONE : PS_HEADER;
            }
        }
        if (state == PS_SYM) {
            const int sym = rb_decode<kLLBits>(b, s, s.tab_ll, h.c_ll, s.sym_ll, lane);
            if (sym < 256) {
                if (sym < 0) { h.st = ST_BAD_STREAM; state = PS_DONE; }
                else if (h.produced >= h.isize) { h.st = ST_SIZE; state = PS_DONE; }
                else out[h.produced++] = (uint8_t)sym;
            } else if (sym == 256) {
                if (rb_used(b) > (uint64_t)b.in_len * 8) { h.st = ST_INPUT_END; state = PS_DONE; }
                else state = h.last ? PS_DONE : PS_HEADER;
            } else {
                const int li = sym - 257;
                if (li >= 29) { h.st = ST_BAD_STREAM; state = PS_DONE; }
                else {
                    uint32_t len;
                    if (li < 8) {
                        len = 3u + (uint32_t)li;
                    } else if (li == 28) {
                        len = 258u;
                    } else {
                        const uint32_t e = ((uint32_t)li - 4u) >> 2;
                        len = 3u + ((4u + ((uint32_t)li & 3u)) << e) + rb_get(b, s, lane, e);
                    }
                    const int ds = rb_decode<kDBits>(b, s, s.tab_d, h.c_d, s.sym_d, lane);
                    if (ds < 0 || ds >= 30) { h.st = ST_BAD_STREAM; state = PS_DONE; }
                    else {
                        uint32_t dist;
                        if (ds < 4) {
                            dist = 1u + (uint32_t)ds;
                        } else {
                            const uint32_t ex = ((uint32_t)ds - 2u) >> 1;
                            dist = 1u + ((2u + ((uint32_t)ds & 1u)) << ex);
                            if (ex > 8) {  // up to 13 extra bits: two reads keep each within the refill guarantee
                                const uint32_t lo = rb_get(b, s, lane, 8);
                                dist += lo | (rb_get(b, s, lane, ex - 8) << 8);
                            } else {
                                dist += rb_get(b, s, lane, ex);
                            }
                        }
                        if (dist > h.produced) { h.st = ST_BAD_STREAM; state = PS_DONE; }
                        else if (h.produced + len > h.isize) { h.st = ST_SIZE; state = PS_DONE; }
                        else if (rb_used(b) > (uint64_t)b.in_len * 8) { h.st = ST_INPUT_END; state = PS_DONE; }
                        else {
                            tok[n_tok++] = make_uint2(h.produced | (len << 16), dist);  // (produced <= 65535, len <= 258)
                            h.produced += len;
                        }
                    }
                }
            }
        }
    }
    if (mine) {
        uint32_t st = h.st;
        if (st == ST_OK && h.produced != h.isize) st = ST_SIZE;
        a.status[m] = st;
        t.n_tok[m] = st == ST_OK ? n_tok : kTokFinal;
    }
}

// ==================================================================================================================
// Wave-parallel parse (round 6).  k_inflate_parse gives a member ONE lane, and the member's symbols are a chain of
// ~40 000 steps that nothing shortens.  Here a member has a WAVE, and the 64 lanes decode 64 stretches of the member's
// bit stream at once — which they can, because a Huffman-coded stream resynchronises: a decoder started at a wrong bit
// falls onto true symbol boundaries within a few symbols.
//   header   the block's header out of a 1 KiB window staged in LDS; the code-length symbols one after the other (all
//            lanes the same way), everything else across the lanes: code counts by LDS atomics, the canonical order by
//            ballots, the look-up tables (10 bits literal/length, 8 bits distance) a symbol per lane; longer codes are
//            found by comparing the next 15 bits against the left-justified code limit of each length (6 or 7 compares).
//   window   the bits from the header's end to the (guessed) end of the block, cut into up to 64 stretches of >= 256 bits.
//     pass 1   lane i decodes from the first bit of stretch i — a guess — to the first symbol that starts behind the
//              stretch, counting the bytes and the matches it would produce;
//     pass 2+  lane i starts again at the bit where lane i - 1 has ended, if that is not where it had started; until every
//              lane up to the first one that met the end-of-block code starts where its predecessor ended.  Lane 0 starts
//              at the true position, so by induction all of these are the member's true symbols (two passes when every
//              lane's guess resynchronised inside its stretch: the rule, at 600 symbols per stretch);
//     write    prefix sums of the byte and match counts give every lane its place in the output and in the token list;
//              the lanes decode their stretches once more, storing literals and tokens as k_inflate_parse does.
//   The first window of a member is half of it (nothing is known), later ones 1.125 x the block before; a block that
//   outlasts its window goes on in the next window with the same tables.
// Anything unusual — a stored block, a code set that is not complete, an invalid code or distance on the true chain, a
// size that does not match — is NOT judged here: the member is marked kTokPending and k_inflate_parse (launched behind
// this kernel for exactly these members) decodes it from the start and gives it its status.
constexpr int kWLL = 10, kWD = 8;
constexpr uint32_t kMinStretchBits = 256;
constexpr int kMaxSyncPasses = 24;
#ifndef SVX_NEXT_WINDOW_16THS
#define SVX_NEXT_WINDOW_16THS 2
#endif
constexpr uint32_t kNextWindowSixteenths = SVX_NEXT_WINDOW_16THS;  // a block's first window: the block before + this many sixteenths of it
#ifndef SVX_FIRST_WINDOW_PCT
#define SVX_FIRST_WINDOW_PCT 50
#endif
constexpr uint32_t kFirstWindowPercent = SVX_FIRST_WINDOW_PCT;  // how much of a member its first window covers: nothing is known yet, and a window that
// runs past its block's end is work for nothing (100 / 50 / 33 / 25 %: 7.6 / 6.9 / 7.0 / 6.7 ms for 7 000 members written by zlib 1,
// 11.8 / 11.05 / 12.0 / 11.3 by libdeflate 6, 11.5 / 11.2 / 10.7 / 10.9 by zlib 6)
#ifndef SVX_JOIN_BITS
#define SVX_JOIN_BITS 512
#endif
constexpr uint32_t kJoinBits = SVX_JOIN_BITS;        // how far into its stretch a lane looks for the place where it joins its earlier pass
struct WaveLds {
    // next bits -> symbol << 4 | code length (0: a longer code, or none); where the bits hold TWO whole literals:
    // literal 1 << 4 | both lengths, and in the upper half 0x1000 | length 1 << 8 | literal 2 (wave_pair_literals)
    uint32_t tab_ll[1 << kWLL];
    uint16_t tab_d[1 << kWD];
    uint16_t tab_cl[128];
    uint16_t sym_ll[kLL];        // symbols sorted by (code length, symbol)
    uint16_t sym_d[32];
    uint32_t lim_ll[16], lim_d[16], lim_cl[16];  // per length: left-justified code limit << 16 | (first index - first code + 32768)
    uint32_t cnt[16], base[16];
    uint8_t lens[kLL + 32 + 8];
    uint32_t hw[258];            // the header window: 1 KiB of the member's input from the word the header starts in
};

__device__ __forceinline__ void wave_sync() {  // LDS written by one lane, read by another of the same wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t wave_scan_excl(uint32_t v, int lane, uint32_t* total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if (lane >= d) x += y;
    }
    *total = (uint32_t)__shfl((int)x, 63);
    return x - v;
}

// The canonical code of `n` lengths (lens[0..n)), all lanes together: the primary table of PB bits, the sorted symbols and
// the per-length limits.  0: built; 1: over-subscribed, or incomplete in a way zlib does not allow (`any_incomplete`: the
// fixed codes, which leave two distance codes unused).
template <int PB, typename ENTRY>
__device__ __forceinline__ int wave_build_code(WaveLds& s, const uint8_t* lens, int n, int chunks, ENTRY* tab, uint16_t* sorted,
                                               uint32_t* lim, bool must_be_complete, bool any_incomplete, int lane) {
    if (lane < 16) s.cnt[lane] = 0;
    for (int i = lane; i < (1 << PB); i += 64) tab[i] = 0;
    wave_sync();
    for (int c = 0; c < chunks; ++c) {
        const int si = c * 64 + lane;
        const uint32_t l = si < n ? lens[si] : 0u;
        if (l) atomicAdd(&s.cnt[l], 1u);
    }
    wave_sync();
    uint32_t code = 0, index = 0, kraft = 0;
    for (int l = 1; l <= kMaxBits; ++l) {  // (every lane the same)
        const uint32_t c = s.cnt[l];
        kraft += c << (kMaxBits - l);
        if (lane == 0) {
            lim[l] = (((code + c) << (kMaxBits - l)) << 16) | ((index - code + 32768u) & 0xFFFFu);
            s.base[l] = index;
        }
        code = (code + c) << 1;
        index += c;
    }
    if (kraft > (1u << kMaxBits)) return 1;
    if (kraft < (1u << kMaxBits) && !any_incomplete) {
        if (must_be_complete) return 1;
        if (!(index == 0u || (index == 1u && s.cnt[1] == 1u))) return 1;
    }
    wave_sync();
    for (int c = 0; c < chunks; ++c) {
        const int si = c * 64 + lane;
        const uint32_t l = si < n ? lens[si] : 0u;
        uint32_t rank = 0;
        uint64_t todo = __ballot(l != 0u);
        while (todo) {  // one round per distinct length among the chunk's symbols
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t l0 = (uint32_t)__shfl((int)l, leader);
            const uint64_t same = __ballot(l == l0);
            const uint32_t b0 = s.base[l0];
            if (l == l0) rank = b0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) s.base[l0] = b0 + (uint32_t)__popcll(same);
            todo &= ~same;
        }
        if (l) {
            sorted[rank] = (uint16_t)si;
            if (l <= (uint32_t)PB) {
                const uint32_t cd = rank + 32768u - (lim[l] & 0xFFFFu);  // rank - (first index - first code)
                const uint32_t rev = __brev(cd) >> (32 - l);
                const ENTRY e = (ENTRY)(((uint32_t)si << 4) | l);
                for (uint32_t i = rev; i < (1u << PB); i += 1u << l) tab[i] = e;
            }
        }
        wave_sync();
    }
    return 0;
}

// Two literals per look-up: where the kWLL bits of an index hold a literal AND the whole code of a second one, the entry says
// so (packed sequence is literals of 4-5 bits for the most part: libdeflate writes hardly anything else into a SEQ member).
// Every lane takes 16 entries: the new values from the single-symbol table first, all of them, then the stores.
__device__ __forceinline__ void wave_pair_literals(uint32_t* tab, int lane) {
    constexpr int kPer = (1 << kWLL) / 64;
    uint32_t made[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const uint32_t i = (uint32_t)lane + 64u * (uint32_t)q;
        const uint32_t e1 = tab[i];
        uint32_t v = e1;
        const uint32_t l1 = e1 & 15u, s1 = e1 >> 4;
        if (e1 && s1 < 256u && l1 < (uint32_t)kWLL) {
            const uint32_t e2 = tab[i >> l1];  // (the index's upper bits are zeros here, not stream bits: only a code that
            const uint32_t l2 = e2 & 15u, s2 = e2 >> 4;                       //  ends inside the real ones counts)
            if (e2 && s2 < 256u && l1 + l2 <= (uint32_t)kWLL) v = (s1 << 4) | (l1 + l2) | ((0x1000u | (l1 << 8) | s2) << 16);
        }
        made[q] = v;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < kPer; ++q) tab[(uint32_t)lane + 64u * (uint32_t)q] = made[q];
    wave_sync();
}

// bits [bit, bit + n) of the header window, n <= 16
__device__ __forceinline__ uint32_t hw_peek(const WaveLds& s, uint32_t bit, uint32_t n) {
    const uint32_t i = bit >> 5;
    const uint64_t w = (uint64_t)s.hw[i] | ((uint64_t)s.hw[i + 1] << 32);
    return (uint32_t)(w >> (bit & 31u)) & ((1u << n) - 1u);
}

// The block header at bit `pos` of the member: 0 and the tables of the block, `pos` behind the header, `last`; or 1: not
// for this kernel.
__device__ __forceinline__ int wave_header(WaveLds& s, const uint8_t* in, uint32_t in_len, uint32_t* pos_io, uint32_t* last_out, int lane) {
    const uint32_t pos = *pos_io;
    const uint32_t byte0 = (pos >> 5) << 2;
    for (int i = lane; i < 258; i += 64) {
        uint32_t real;
        s.hw[i] = load_word(in, byte0 + 4u * (uint32_t)i, in_len, &real);
    }
    wave_sync();
    uint32_t hb = pos & 31u;
    *last_out = hw_peek(s, hb, 1);
    const uint32_t type = hw_peek(s, hb + 1, 2);
    hb += 3;
    if (type == 0u || type == 3u) return 1;
    uint32_t nlen, ndist;
    if (type == 1u) {
        nlen = 288; ndist = 30;
        for (int i = lane; i < 288; i += 64) s.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
        if (lane < 30) s.lens[288 + lane] = 5;
        wave_sync();
    } else {
        nlen = hw_peek(s, hb, 5) + 257u;
        ndist = hw_peek(s, hb + 5, 5) + 1u;
        const uint32_t ncode = hw_peek(s, hb + 10, 4) + 4u;
        hb += 14;
        if (nlen > 286u || ndist > 30u) return 1;
        if (lane < 19) s.lens[kClOrder[lane]] = (uint8_t)((uint32_t)lane < ncode ? hw_peek(s, hb + 3u * (uint32_t)lane, 3) : 0u);
        hb += 3u * ncode;
        wave_sync();
        if (wave_build_code<7>(s, s.lens, 19, 1, s.tab_cl, s.sym_d, s.lim_cl, true, false, lane)) return 1;
        // the code lengths, one symbol after the other
        const uint32_t total = nlen + ndist;
        uint32_t idx = 0, prev = 0;
        while (idx < total) {
            const uint32_t e = s.tab_cl[hw_peek(s, hb, 7)];
            if (e == 0u) return 1;
            hb += e & 15u;
            const uint32_t sym = e >> 4;
            if (sym < 16u) {
                if (lane == 0) s.lens[idx] = (uint8_t)sym;
                ++idx;
                prev = sym;
            } else {
                uint32_t rep, v = 0;
                if (sym == 16u) {
                    if (idx == 0u) return 1;
                    rep = 3u + hw_peek(s, hb, 2); hb += 2; v = prev;
                } else if (sym == 17u) {
                    rep = 3u + hw_peek(s, hb, 3); hb += 3;
                } else {
                    rep = 11u + hw_peek(s, hb, 7); hb += 7;
                }
                if (idx + rep > total) return 1;
                for (uint32_t j = (uint32_t)lane; j < rep; j += 64u) s.lens[idx + j] = (uint8_t)v;
                idx += rep;
                prev = v;
            }
        }
        wave_sync();
        if (s.lens[256] == 0u) return 1;  // no end-of-block code
    }
    const uint32_t after = byte0 * 8u + hb;
    if (after > in_len * 8u) return 1;
    *pos_io = after;
    const bool fixed = type == 1u;
    if (wave_build_code<kWLL>(s, s.lens, (int)nlen, 5, s.tab_ll, s.sym_ll, s.lim_ll, false, fixed, lane)) return 1;
    wave_pair_literals(s.tab_ll, lane);
    if (wave_build_code<kWD>(s, s.lens + nlen, (int)ndist, 1, s.tab_d, s.sym_d, s.lim_d, false, fixed, lane)) return 1;
    return 0;
}

// A lane's bit reader on the member's input in global memory.  The input comes 16 bytes at a time and the 16 bytes behind
// the ones in use are requested when those are taken up: a load has ~14 symbols to arrive (the lanes of a wave read 64
// different places and the waves of a CU more lines than its L1 holds: most loads come from the L2 or from further away).
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
struct WBits {
    uint64_t buf;
    uint32_t cnt, pos;   // pos: input bytes taken into buf
    uint32_t q0, q1, q2, q3, left;  // the next `left` words of the input, q0 first
    u32x4_a1 ahead;      // the 16 bytes behind q's
};
__device__ __forceinline__ u32x4_a1 wb_load16(const uint8_t* in, uint32_t at, uint32_t in_len) {
    u32x4_a1 v;
    if (__builtin_expect(at + 16u <= in_len, 1)) {
        v = *reinterpret_cast<const u32x4_a1*>(in + at);
    } else {  // the member's last bytes: zero behind them
        uint32_t real;
        v.x = load_word(in, at, in_len, &real);
        v.y = load_word(in, at + 4u, in_len, &real);
        v.z = load_word(in, at + 8u, in_len, &real);
        v.w = load_word(in, at + 12u, in_len, &real);
    }
    return v;
}
__device__ __forceinline__ void wb_init(WBits& b, const uint8_t* in, uint32_t in_len, uint32_t bit) {
    const uint32_t byte0 = (bit >> 5) << 2, sh = bit & 31u;
    const u32x4_a1 first = wb_load16(in, byte0, in_len);
    b.ahead = wb_load16(in, byte0 + 16u, in_len);
    b.buf = (uint64_t)(first.x >> sh);
    b.cnt = 32u - sh;
    b.pos = byte0 + 4u;
    b.q0 = first.y; b.q1 = first.z; b.q2 = first.w; b.q3 = 0;
    b.left = 3;
}
__device__ __forceinline__ void wb_refill(WBits& b, const uint8_t* in, uint32_t in_len) {  // >= 33 bits afterwards
    if (b.cnt <= 32u) {
        b.buf |= (uint64_t)b.q0 << b.cnt;
        b.cnt += 32u;
        b.pos += 4u;
        b.q0 = b.q1; b.q1 = b.q2; b.q2 = b.q3;
        if (--b.left == 0u) {
            b.q0 = b.ahead.x; b.q1 = b.ahead.y; b.q2 = b.ahead.z; b.q3 = b.ahead.w;
            b.left = 4;
            b.ahead = wb_load16(in, b.pos + 16u, in_len);
        }
    }
}
__device__ __forceinline__ uint32_t wb_at(const WBits& b) { return b.pos * 8u - b.cnt; }

// the code at the head of the bits that no table entry resolves: symbol << 4 | length, or 0
template <int PB, int NSYM>
__device__ __forceinline__ uint32_t wave_long_code(uint32_t bits, const uint16_t* sorted, const uint32_t (&lim)[kMaxBits - PB]) {
    const uint32_t code15 = __brev(bits) >> 17;  // the next 15 bits, first bit on top
    uint32_t len = 0, packed = 0;
#pragma unroll
    for (int l = kMaxBits; l > PB; --l) {  // the shortest length whose limit lies above the bits
        const uint32_t p = lim[l - PB - 1];
        if (code15 < (p >> 16)) { len = (uint32_t)l; packed = p; }
    }
    if (len == 0u) return 0u;
    const uint32_t idx = (code15 >> (kMaxBits - len)) + (packed & 0xFFFFu) - 32768u;
    if (idx >= (uint32_t)NSYM) return 0u;
    return ((uint32_t)sorted[idx] << 4) | len;
}

template <int PB, int NSYM>
__device__ __forceinline__ int wave_symbol(WBits& b, const uint16_t* tab, const uint16_t* sorted, const uint32_t (&lim)[kMaxBits - PB]) {
    const uint32_t bits = (uint32_t)b.buf;
    uint32_t e = tab[bits & ((1u << PB) - 1u)];
    if (__builtin_expect(e == 0u, 0)) {
        const uint32_t code15 = __brev(bits) >> 17;  // the next 15 bits, first bit on top
        uint32_t len = 0, packed = 0;
#pragma unroll
        for (int l = kMaxBits; l > PB; --l) {  // the shortest length whose limit lies above the bits
            const uint32_t p = lim[l - PB - 1];
            if (code15 < (p >> 16)) { len = (uint32_t)l; packed = p; }
        }
        if (len == 0u) return -1;
        const uint32_t idx = (code15 >> (kMaxBits - len)) + (packed & 0xFFFFu) - 32768u;
        if (idx >= (uint32_t)NSYM) return -1;
        e = ((uint32_t)sorted[idx] << 4) | len;
    }
    const uint32_t l = e & 15u;
    b.buf >>= l;
    b.cnt -= l;
    return (int)(e >> 4);
}

enum { WF_NONE = 0, WF_EOB = 1, WF_BAD = 2 };

// What a lane keeps of its last pass over its stretch.  `chk_at`: the first symbol start at or behind `chk_limit` (a little
// way into the stretch), with the bytes and matches counted in front of it: a later pass from another start that arrives
// at exactly this bit has joined the old pass — everything behind it is the same, and it stops there.
struct Stretch {
    uint32_t start, limit, chk_limit;
    uint32_t end, flag, n_bytes, n_toks;
    uint32_t chk_at, chk_bytes, chk_toks;
};
constexpr uint32_t kNoCheckpoint = 0xFFFFFFFFu;

// One lane's stretch: the symbols that start in [start, limit).  WRITE: literals to out[o..], matches to tok[..].
template <bool WRITE>
__device__ __forceinline__ void wave_decode(const uint8_t* in, uint32_t in_len, const WaveLds& s, const uint32_t (&lim_ll)[kMaxBits - kWLL],
                                            const uint32_t (&lim_d)[kMaxBits - kWD], bool active, Stretch& st, uint8_t* out, uint32_t o,
                                            uint2* tok, uint32_t o_end = 0) {  // (o_end: the member's ISIZE — the write pass never stores behind it)
    if (!active) return;
    WBits b;
    wb_init(b, in, in_len, st.start);
    uint32_t bytes = 0, toks = 0, fl = WF_NONE, at = st.start;
    bool before_chk = !WRITE;
    for (;;) {
        at = wb_at(b);
        if (at >= st.limit) break;
        if (!WRITE && before_chk && at >= st.chk_limit) {
            before_chk = false;
            if (at == st.chk_at) {  // joined the pass before: its end, its counts from here on
                st.n_bytes = bytes + (st.n_bytes - st.chk_bytes);
                st.n_toks = toks + (st.n_toks - st.chk_toks);
                st.chk_bytes = bytes;
                st.chk_toks = toks;
                return;
            }
            st.chk_at = at;
            st.chk_bytes = bytes;
            st.chk_toks = toks;
        }
        wb_refill(b, in, in_len);
        int sym;
        {
            const uint32_t bits = (uint32_t)b.buf;
            uint32_t e = s.tab_ll[bits & ((1u << kWLL) - 1u)];
            if (__builtin_expect(e == 0u, 0)) {
                e = wave_long_code<kWLL, kLL>(bits, s.sym_ll, lim_ll);
                if (e == 0u) { fl = WF_BAD; break; }
            }
            uint32_t l = e & 15u;
            const uint32_t up = e >> 16;
            if (up) {  // two literals — if the second one still starts inside this stretch (and in front of the checkpoint's place)
                const uint32_t l1 = (up >> 8) & 15u;
                const uint32_t bound = (!WRITE && before_chk) ? st.chk_limit : st.limit;
                if (at + l1 < bound) {
                    if (WRITE) {
                        if (o + 2u > o_end) { fl = WF_BAD; break; }
                        out[o] = (uint8_t)(e >> 4);
                        out[o + 1] = (uint8_t)up;
                    }
                    b.buf >>= l;
                    b.cnt -= l;
                    o += 2;
                    bytes += 2;
                    continue;
                }
                l = l1;
            }
            b.buf >>= l;
            b.cnt -= l;
            sym = (int)((e >> 4) & 0xFFFu);
        }
        if (sym < 256) {
            if (WRITE) {
                if (o >= o_end) { fl = WF_BAD; break; }
                out[o] = (uint8_t)sym;
            }
            ++o;
            ++bytes;
            continue;
        }
        if (sym == 256) { fl = WF_EOB; at = wb_at(b); break; }
        const int li = sym - 257;
        if (li >= 29) { fl = WF_BAD; break; }
        uint32_t len;
        if (li < 8) {
            len = 3u + (uint32_t)li;
        } else if (li == 28) {
            len = 258u;
        } else {
            const uint32_t e = ((uint32_t)li - 4u) >> 2;
            len = 3u + ((4u + ((uint32_t)li & 3u)) << e) + ((uint32_t)b.buf & ((1u << e) - 1u));
            b.buf >>= e;
            b.cnt -= e;
        }
        wb_refill(b, in, in_len);
        const int ds = wave_symbol<kWD, 32>(b, s.tab_d, s.sym_d, lim_d);
        if (ds < 0 || ds >= 30) { fl = WF_BAD; break; }
        uint32_t dist;
        if (ds < 4) {
            dist = 1u + (uint32_t)ds;
        } else {
            const uint32_t ex = ((uint32_t)ds - 2u) >> 1;  // <= 13: there after a 15-bit code (33 bits behind a refill)
            dist = 1u + ((2u + ((uint32_t)ds & 1u)) << ex) + ((uint32_t)b.buf & ((1u << ex) - 1u));
            b.buf >>= ex;
            b.cnt -= ex;
        }
        if (WRITE) {
            if (dist > o || o + len > o_end) { fl = WF_BAD; break; }
            tok[toks] = make_uint2(o | (len << 16), dist);
        }
        o += len;
        bytes += len;
        ++toks;
    }
    if (!WRITE && before_chk) st.chk_at = kNoCheckpoint;  // this pass ended in front of the checkpoint's place
    st.end = at;
    st.flag = fl;
    st.n_bytes = bytes;
    st.n_toks = toks;
}

#ifdef SVX_WPARSE_STATS  // experiments: where a member's wave spends its clocks (tools/r06_wave_stats.sh)
__device__ unsigned long long g_wstats[16];
#define WSTAT_CLOCK() __builtin_readcyclecounter()
#define WSTAT_ADD(i, v) do { if (lane == 0) atomicAdd(&g_wstats[i], (unsigned long long)(v)); } while (0)
#else
#define WSTAT_CLOCK() 0ull
#define WSTAT_ADD(i, v) do { } while (0)
#endif
#ifdef SVX_WPARSE_WAVES  // waves per SIMD the register allocator leaves room for (experiments)
#define SVX_WPARSE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SVX_WPARSE_WAVES, SVX_WPARSE_WAVES)))
#else
#define SVX_WPARSE_OCCUPANCY
#endif
__global__ __launch_bounds__(64) SVX_WPARSE_OCCUPANCY void k_inflate_wparse(TwoPassArgs t) {
    __shared__ WaveLds s;
    const InfArgs& a = t.a;
    const int lane = (int)threadIdx.x;
    if (blockIdx.x >= t.count) return;
    const uint32_t m = t.first + blockIdx.x;
    const uint8_t* in = a.in + a.in_off[m];
    const uint32_t in_len = a.in_len[m], isize = a.isize[m], nbits = in_len * 8u;
    uint8_t* out = a.out + a.out_off[m];
    uint2* tok = t.tok + (uint64_t)blockIdx.x * kTokStride;
    if (isize > 65536u || in_len > 65536u) {  // (no BGZF member is longer; the serial parse says what it is)
        if (lane == 0) t.n_tok[m] = kTokPending;
        return;
    }
    [[maybe_unused]] const unsigned long long c_begin = WSTAT_CLOCK();
    uint32_t pos = 0, produced = 0, n_tok = 0, last = 0, block_start = 0;
    uint32_t guess = max((uint32_t)((uint64_t)nbits * kFirstWindowPercent / 100u), 64u * kMinStretchBits);
    bool need_header = true, give_up = false;
    uint32_t lim_ll[kMaxBits - kWLL], lim_d[kMaxBits - kWD];
    for (;;) {
        if (pos >= nbits) { give_up = true; break; }
        if (need_header) {
            [[maybe_unused]] const unsigned long long c0 = WSTAT_CLOCK();
            if (wave_header(s, in, in_len, &pos, &last, lane)) { give_up = true; break; }
            WSTAT_ADD(0, WSTAT_CLOCK() - c0);
            WSTAT_ADD(1, 1);
#pragma unroll
            for (int l = kWLL + 1; l <= kMaxBits; ++l) lim_ll[l - kWLL - 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.lim_ll[l]);
#pragma unroll
            for (int l = kWD + 1; l <= kMaxBits; ++l) lim_d[l - kWD - 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.lim_d[l]);
            block_start = pos;
            need_header = false;
            if (pos >= nbits) { give_up = true; break; }
        }
        // ---- the window and its stretches
        const uint32_t span = min(nbits - pos, guess), w_end = pos + span;
        const uint32_t stretch = max((span + 63u) / 64u, kMinStretchBits);
        const uint32_t n_lanes = (span + stretch - 1u) / stretch;  // 1..64
        const bool active = (uint32_t)lane < n_lanes;
        Stretch st;
        st.start = pos + (uint32_t)lane * stretch;
        st.limit = (uint32_t)lane + 1u == n_lanes ? w_end : min(st.start + stretch, w_end);
        st.chk_limit = st.start + min(stretch / 2u, kJoinBits);
        st.end = w_end; st.flag = WF_NONE; st.n_bytes = 0; st.n_toks = 0;
        st.chk_at = kNoCheckpoint; st.chk_bytes = 0; st.chk_toks = 0;
        bool decode = active;
        int passes = 0, k = 0;
        WSTAT_ADD(2, 1);
        WSTAT_ADD(9, span);
        for (;;) {
            [[maybe_unused]] const unsigned long long c1 = WSTAT_CLOCK();
            wave_decode<false>(in, in_len, s, lim_ll, lim_d, decode, st, nullptr, 0u, nullptr);
            WSTAT_ADD(passes == 0 ? 3 : 4, WSTAT_CLOCK() - c1);
            WSTAT_ADD(5, 1);
            const uint64_t flagged = __ballot(active && st.flag != WF_NONE);
            k = flagged ? __ffsll((unsigned long long)flagged) - 1 : (int)n_lanes - 1;  // the chain runs to lane k
            const uint32_t before = (uint32_t)__shfl_up((int)st.end, 1);
            const uint32_t true_start = lane == 0 ? pos : before;
            decode = active && lane <= k && true_start != st.start;
            if (__ballot(decode) == 0ull) break;
            if (++passes > kMaxSyncPasses) { give_up = true; break; }
            if (decode) st.start = true_start;
        }
        if (give_up) break;
        const uint32_t flag_k = (uint32_t)__shfl((int)st.flag, k), end_k = (uint32_t)__shfl((int)st.end, k);
        if (flag_k == WF_BAD) { give_up = true; break; }
        const bool in_chain = active && lane <= k;
        uint32_t sum_b, sum_t;
        const uint32_t off_b = wave_scan_excl(in_chain ? st.n_bytes : 0u, lane, &sum_b);
        const uint32_t off_t = wave_scan_excl(in_chain ? st.n_toks : 0u, lane, &sum_t);
        if (produced + sum_b > isize) { give_up = true; break; }
        Stretch wr = st;
        [[maybe_unused]] const unsigned long long c2 = WSTAT_CLOCK();
        wave_decode<true>(in, in_len, s, lim_ll, lim_d, in_chain, wr, out, produced + off_b, tok + n_tok + off_t, isize);
        WSTAT_ADD(6, WSTAT_CLOCK() - c2);
        WSTAT_ADD(10, end_k - pos);
        if (__ballot(in_chain && (wr.flag != st.flag || wr.end != st.end || wr.n_bytes != st.n_bytes || wr.n_toks != st.n_toks)) != 0ull) {
            give_up = true;
            break;
        }
        produced += sum_b;
        n_tok += sum_t;
        pos = end_k;
        if (flag_k == WF_EOB) {
            if (last) break;
            const uint32_t block_bits = pos - block_start;
            guess = max(block_bits + block_bits * kNextWindowSixteenths / 16u, 64u * kMinStretchBits);
            need_header = true;
        } else if (w_end >= nbits) {  // the input ends inside a block
            give_up = true;
            break;
        }
    }
    if (!give_up && (pos > nbits || produced != isize)) give_up = true;
    WSTAT_ADD(7, WSTAT_CLOCK() - c_begin);
    WSTAT_ADD(8, give_up ? 1 : 0);
    WSTAT_ADD(11, n_tok);
    WSTAT_ADD(12, 1);
    if (lane == 0) {
        if (give_up) {
            t.n_tok[m] = kTokPending;
        } else {
            a.status[m] = ST_OK;
            t.n_tok[m] = n_tok;
        }
    }
}

// x -> x * z^(8 * bytes) in the CRC's field, as the 32 columns of the operator: out = XOR of col[i] over the set bits i of x
__device__ __forceinline__ uint32_t crc_apply(const uint32_t (&col)[32], uint32_t x) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r ^= (0u - ((x >> i) & 1u)) & col[i];
    return r;
}

__global__ __launch_bounds__(256) void k_inflate_resolve(TwoPassArgs t) {
    __shared__ uint32_t s_crc[4][256];  // slicing-by-4 tables of CRC-32 (reflected 0xEDB88320)
    __shared__ uint32_t s_shift[6][32];
    __shared__ uint32_t s_dst[4][2][64];  // per wave: where the tokens of the batch in hand start and end
    const InfArgs& a = t.a;
    for (int i = threadIdx.x; i < 256; i += 256) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        s_crc[0][i] = c;
    }
    if (threadIdx.x < 192) s_shift[threadIdx.x >> 5][threadIdx.x & 31] = t.shift[threadIdx.x >> 5][threadIdx.x & 31];
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 256) {
        uint32_t c = s_crc[0][i];
        for (int k = 1; k < 4; ++k) {
            c = s_crc[0][c & 0xFFu] ^ (c >> 8);
            s_crc[k][i] = c;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t in_slice = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (in_slice >= t.count) return;
    const uint32_t m = t.first + in_slice;
    const uint32_t n_tok = t.n_tok[m];
    if (n_tok == kTokFinal) return;  // the parse has written the member's status
    uint8_t* out = a.out + a.out_off[m];
    const uint32_t isize = a.isize[m];
    const uint2* tok = t.tok + (uint64_t)in_slice * kTokStride;
    // ---- the matches, 64 at a time
    auto copy_one = [&](uint32_t dst, uint32_t len, uint32_t dist) {
        const uint8_t* src = out + dst - dist;
        uint8_t* d = out + dst;
        if (dist >= 8) {
            uint32_t i = 0;
            for (; i + 8 <= len; i += 8) {
                const uint64_t v = __builtin_nontemporal_load(reinterpret_cast<const u64_unaligned*>(src + i));
                *reinterpret_cast<u64_unaligned*>(d + i) = v;
            }
            if (i < len) {  // the last 1..7 bytes: exactly (the bytes behind a match are literals the parse has stored)
                uint64_t v = __builtin_nontemporal_load(reinterpret_cast<const u64_unaligned*>(src + i));
                const uint32_t rem = len - i;
                if (rem & 4u) { *reinterpret_cast<u32_unaligned*>(d + i) = (uint32_t)v; v >>= 32; i += 4; }
                if (rem & 2u) { *reinterpret_cast<u16_unaligned*>(d + i) = (uint16_t)v; v >>= 16; i += 2; }
                if (rem & 1u) d[i] = (uint8_t)v;
            }
        } else {  // the source repeats with period `dist`: every byte of it lies in front of the destination
            uint64_t pat = 0;
            for (uint32_t i = 0; i < dist; ++i) pat |= (uint64_t)__builtin_nontemporal_load(src + i) << (8 * i);
            for (uint32_t i = 0; i < len; ++i) d[i] = (uint8_t)(pat >> (8 * (i % dist)));
        }
    };
    // A token depends on the tokens of its batch whose destination its source touches — destinations ascend, so these are
    // a run of the batch, found by two binary searches over the batch's destinations in LDS.  A round copies every token
    // whose run is done; the number of rounds is the depth of the batch's dependences, not their number (matches a few
    // bytes back — the rule in packed sequence at the fast deflate levels — chain two or three deep, not twenty).
    uint32_t* d_lo = s_dst[threadIdx.x >> 6][0];
    uint32_t* d_hi = s_dst[threadIdx.x >> 6][1];
    for (uint32_t base = 0; base < n_tok; base += 64) {
        const bool valid = base + (uint32_t)lane < n_tok;
        uint2 tk = make_uint2(0xFFFFu, 1);  // (behind the last token: destinations that nothing reaches)
        if (valid) tk = tok[base + lane];
        const uint32_t dst = tk.x & 0xFFFFu, len = tk.x >> 16, dist = tk.y;
        d_lo[lane] = dst;
        d_hi[lane] = dst + len;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t src = dst - dist, src_end = src + (len < dist ? len : dist);
        uint32_t first = 0, behind = 0;  // first token whose destination ends behind src; first one that starts at or behind src_end
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            if (d_hi[first + step - 1] <= src) first += step;
            if (d_lo[behind + step - 1] < src_end) behind += step;
        }
        // (both searches stop at 63 when every entry qualifies; the token's own entry never does: its destination lies behind its source's start)
        if (behind > (uint32_t)lane) behind = (uint32_t)lane;
        uint64_t needs = first < behind ? (((behind - first >= 64u) ? ~0ull : ((1ull << (behind - first)) - 1ull)) << first) : 0ull;
        uint64_t done = ~__ballot(valid);
        bool mine_done = !valid;
        while (~done) {
            const bool ready = !mine_done && (needs & ~done) == 0ull;
            if (ready) copy_one(dst, len, dist);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the copies' bytes are in the XCD's L2 (write-through L1) before the next round reads
            done |= __ballot(ready);
            mine_done |= ready;
        }
    }
    // ---- CRC-32: the member's bytes in pieces of 1 KiB, the LAST piece on lane 63 and the odd-sized one first, so that
    // every fold has a right-hand side of 2^j whole pieces; lanes in front of the first piece hold 0 (the field's zero)
    const uint32_t n_pc = (isize + 1023u) / 1024u;   // <= 64
    uint32_t state_crc = 0;
    if (n_pc) {
        const uint32_t first_lane = 64u - n_pc, r = isize - (n_pc - 1u) * 1024u;  // 1..1024 bytes in the first piece
        if ((uint32_t)lane >= first_lane) {
            const uint32_t k = (uint32_t)lane - first_lane;
            const uint32_t lo = k == 0 ? 0u : r + (k - 1u) * 1024u, hi = k == 0 ? r : lo + 1024u;
            uint32_t crc = k == 0 ? 0xFFFFFFFFu : 0u;
            uint32_t i = lo;
            for (; i + 16 <= hi; i += 16) {
                const u32x4_a1 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a1*>(out + i));
                const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t w = ws[q] ^ crc;
                    crc = s_crc[3][w & 0xFFu] ^ s_crc[2][(w >> 8) & 0xFFu] ^ s_crc[1][(w >> 16) & 0xFFu] ^ s_crc[0][w >> 24];
                }
            }
            for (; i < hi; ++i) crc = s_crc[0][(crc ^ __builtin_nontemporal_load(out + i)) & 0xFFu] ^ (crc >> 8);
            state_crc = crc;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {  // lane l (a multiple of 2^(j+1)) takes in the 2^j pieces to its right
            uint32_t col[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) col[i] = s_shift[j][i];
            const uint32_t right = (uint32_t)__shfl_down((int)state_crc, 1 << j);
            const uint32_t folded = crc_apply(col, state_crc) ^ right;
            if ((lane & ((2 << j) - 1)) == 0) state_crc = folded;
        }
    } else {
        state_crc = 0xFFFFFFFFu;
    }
    if (lane == 0) {
        uint32_t st = a.status[m];
        if (st == ST_OK && ((uint32_t)__builtin_amdgcn_readfirstlane((int)state_crc) ^ 0xFFFFFFFFu) != a.crc[m]) st = ST_CRC;
        a.status[m] = st;
    }
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
// the columns of x -> x * z^(8 * 1024 * 2^j) for the CRC-32 state (reflected polynomial 0xEDB88320): column i = the
// state after 1024 * 2^j zero bytes from the state 1 << i; j = 0 by running the bytes, j + 1 by applying j to itself
static const uint32_t (*crc_shift_columns())[32] {
    static uint32_t cols[6][32];
    static const bool made = [] {
        uint32_t table[256];
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        for (int i = 0; i < 32; ++i) {
            uint32_t x = 1u << i;
            for (int b = 0; b < 1024; ++b) x = table[x & 0xFFu] ^ (x >> 8);
            cols[0][i] = x;
        }
        for (int j = 1; j < 6; ++j)
            for (int i = 0; i < 32; ++i) {
                const uint32_t x = cols[j - 1][i];
                uint32_t r = 0;
                for (int q = 0; q < 32; ++q)
                    if ((x >> q) & 1u) r ^= cols[j - 1][q];
                cols[j][i] = r;
            }
        return true;
    }();
    (void)made;
    return cols;
}

#ifdef SVX_WPARSE_STATS
extern "C" int svx_debug_wparse_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wstats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
// which form svx_bgzf_inflate_dev and the BAM reader's device leg launch (process-wide; SVX_INFLATE_KERNEL=1 in the
// environment: the one-pass kernel from the start)
// 1: the one-launch kernel; 2: two passes, a lane per member in the parse; 3: two passes, a wave per member in the parse
static std::atomic<int> g_form{[] {
    const char* e = getenv("SVX_INFLATE_KERNEL");
    return e && e[0] >= '1' && e[0] <= '3' && !e[1] ? e[0] - '0' : 3;
}()};
extern "C" int svx_bgzf_inflate_set_two_pass(int on) {
    const int was = g_form.exchange(on == 0 ? 1 : on == 2 ? 2 : 3);
    return was == 1 ? 0 : was == 2 ? 2 : 1;
}

// members per slice of launches in svx_bgzf_inflate_dev (tests: many slices over few members)
static std::atomic<uint32_t> g_arena_members{SVX_INFLATE_ARENA_MEMBERS};
extern "C" uint32_t svx_bgzf_inflate_set_arena(uint32_t members) {
    return g_arena_members.exchange(members ? members : SVX_INFLATE_ARENA_MEMBERS);
}

// The launches of the two-pass forms: the token lists live in an arena of `tok_members` slices (kTokStride slots of 8 bytes
// each), so the members go out `tok_members` at a time, one slice of launches behind the other on the stream.
static int inflate_two_pass(hipStream_t stream, int form, const InfArgs& a, uint32_t* d_n_tok, void* d_tok, uint32_t tok_members) {
    TwoPassArgs t;
    t.a = a;
    t.n_tok = d_n_tok;
    t.tok = static_cast<uint2*>(d_tok);
    memcpy(t.shift, crc_shift_columns(), sizeof(t.shift));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_parse), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(ParseLds));
    if (e != hipSuccess) return (int)e;
    t.redo_only = form == 3 ? 1u : 0u;
    for (uint32_t first = 0; first < a.n; first += tok_members) {
        t.first = first;
        t.count = a.n - first < tok_members ? a.n - first : tok_members;
        if (form == 3) hipLaunchKernelGGL(k_inflate_wparse, dim3(t.count), dim3(64), 0, stream, t);
        hipLaunchKernelGGL(k_inflate_parse, dim3((t.count + kLanes - 1) / kLanes), dim3(64), sizeof(ParseLds), stream, t);
        hipLaunchKernelGGL(k_inflate_resolve, dim3((t.count + 3) / 4), dim3(256), 0, stream, t);
    }
    return (int)hipGetLastError();
}

int svx_bgzf_inflate_on_stream(void* stream, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                               const uint64_t* d_out_off, uint32_t* d_status, uint32_t* d_n_tok, void* d_tok, uint32_t tok_members) {
    if (n_members == 0) return 0;
    InfArgs a{d_in, d_in_off, d_in_len, d_isize, d_crc, d_out, d_out_off, d_status, n_members};
    const int form = g_form.load();
    if (d_n_tok && d_tok && tok_members && form != 1) return inflate_two_pass(static_cast<hipStream_t>(stream), form, a, d_n_tok, d_tok, tok_members);
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
extern "C" void svx_bam_register_device_kernels(svx_inflate_launch_fn, svx_gather_launch_fn);
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
    // the two-pass forms' token lists: an arena for up to kArenaMembers members at a time out of the context's workspace
    // (175 KB a member: 8 bytes per 3 bytes of output at most); a smaller one when the device has no room for it
    const uint32_t arena_max = g_arena_members.load();
    uint32_t arena = n_members < arena_max ? n_members : arena_max;
    uint32_t* d_n_tok = nullptr;
    void* d_tok = nullptr;
    for (; arena; arena = arena > 512u ? arena / 2u : 0u) {
        if (svx_ws_reserve(ctx, svx_take_bytes(n_members, 4) + svx_take_bytes((size_t)arena * kTokStride, 8)) != SVX_OK) continue;
        d_n_tok = svx_ws_take<uint32_t>(ctx, n_members);
        d_tok = svx_ws_take<uint2>(ctx, (size_t)arena * kTokStride);
        break;
    }  // (no room at all: the one-launch kernel)
    rc = svx_timing_mark(ctx, 1);  // (behind the reservation: a workspace that has to grow is not kernel time)
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, (hipError_t)svx_bgzf_inflate_on_stream(ctx->stream, d_in, d_in_off, d_in_len, d_isize, d_crc, n_members, d_out, d_out_off,
                                                        d_status, d_n_tok, d_tok, d_tok ? arena : 0u));
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    return svx_timing_end(ctx);
}
