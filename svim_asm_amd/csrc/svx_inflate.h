// svx_inflate.h — raw DEFLATE (RFC 1951) decoder of the BAM ingest, host C++.
//
// Why not zlib / libdeflate alone: the members of an assembly-to-reference BAM are SEQ bytes for the most part — two
// 4-bit bases per byte, sixteen byte values that are about equally likely.  zlib's deflate turns them into 4–5-bit
// literals and, in about equal number, matches of three or four bytes at distances anywhere in the window, in no
// predictable order (a 64 KiB member: ≈ 15 000 matches, ≈ 16 000 literals); libdeflate's — what htslib is usually
// built with — into 4-bit literals almost only.  A decoder that takes one symbol per table look-up and branches on
// its kind pays a mispredicted branch per symbol on the first kind of stream.  Here an entry of the primary table
// holds up to three literals, or up to two literals and the length of a short match behind them — whatever fits the
// index bits — and the fast loop runs the same straight-line code for every such entry, two streams side by side
// (pack_entries, Stream::round, Stream::run_pair).  Per member on the GPU box's EPYC 9575F: zlib-written 208 µs with
// zlib, 166 with libdeflate, 88 / 74 here (alone / side by side); libdeflate-written 119, 114, 66 / 57.  And the
// decoder stops and resumes at any output position: an inserted-sequence slice lies somewhere inside a member and only
// the bytes up to its end are wanted (svx_bam_seq_slices), which libdeflate cannot do.
//
// The reference gets here through pysam → htslib → zlib (SVIM_COLLECT.py:68 `bam.fetch`, SVIM_intra.py:40
// `alignment.query_sequence`); the format is the published one, nothing of htslib's is restated.
// zlib stays in the build as the differential oracle of this file (tests/test_inflate.py, tests/native/) and as the
// `SVX_BAM_ZLIB=1` path.
//
// Stream acceptance follows zlib's inflate: over-subscribed code sets are errors; incomplete ones too, except a set
// with a single 1-bit code (and an empty distance set); a literal/length set without the end-of-block code is an
// error; distances beyond the output produced so far are errors (a BGZF member has no preset dictionary).
#ifndef SVX_INFLATE_H_
#define SVX_INFLATE_H_

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace svx_inflate {

constexpr int kLitBits = 12;     // index bits of the primary literal/length table: a block chooses 12 or
constexpr int kLitBitsMin = 11;  // 11 (lit_bits_for)
constexpr int kDistBits = 9;     // … of the primary distance table
constexpr int kPreBits = 7;      // the code-length code has no longer codes
constexpr int kLitSize = (1 << kLitBits) + 288 * (1 << (15 - kLitBitsMin));  // + one sub-table per longer code, at most
constexpr int kDistSize = (1 << kDistBits) + 32 * 64;  // + one 6-bit sub-table per code longer than 9 bits, at most

// Table entries (32 bits).  [5:0] = ALL the stream bits the entry consumes (codes and extra bits: an x86 shift takes
// its count modulo 64, so the entry itself is the shift count).
//
// Distance and code-length tables, and the literal/length table while it is built — [7:6] kind:
//   kind 0  one literal: [15:8] the byte                    (the all-zero entry of a distance table = "no match")
//   kind 1  a value with extra bits: [27:24] bits of the code itself, [31:28] number of extra bits (they follow the
//           code in the stream), [23:8] base — a length, a distance or a code-length symbol
//   kind 2  end of block
//   kind 3  sub-table: [23:8] first entry, [27:24] index bits; index bits 0 = invalid code
constexpr uint32_t kKindLit = 0u << 6, kKindVal = 1u << 6, kKindEnd = 2u << 6, kKindSub = 3u << 6, kKindMask = 3u << 6;
constexpr uint32_t kInvalid = kKindSub;  // consumes nothing, index bits 0
constexpr uint32_t kBitsMask = 63;
//
// The finished literal/length table (pack_entries) — [5:0] != 0: a "fast" entry,
//   [7:6]    number of literals, 0..3; their bytes in [15:8], [23:16], [31:24]
//   [27:24]  with fewer than three literals: the length (3..10, the codes without extra bits) of a match that follows
//            them, 0 = none.  No literals + a length = a short match on its own.
// [5:0] == 0: everything else — [7:6] kind as above, [13:8] the bits it consumes,
//   kind 1  a length with extra bits: [17:14] bits of the code, [20:18] number of extra bits, [29:21] base
//   kind 2  end of block
//   kind 3  sub-table: [17:14] index bits (0 = invalid code), [31:18] first entry
inline uint32_t l_total(uint32_t e) { return (e >> 8) & 63; }
inline uint32_t l_code_bits(uint32_t e) { return (e >> 14) & 15; }   // kind 3: index bits of the sub-table
inline uint32_t l_extra_bits(uint32_t e) { return (e >> 18) & 7; }
inline uint32_t l_base(uint32_t e) { return (e >> 21) & 0x1FF; }
inline uint32_t l_sub_start(uint32_t e) { return e >> 18; }
inline uint32_t l_from_building(uint32_t e) {  // an entry as build_table leaves it → its final form
    const uint32_t bits = e & kBitsMask;
    switch (e & kKindMask) {
        case kKindLit: return bits | (1u << 6) | (((e >> 8) & 0xFF) << 8);
        case kKindVal: return kKindVal | (bits << 8) | (((e >> 24) & 15) << 14) | ((e >> 28) << 18) | (((e >> 8) & 0x1FF) << 21);
        case kKindEnd: return kKindEnd | (bits << 8);
        default: return kKindSub | (((e >> 24) & 15) << 14) | (((e >> 8) & 0xFFFF) << 18);
    }
}

// the distance table of a round without a match (internal linkage: no GOT detour in a shared library)
static const uint32_t kNoMatch[1 << kDistBits] = {0};

struct Tables {
    int lit_bits;  // index bits of this block's primary literal/length table
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
};

inline uint32_t bit_reverse(uint32_t v, int n) {  // the low n (<= 16) bits of v, mirrored
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
    return v >> (16 - n);
}

// symbol payload + the bits of its code (or of the code's tail inside a sub-table) → table entry
inline uint32_t entry_of(uint32_t payload, uint32_t code_bits) {
    if ((payload & kKindMask) == kKindVal) return payload | (code_bits + (payload >> 28)) | (code_bits << 24);
    return payload | code_bits;
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused) → look-up table indexed by the next `tb`
// stream bits (LSB first), sub-tables behind it for longer codes.  payload[s] = entry of symbol s without its bit
// count.  false: over-subscribed, or incomplete in a way zlib refuses.
inline bool build_table(const uint8_t* lens, int n, int tb, uint32_t* tab, int cap, const uint32_t* payload,
                        bool single_code_ok, bool lit_form = false, uint32_t* firsts = nullptr, int* n_firsts = nullptr) {
    uint16_t count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    int max = 15;
    while (max > 0 && !count[max]) --max;
    const int primary = 1 << tb;
    const uint32_t invalid = lit_form ? l_from_building(kInvalid) : kInvalid;
    if (max == 0) {  // no symbols at all: every look-up is an invalid code
        for (int i = 0; i < primary; ++i) tab[i] = invalid;
        return true;
    }
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= count[len];
        if (left < 0) return false;
    }
    if (left > 0) {
        if (!(single_code_ok && max == 1)) return false;
        for (int i = 0; i < primary; ++i) tab[i] = invalid;
    }
    uint16_t offs[17];
    offs[1] = 0;
    for (int len = 1; len <= 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
    uint16_t sorted[320];
    {
        uint16_t at[17];
        memcpy(at, offs, sizeof(at));
        for (int s = 0; s < n; ++s)
            if (lens[s]) sorted[at[lens[s]]++] = (uint16_t)s;
    }
    uint32_t code = 0;
    int idx = 0;
    const int direct = max < tb ? max : tb;
    for (int len = 1; len <= direct; ++len) {
        for (int k = 0; k < count[len]; ++k, ++code) {
            uint32_t e = entry_of(payload[sorted[idx++]], (uint32_t)len);
            if (lit_form) e = l_from_building(e);
            const uint32_t rev = bit_reverse(code, len);
            if (firsts) firsts[(*n_firsts)++] = rev | ((uint32_t)len << 16);
            for (uint32_t i = rev; i < (uint32_t)primary; i += 1u << len) tab[i] = e;
        }
        code <<= 1;
    }
    if (max <= tb) return true;
    // longer codes: in canonical order the codes that share their first tb bits are neighbours and grow in length,
    // so a group's last member fixes the size of its sub-table
    int next_free = primary;
    int len = tb + 1;
    int in_len = 0;  // codes of length `len` already placed
    while (len <= max) {
        if (in_len == count[len]) {
            ++len;
            code <<= 1;
            in_len = 0;
            continue;
        }
        const uint32_t prefix = code >> (len - tb);
        // look ahead to the end of the group
        int sb;
        {
            uint32_t c = code;
            int l = len, done = in_len, last = len;
            for (;;) {
                if (done == count[l]) {
                    if (l == max) break;
                    ++l;
                    c <<= 1;
                    done = 0;
                    continue;
                }
                if ((c >> (l - tb)) != prefix) break;
                last = l;
                ++c;
                ++done;
            }
            sb = last - tb;
        }
        const int start = next_free;
        next_free += 1 << sb;
        if (next_free > cap) return false;
        for (int i = start; i < next_free; ++i) tab[i] = invalid;
        const uint32_t sub = kKindSub | ((uint32_t)start << 8) | ((uint32_t)sb << 24) | (uint32_t)tb;
        tab[bit_reverse(prefix, tb)] = lit_form ? l_from_building(sub) : sub;
        for (;;) {  // place the group's codes
            if (in_len == count[len]) {
                if (len == max) { ++len; break; }
                ++len;
                code <<= 1;
                in_len = 0;
                continue;
            }
            if ((code >> (len - tb)) != prefix) break;
            const int rest = len - tb;
            uint32_t e = entry_of(payload[sorted[idx++]], (uint32_t)rest);
            if (lit_form) e = l_from_building(e);
            for (uint32_t i = bit_reverse(code & ((1u << rest) - 1), rest); i < (1u << sb); i += 1u << rest)
                tab[start + i] = e;
            ++code;
            ++in_len;
        }
    }
    return true;
}

// Second pass over the literal/length table (built in its final form, build_table(lit_form)): a primary entry takes
// along what follows its first symbol as far as the index bits reach.  What the SEQ members of a BAM are made of costs a
// one-symbol-per-look-up decoder a mispredicted branch per symbol: zlib writes them as 4–5-bit literals and matches of
// three to five bytes in about equal numbers, in no predictable order; libdeflate (what htslib is usually built
// with) as 4-bit literals almost only.  So an entry holds up to three literals, or up to two literals and the length
// code of a short match behind them; the fast loop then runs the same straight-line code for every such entry
// (Stream::round).
inline void pack_entries(uint32_t* tab, int tb, const uint32_t* firsts, int n_firsts) {
    auto short_length = [](uint32_t e) {  // a length code without extra bits, length <= 10
        return !(e & kBitsMask) && (e & kKindMask) == kKindVal && l_extra_bits(e) == 0 && l_base(e) <= 10;
    };
    // What can follow a first literal of L bits depends only on the tb - L index bits behind it, not on the literal:
    // the continuations are worked out once per budget b = tb - L (2^b of them, kept at cont[2^b ...]) — from the
    // table as build_table left it, one symbol per entry — and every entry that starts with a literal then becomes
    // its literal plus the continuation of its remaining bits.  A continuation: [5:0] bits, [7:6] further literals
    // (0..2), [15:8] the first of them, [23:16] the second — or the length of a short match when fewer than two
    // literals came before it.
    uint32_t cont[1 << kLitBits];
    uint32_t budgets = 0;
    for (int k = 0; k < n_firsts; ++k)
        if (tab[firsts[k] & 0xFFFF] & kBitsMask) budgets |= 1u << ((uint32_t)tb - (firsts[k] >> 16));
    for (uint32_t budget = 0; budget < (uint32_t)tb; ++budget) {
        if (!(budgets >> budget & 1)) continue;
        for (uint32_t r = 0; r < (1u << budget); ++r) {
            uint32_t bits = 0, cnt = 0, a = 0, b2 = 0;
            uint32_t e = tab[r];
            if ((e & kBitsMask) && (e & kBitsMask) <= budget) {
                cnt = 1;
                a = (e >> 8) & 0xFF;
                bits = e & kBitsMask;
                e = tab[r >> bits];
                if ((e & kBitsMask) && bits + (e & kBitsMask) <= budget) {
                    cnt = 2;
                    b2 = (e >> 8) & 0xFF;
                    bits += e & kBitsMask;
                } else if (short_length(e) && bits + l_total(e) <= budget) {
                    b2 = l_base(e);
                    bits += l_total(e);
                }
            } else if (short_length(e) && l_total(e) <= budget) {
                b2 = l_base(e);
                bits = l_total(e);
            }
            cont[(1u << budget) + r] = bits | (cnt << 6) | (a << 8) | (b2 << 16);
        }
    }
    for (int k = 0; k < n_firsts; ++k) {  // the symbols whose codes index the table directly, each with its replicas
        const uint32_t rev = firsts[k] & 0xFFFF, l1 = firsts[k] >> 16;
        const uint32_t e1 = tab[rev];
        if (e1 & kBitsMask) {  // a literal: bits | 1 << 6 | byte << 8
            const uint32_t budget = (uint32_t)tb - l1;
            const uint32_t* c = cont + (1u << budget);
            for (uint32_t r = 0; r < (1u << budget); ++r) tab[rev | (r << l1)] = e1 + (c[r] & 0xFF) + ((c[r] >> 8) << 16);
        } else if (short_length(e1)) {
            const uint32_t e = l_total(e1) | (l_base(e1) << 24);
            for (uint32_t i = rev; i < (1u << tb); i += 1u << l1) tab[i] = e;
        }
    }
}

// Index bits for a block with these literal/length code lengths: 12 when it is literals for the most part and they are
// 4-bit codes (three of them then share an entry, and the fast loop takes runs of such entries without the distance
// half: what libdeflate makes of SEQ bytes) — that is, twelve or more literals have codes of at most 4 bits and no
// length code is shorter than 6 bits (a match less than every 30th symbol or so); 11 otherwise (half the table to
// build and to pack; zlib's mix of 4-5-bit literals and short matches gains nothing from the twelfth bit, and a
// "does a match follow" branch would mispredict on it).
inline int lit_bits_for(const uint8_t* lens) {
    int short_literals = 0;
    for (int s = 0; s < 256; ++s) short_literals += lens[s] != 0 && lens[s] <= 4;
    if (short_literals < 12) return kLitBitsMin;
    for (int s = 257; s < 286; ++s)
        if (lens[s] != 0 && lens[s] < 6) return kLitBitsMin;
    return kLitBits;
}

struct SymbolPayloads {
    uint32_t lit[288];
    uint32_t dist[32];
    uint32_t pre[19];
    SymbolPayloads() {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                           67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lextra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                           1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (uint32_t s = 0; s < 256; ++s) lit[s] = kKindLit | (s << 8);
        lit[256] = kKindEnd;
        for (int s = 257; s < 286; ++s) lit[s] = kKindVal | ((uint32_t)lbase[s - 257] << 8) | ((uint32_t)lextra[s - 257] << 28);
        lit[286] = lit[287] = kInvalid;  // in the fixed code, never valid in a stream
        for (int s = 0; s < 30; ++s) dist[s] = kKindVal | ((uint32_t)dbase[s] << 8) | ((uint32_t)dextra[s] << 28);
        dist[30] = dist[31] = kInvalid;
        for (uint32_t s = 0; s < 19; ++s) pre[s] = kKindVal | (s << 8);
    }
};
inline const SymbolPayloads& payloads() {
    static const SymbolPayloads p;
    return p;
}

inline const Tables& fixed_tables() {
    static const Tables t = [] {
        Tables f;
        uint8_t lens[288];
        for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        f.lit_bits = kLitBitsMin;
        uint32_t firsts[288];
        int n_firsts = 0;
        build_table(lens, 288, f.lit_bits, f.lit, kLitSize, payloads().lit, false, true, firsts, &n_firsts);
        pack_entries(f.lit, f.lit_bits, firsts, n_firsts);
        for (int s = 0; s < 32; ++s) lens[s] = 5;
        build_table(lens, 32, kDistBits, f.dist, kDistSize, payloads().dist, false);
        return f;
    }();
    return t;
}

inline uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little-endian host (x86-64)
}

class Stream {
  public:
    // Start a raw DEFLATE stream.  The output buffer is handed over with every run() so that it may move between
    // calls as long as its content is kept.
    void begin(const uint8_t* in, size_t in_len) {
        in_ = in;
        in_n_ = in_len;
        ip_ = 0;
        bitbuf_ = 0;
        bitcnt_ = 0;
        op_ = 0;
        phase_ = kHeader;
        final_ = false;
        cur_ = nullptr;
    }
    size_t produced() const { return op_; }
    bool finished() const { return phase_ == kDone; }

    // Decode until at least `stop` bytes are out (to_end: until the final block has ended).  `cap` bounds the output:
    // a stream that yields more is an error.  May go past `stop` by up to one match; never past `cap`.
    bool run(uint8_t* out, size_t cap, size_t stop, bool to_end) {
        aim(out, cap, stop, to_end);
        for (;;) {
            const Ready r = prepare();
            if (r != kReadyFast) return r == kFinished;
            if (!decode_fast()) return fail();
        }
    }
    // Two independent streams side by side.  A round of the fast loop is a chain of dependent look-ups and shifts that
    // leaves most of a core's issue slots empty; the rounds of a second stream fill them (decode_fast_pair).  Same
    // arguments and results as two run() calls.
    static void run_pair(Stream& a, uint8_t* out_a, size_t cap_a, size_t stop_a, bool end_a, bool* ok_a,
                         Stream& b, uint8_t* out_b, size_t cap_b, size_t stop_b, bool end_b, bool* ok_b) {
        a.aim(out_a, cap_a, stop_a, end_a);
        b.aim(out_b, cap_b, stop_b, end_b);
        for (;;) {
            const Ready ra = a.prepare(), rb = b.prepare();
            if (ra == kReadyFast && rb == kReadyFast) {
                decode_fast_pair(a, b);
            } else if (ra == kReadyFast) {
                if (!a.decode_fast()) a.fail();
            } else if (rb == kReadyFast) {
                if (!b.decode_fast()) b.fail();
            } else {
                *ok_a = ra == kFinished;
                *ok_b = rb == kFinished;
                return;
            }
        }
    }

  private:
    enum Phase { kHeader, kBlock, kStored, kDone, kFailed };
    enum Ready { kReadyFast, kFinished, kError };

    void aim(uint8_t* out, size_t cap, size_t stop, bool to_end) {
        out_ = out;
        cap_ = cap;
        to_end_ = to_end;
        stop_ = stop > cap ? cap : stop;
        target_ = to_end ? (size_t)-1 : stop_;
    }
    bool fast_possible() const {
        return in_n_ >= 16 && cap_ >= 320 && ip_ <= in_n_ - 16 && op_ <= cap_ - 320 && op_ < target_;
    }
    // Everything that is not the fast loop: block headers, stored blocks, the checked loop near the ends of the
    // buffers.  Returns when the fast loop can run, when the aim is reached, or on an error.
    Ready prepare() {
        if (phase_ == kFailed) return kError;
        for (;;) {
            if (phase_ == kDone) {
                if (op_ < stop_) { fail(); return kError; }  // the stream ended before the bytes asked for
                return kFinished;
            }
            if (!to_end_ && op_ >= stop_) return kFinished;
            if (phase_ == kHeader) {
                if (!block_header()) { fail(); return kError; }
                continue;
            }
            if (phase_ == kStored) {
                if (stored_left_ > in_n_ - ip_ || stored_left_ > cap_ - op_) { fail(); return kError; }
                memcpy(out_ + op_, in_ + ip_, stored_left_);
                ip_ += stored_left_;
                op_ += stored_left_;
                phase_ = final_ ? kDone : kHeader;
                continue;
            }
            if (fast_possible()) return kReadyFast;
            if (!decode_careful(target_)) { fail(); return kError; }
        }
    }

    bool fail() {
        phase_ = kFailed;
        return false;
    }
    void fill() {
        while (bitcnt_ < 56 && ip_ < in_n_) {  // never beyond 63 bits: the fast loop shifts by the count
            bitbuf_ |= (uint64_t)in_[ip_++] << bitcnt_;
            bitcnt_ += 8;
        }
    }
    bool take(uint32_t n, uint32_t* v) {  // n <= 16
        if (bitcnt_ < n) {
            fill();
            if (bitcnt_ < n) return false;
        }
        *v = (uint32_t)(bitbuf_ & ((1u << n) - 1));
        bitbuf_ >>= n;
        bitcnt_ -= n;
        return true;
    }

    bool block_header() {
        uint32_t v;
        if (!take(3, &v)) return false;
        final_ = v & 1;
        const uint32_t type = v >> 1;
        if (type == 0) {
            // back to a byte boundary: whole bytes still in the bit buffer go back to the input
            const uint32_t drop = bitcnt_ & 7;
            bitbuf_ >>= drop;
            bitcnt_ -= drop;
            ip_ -= bitcnt_ >> 3;
            bitbuf_ = 0;
            bitcnt_ = 0;
            if (in_n_ - ip_ < 4) return false;
            const uint32_t len = in_[ip_] | ((uint32_t)in_[ip_ + 1] << 8);
            const uint32_t nlen = in_[ip_ + 2] | ((uint32_t)in_[ip_ + 3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            ip_ += 4;
            stored_left_ = len;
            phase_ = kStored;
            return true;
        }
        if (type == 1) {
            cur_ = &fixed_tables();
            phase_ = kBlock;
            return true;
        }
        if (type != 2) return false;
        uint32_t hlit, hdist, hclen;
        if (!take(5, &hlit) || !take(5, &hdist) || !take(4, &hclen)) return false;
        const int nlit = (int)hlit + 257, ndist = (int)hdist + 1, ncl = (int)hclen + 4;
        if (nlit > 286 || ndist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < ncl; ++i) {
            if (!take(3, &v)) return false;
            cl[order[i]] = (uint8_t)v;
        }
        uint32_t pre[1 << kPreBits];
        if (!build_table(cl, 19, kPreBits, pre, 1 << kPreBits, payloads().pre, false)) return false;
        uint8_t lens[288 + 32];
        const int total = nlit + ndist;
        int i = 0;
        while (i < total) {
            fill();
            const uint32_t e = pre[bitbuf_ & ((1u << kPreBits) - 1)];
            const uint32_t nb = e & kBitsMask;
            if ((e & kKindMask) != kKindVal || nb > bitcnt_) return false;
            bitbuf_ >>= nb;
            bitcnt_ -= nb;
            const uint32_t sym = (e >> 8) & 0xFF;
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
                continue;
            }
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (i == 0 || !take(2, &rep)) return false;
                rep += 3;
                val = lens[i - 1];
            } else if (sym == 17) {
                if (!take(3, &rep)) return false;
                rep += 3;
            } else {
                if (!take(7, &rep)) return false;
                rep += 11;
            }
            if (i + (int)rep > total) return false;
            memset(lens + i, (int)val, rep);
            i += (int)rep;
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        uint8_t padded[288];
        memcpy(padded, lens, nlit);
        memset(padded + nlit, 0, 288 - nlit);
        own_.lit_bits = lit_bits_for(padded);
        uint32_t firsts[288];
        int n_firsts = 0;
        if (!build_table(padded, 288, own_.lit_bits, own_.lit, kLitSize, payloads().lit, true, true, firsts, &n_firsts)) return false;
        pack_entries(own_.lit, own_.lit_bits, firsts, n_firsts);
        uint8_t dl[32] = {0};
        memcpy(dl, lens + nlit, ndist);
        if (!build_table(dl, 32, kDistBits, own_.dist, kDistSize, payloads().dist, true)) return false;
        cur_ = &own_;
        phase_ = kBlock;
        return true;
    }

    // The distance half of a match whose length is known; careful mode: every step checks what is left.
    bool match_checked(uint32_t len) {
        fill();
        uint32_t d = cur_->dist[bitbuf_ & ((1u << kDistBits) - 1)];
        if ((d & kKindMask) == kKindSub) {
            const uint32_t sb = (d >> 24) & 15;
            if (sb == 0 || (uint32_t)kDistBits > bitcnt_) return false;
            bitbuf_ >>= kDistBits;
            bitcnt_ -= kDistBits;
            d = cur_->dist[((d >> 8) & 0xFFFF) + (bitbuf_ & ((1u << sb) - 1))];
        }
        if ((d & kKindMask) != kKindVal) return false;
        const uint32_t all = d & kBitsMask;
        if (all > bitcnt_) return false;
        const size_t dist = ((d >> 8) & 0xFFFF) + (size_t)((bitbuf_ >> ((d >> 24) & 15)) & ((1u << (d >> 28)) - 1));
        bitbuf_ >>= all;
        bitcnt_ -= all;
        if (dist > op_ || len > cap_ - op_) return false;
        uint8_t* dst = out_ + op_;
        const uint8_t* src = dst - dist;
        if (dist == 1) memset(dst, src[0], len);
        else for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
        op_ += len;
        return true;
    }

    // Fast loop: while 16 input bytes and 320 output bytes are in hand nothing inside needs a bounds check.  One
    // literal/length entry per round.  The round is bound by the chain look-up → shift → look-up → shift through the
    // bit buffer, so everything else is kept off that chain: an entry's low six bits are the whole shift count, the
    // extra bits are read beside the shift, and for an entry of kind 0 — literals, a short match, or both — there is
    // no branch at all: the distance half always runs, against a table of zero entries when no match follows
    // (sixteen bytes are then copied onto themselves).
    struct Lane {
        const uint8_t* in;
        uint8_t* out;
        const uint32_t* lit;
        const uint32_t* dtab;
        size_t ip, op, in_last, out_last, stop;
        uint64_t bb;
        uint32_t bc, lit_bits, lit_mask;
    };
    enum Round { kGoOn, kBlockEnded, kBad };
    void load(Lane& l) const {
        l.in = in_;
        l.out = out_;
        l.lit = cur_->lit;
        l.dtab = cur_->dist;
        l.ip = ip_;
        l.op = op_;
        l.in_last = in_n_ - 16;
        l.out_last = cap_ - 320;
        l.stop = target_;
        l.bb = bitbuf_;
        l.bc = bitcnt_;
        l.lit_bits = (uint32_t)cur_->lit_bits;
        l.lit_mask = (1u << cur_->lit_bits) - 1;
    }
    void store(const Lane& l, Round r) {
        ip_ = l.ip;
        op_ = l.op;
        bitcnt_ = l.bc;
        bitbuf_ = l.bb & ((1ull << l.bc) - 1);  // the refill leaves stream bytes above bit `bc`: the buffer is exact again
        if (r == kBlockEnded) phase_ = final_ ? kDone : kHeader;
        if (r == kBad) fail();
    }
    static bool in_hand(const Lane& l) { return l.ip <= l.in_last && l.op <= l.out_last && l.op < l.stop; }
    static inline __attribute__((always_inline)) Round round(Lane& l) {
        constexpr uint32_t kDMask = (1u << kDistBits) - 1;
        const uint32_t kMask = l.lit_mask;
        uint64_t bb = l.bb | (load64(l.in + l.ip) << l.bc);  // at least 56 bits: 15 + 5 for a length, 15 + 13 for a distance
        uint32_t bc = l.bc | 56;
        l.ip += (63 - l.bc) >> 3;
        uint32_t e = l.lit[bb & kMask];
        uint32_t len;
        size_t op = l.op;
        if (__builtin_expect(e & kBitsMask, 1)) {
            const uint32_t w = e >> 8;
            memcpy(l.out + op, &w, 4);
            const uint32_t cnt = (e >> 6) & 3;
            op += cnt;
            bb >>= e & kBitsMask;
            bc -= e & kBitsMask;
            len = cnt == 3 ? 0 : (e >> 24) & 15;
            // (the order of the tests matters: "three literals" is rare and "a block of literals" constant within a block
            //  in zlib's mix, where "no match follows" alone would be a coin toss for the branch predictor)
            if (cnt == 3 || (l.lit_bits == (uint32_t)kLitBits && len == 0)) {
                if (l.lit_bits == (uint32_t)kLitBits) {
                    for (int k = 0; k < 3; ++k) {
                        e = l.lit[bb & kMask];
                        const uint32_t c2 = (e >> 6) & 3;
                        if (!(e & kBitsMask) || (c2 != 3 && ((e >> 24) & 15))) break;  // not literals only
                        const uint32_t w2 = e >> 8;
                        memcpy(l.out + op, &w2, 4);
                        op += c2;
                        bb >>= e & kBitsMask;
                        bc -= e & kBitsMask;
                    }
                }
                l.op = op;
                l.bb = bb;
                l.bc = bc;
                return kGoOn;
            }
        } else {
            if ((e & kKindMask) == kKindSub) {
                const uint32_t sb = l_code_bits(e);
                if (sb == 0) return kBad;
                bb >>= l.lit_bits;
                bc -= l.lit_bits;
                e = l.lit[l_sub_start(e) + (bb & ((1u << sb) - 1))];
                if (e & kBitsMask) {  // sub-table entries hold one symbol: here a literal
                    l.out[op] = (uint8_t)(e >> 8);
                    l.op = op + 1;
                    l.bb = bb >> (e & kBitsMask);
                    l.bc = bc - (e & kBitsMask);
                    return kGoOn;
                }
                if ((e & kKindMask) == kKindSub) return kBad;
            }
            if ((e & kKindMask) == kKindEnd) {
                l.bb = bb >> l_total(e);
                l.bc = bc - l_total(e);
                return kBlockEnded;
            }
            if ((e & kKindMask) != kKindVal) return kBad;
            len = l_base(e) + (uint32_t)((bb >> l_code_bits(e)) & ((1u << l_extra_bits(e)) - 1));
            bb >>= l_total(e);
            bc -= l_total(e);
        }
        const uint32_t* dsel = len ? l.dtab : kNoMatch;
        uint32_t d = dsel[bb & kDMask];
        if (__builtin_expect((d & kKindMask) == kKindSub, 0)) {
            const uint32_t sb = (d >> 24) & 15;
            if (sb == 0) return kBad;
            bb >>= kDistBits;
            bc -= kDistBits;
            d = l.dtab[((d >> 8) & 0xFFFF) + (bb & ((1u << sb) - 1))];
            if ((d & kKindMask) != kKindVal) return kBad;
        }
        const size_t dist = ((d >> 8) & 0xFFFF) + (size_t)((bb >> ((d >> 24) & 15)) & ((1u << (d >> 28)) - 1));
        l.bb = bb >> (d & kBitsMask);
        l.bc = bc - (d & kBitsMask);
        if (__builtin_expect(dist > op, 0)) return kBad;
        uint8_t* dst = l.out + op;
        const uint8_t* src = dst - dist;
        if (__builtin_expect(dist >= 8 || dist == 0, 1)) {
            memcpy(dst, src, 8);
            memcpy(dst + 8, src + 8, 8);
            for (uint32_t k = 16; k < len; k += 8) memcpy(dst + k, src + k, 8);
        } else if (dist == 1) {
            memset(dst, src[0], len);
        } else {
            for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
        }
        l.op = op + len;
        return kGoOn;
    }
    // The loops themselves, compiled twice: for the baseline x86-64 and with BMI1/2 (shifts that take their count from
    // any register, bzhi: about a tenth fewer instructions per round), chosen once at run time.
    static inline __attribute__((always_inline)) bool fast_loop(Stream& s) {
        Lane l;
        s.load(l);
        Round r = kGoOn;
        while (in_hand(l) && (r = round(l)) == kGoOn) {}
        s.store(l, r);
        return r != kBad;
    }
    static inline __attribute__((always_inline)) void fast_loop_pair(Stream& a, Stream& b) {
        Lane la, lb;
        a.load(la);
        b.load(lb);
        Round ra = kGoOn, rb = kGoOn;
        while (in_hand(la) && in_hand(lb)) {
            ra = round(la);
            rb = round(lb);
            if (ra | rb) break;
        }
        a.store(la, ra);
        b.store(lb, rb);
    }
#if defined(__x86_64__)
    __attribute__((target("bmi,bmi2"))) static bool fast_loop_bmi2(Stream& s) { return fast_loop(s); }
    __attribute__((target("bmi,bmi2"))) static void fast_loop_pair_bmi2(Stream& a, Stream& b) { fast_loop_pair(a, b); }
    static bool have_bmi2() {
        static const bool yes = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
        return yes;
    }
#else
    static bool fast_loop_bmi2(Stream& s) { return fast_loop(s); }
    static void fast_loop_pair_bmi2(Stream& a, Stream& b) { fast_loop_pair(a, b); }
    static bool have_bmi2() { return false; }
#endif
    static bool fast_loop_base(Stream& s) { return fast_loop(s); }
    static void fast_loop_pair_base(Stream& a, Stream& b) { fast_loop_pair(a, b); }
    bool decode_fast() {  // after prepare() said kReadyFast
        return have_bmi2() ? fast_loop_bmi2(*this) : fast_loop_base(*this);
    }
    static void decode_fast_pair(Stream& a, Stream& b) {
        if (have_bmi2()) fast_loop_pair_bmi2(a, b);
        else fast_loop_pair_base(a, b);
    }

    // The first and last few bytes of a stream: one entry at a time, every read and write checked.
    bool decode_careful(size_t stop) {
        const uint32_t* lit = cur_->lit;
        const uint32_t lit_bits = (uint32_t)cur_->lit_bits, kMask = (1u << lit_bits) - 1;
        while (op_ < stop) {
            if (fast_possible()) return true;  // the fast loop can go on
            fill();
            uint32_t e = lit[bitbuf_ & kMask];
            if (!(e & kBitsMask) && (e & kKindMask) == kKindSub) {
                const uint32_t sb = l_code_bits(e);
                if (sb == 0 || lit_bits > bitcnt_) return false;
                bitbuf_ >>= lit_bits;
                bitcnt_ -= lit_bits;
                e = lit[l_sub_start(e) + (bitbuf_ & ((1u << sb) - 1))];
                if (!(e & kBitsMask) && (e & kKindMask) == kKindSub) return false;
            }
            const uint32_t all = (e & kBitsMask) ? (e & kBitsMask) : l_total(e);
            if (all > bitcnt_) return false;
            uint32_t len;
            if (e & kBitsMask) {
                const uint32_t cnt = (e >> 6) & 3;
                if (cnt > cap_ - op_) return false;
                for (uint32_t k = 0; k < cnt; ++k) out_[op_ + k] = (uint8_t)(e >> (8 + 8 * k));
                op_ += cnt;
                len = cnt == 3 ? 0 : (e >> 24) & 15;
            } else if ((e & kKindMask) == kKindEnd) {
                bitbuf_ >>= all;
                bitcnt_ -= all;
                phase_ = final_ ? kDone : kHeader;
                return true;
            } else if ((e & kKindMask) == kKindVal) {
                len = l_base(e) + (uint32_t)((bitbuf_ >> l_code_bits(e)) & ((1u << l_extra_bits(e)) - 1));
            } else {
                return false;
            }
            bitbuf_ >>= all;
            bitcnt_ -= all;
            if (len && !match_checked(len)) return false;
        }
        return true;
    }

    const uint8_t* in_ = nullptr;
    size_t in_n_ = 0, ip_ = 0;
    uint64_t bitbuf_ = 0;
    uint32_t bitcnt_ = 0;
    uint8_t* out_ = nullptr;
    size_t cap_ = 0, op_ = 0, stop_ = 0, target_ = 0;
    bool to_end_ = false;
    Phase phase_ = kFailed;
    bool final_ = false;
    uint32_t stored_left_ = 0;
    const Tables* cur_ = nullptr;
    Tables own_;
};

}  // namespace svx_inflate
#endif  // SVX_INFLATE_H_
