// svx_bam.cpp — native BAM ingest of libsvx.so (C-ABI in include/svx_bam.h).
//
// Host code only (no kernels).  Replaces what the reference gets from pysam/htslib under
// `bam.fetch(contig=...)` (SVIM_COLLECT.py:65-71): BGZF inflate, record walk, CIGAR words,
// aux tags, 4-bit SEQ slices.  Design:
//   * the file is memory-mapped; a BGZF member is inflated only when bytes of it are needed
//     (record header / name / CIGAR / aux, or a requested SEQ slice); everything else is
//     hopped over with the BSIZE / ISIZE fields (SAM spec §4.1);
//   * a `.bai` gives record boundaries (bin chunk begins/ends, linear index entries): the
//     requested contigs' ranges are cut there into pieces of about equal compressed size and
//     walked by a pool of threads, each with its own inflater;
//   * inflate: the build's own DEFLATE decoder (svx_inflate.h; SVX_BAM_ZLIB=1: zlib, its oracle); the
//     CRC32 of every member inflated to its end is checked, as htslib does (libdeflate's CRC routine
//     when the runtime has the library — dlopen, optional —, zlib's otherwise).
#include "svx_bam.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "svx.h"
#include "svx_inflate.h"
#include "svx_inflate_dev.h"

namespace {

inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// ------------------------------------------------------------------ optional libdeflate
struct LibDeflate {
    void* handle = nullptr;
    void* (*alloc)(void) = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
    void (*release)(void*) = nullptr;
};

const LibDeflate* libdeflate() {
    static const LibDeflate lib = [] {
        LibDeflate d;
        const char* off = getenv("SVX_BAM_ZLIB");  // force the zlib path (tests)
        if (off && off[0] == '1') return d;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return d;
        d.alloc = reinterpret_cast<void* (*)(void)>(dlsym(h, "libdeflate_alloc_decompressor"));
        d.decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(
            dlsym(h, "libdeflate_deflate_decompress"));
        d.crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        d.release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        if (d.alloc && d.decompress && d.crc32 && d.release) d.handle = h;
        else dlclose(h);
        return d;
    }();
    return lib.handle ? &lib : nullptr;
}

// Which decoder inflates the members: the build's own (svx_inflate.h; default), or — SVX_BAM_ZLIB=1 — zlib, kept as
// the differential oracle of the former.  libdeflate, when the runtime has it, only lends its CRC32 (carry-less
// multiply: 64 KiB in a few µs where zlib's table walk takes 40).
bool use_zlib() {
    static const bool z = [] { const char* v = getenv("SVX_BAM_ZLIB"); return v && v[0] == '1'; }();
    return z;
}
uint32_t member_crc(const uint8_t* p, size_t n) {
    const LibDeflate* L = libdeflate();
    if (L) return L->crc32(0, p, n);
    return (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), p, (uInt)n);
}

struct Inflater {
    z_stream zs;
    bool z_ready = false;
    std::unique_ptr<svx_inflate::Stream> own;
    uint64_t n_blocks = 0;

    Inflater() { memset(&zs, 0, sizeof(zs)); }
    Inflater(const Inflater&) = delete;
    Inflater& operator=(const Inflater&) = delete;
    ~Inflater() {
        if (z_ready) inflateEnd(&zs);
    }
    // Streaming use: begin() a member, then extend() the inflated prefix as far as somebody needs it.
    // A slice of a contig-sized SEQ field sits somewhere inside a 64 KiB member: inflating only up to its last
    // byte halves the work on average.  The CRC covers whole members, so it is checked when (and only when)
    // the prefix reaches the member's end.
    bool begin(const uint8_t* in, size_t in_len) {
        ++n_blocks;
        if (!use_zlib()) {
            if (!own) own.reset(new svx_inflate::Stream());
            own->begin(in, in_len);
            return true;
        }
        if (!z_ready) {
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            z_ready = true;
        } else if (inflateReset(&zs) != Z_OK) {
            return false;
        }
        zs.next_in = const_cast<Bytef*>(in);
        zs.avail_in = (uInt)in_len;
        return true;
    }
    // `out` holds `have` bytes of the member already (zlib: exactly; own decoder: at least — it may have run past the
    // last request by up to one match, *valid is what is there now)
    bool extend(uint8_t* out, size_t have, size_t want, size_t member_len, uint32_t crc, uint32_t* valid) {
        const bool whole = want == member_len;
        if (!use_zlib()) {
            if (!own->run(out, member_len, want, whole)) return false;
            *valid = (uint32_t)own->produced();
            if (whole) return own->produced() == member_len && member_crc(out, member_len) == crc;
            return true;
        }
        *valid = (uint32_t)want;
        if (want <= have && !whole) return true;
        zs.next_out = out + have;
        zs.avail_out = (uInt)(want - have);
        const int rc = inflate(&zs, whole ? Z_FINISH : Z_SYNC_FLUSH);
        if (zs.avail_out != 0 || (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR)) return false;
        if (whole) {
            if (rc != Z_STREAM_END) return false;
            return member_crc(out, member_len) == crc;
        }
        return true;
    }
    // One or two members at once, each inflated up to want[k] of its isize[k] bytes (want == isize: to the end, CRC32
    // checked); with the build's own decoder two members are decoded side by side (svx_inflate::Stream::run_pair).
    static bool run_two(Inflater inf[2], const uint8_t* const in[2], const size_t in_len[2], std::vector<uint8_t> buf[2],
                        const size_t isize[2], const size_t want[2], const uint32_t crc[2], size_t n) {
        if (n == 2 && !use_zlib()) {
            for (int k = 0; k < 2; ++k) {
                ++inf[k].n_blocks;
                if (!inf[k].own) inf[k].own.reset(new svx_inflate::Stream());
                inf[k].own->begin(in[k], in_len[k]);
            }
            bool ok[2] = {false, false};
            svx_inflate::Stream::run_pair(*inf[0].own, buf[0].data(), isize[0], want[0], want[0] == isize[0], &ok[0],
                                          *inf[1].own, buf[1].data(), isize[1], want[1], want[1] == isize[1], &ok[1]);
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) return false;
                if (want[k] == isize[k] && (inf[k].own->produced() != isize[k] || member_crc(buf[k].data(), isize[k]) != crc[k]))
                    return false;
            }
            return true;
        }
        for (size_t k = 0; k < n; ++k) {
            uint32_t valid = 0;
            if (!inf[k].begin(in[k], in_len[k]) || !inf[k].extend(buf[k].data(), 0, want[k], isize[k], crc[k], &valid)) return false;
        }
        return true;
    }
    // raw deflate stream `in` → exactly out_len bytes, CRC32 checked
    bool run(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, uint32_t crc) {
        uint32_t valid = 0;
        return begin(in, in_len) && extend(out, 0, out_len, out_len, crc, &valid);
    }
};

// ------------------------------------------------------------------ BGZF members
struct Blk {
    uint32_t bsize = 0, isize = 0, payload_off = 0, payload_len = 0, crc = 0;
};

// 0 ok, 1 clean end of file (coff == fsize), -1 malformed
int parse_block(const uint8_t* map, uint64_t fsize, uint64_t coff, Blk* b) {
    if (coff == fsize) return 1;
    if (coff + 18 > fsize) return -1;
    const uint8_t* p = map + coff;
    if (p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || !(p[3] & 4)) return -1;
    const uint32_t xlen = le16(p + 10);
    if (coff + 12 + xlen + 8 > fsize) return -1;
    uint32_t q = 12, end = 12 + xlen, bsize = 0;
    while (q + 4 <= end) {
        const uint32_t slen = le16(p + q + 2);
        if (p[q] == 66 && p[q + 1] == 67 && slen == 2 && q + 6 <= end) bsize = (uint32_t)le16(p + q + 4) + 1;
        q += 4 + slen;
    }
    if (!bsize || bsize < xlen + 20 || coff + bsize > fsize) return -1;
    b->bsize = bsize;
    b->payload_off = 12 + xlen;
    b->payload_len = bsize - xlen - 20;
    b->crc = le32(p + bsize - 8);
    b->isize = le32(p + bsize - 4);
    if (b->isize > 65536) return -1;
    return 0;
}

struct VPos {
    uint64_t coff = 0;
    uint32_t uoff = 0;
    bool operator<(const VPos& o) const { return coff != o.coff ? coff < o.coff : uoff < o.uoff; }
    bool operator==(const VPos& o) const { return coff == o.coff && uoff == o.uoff; }
};

struct File {
    const uint8_t* map = nullptr;
    uint64_t fsize = 0;
};

// Sequential reader over the uncompressed stream that inflates a member only when bytes of it
// are copied out.  Positions are kept canonical: never at the end of a member (uoff == isize
// moves to the start of the next one, empty members are passed), so that they compare equal
// to the virtual offsets htslib writes into an index.
struct Cursor {
    const File* f;
    Inflater* inf;
    uint64_t coff = 0;
    uint32_t uoff = 0;
    Blk blk;
    bool eof = false;
    bool bad = false;
    uint64_t buf_coff = ~0ull;  // member currently held in `buf`
    uint32_t buf_valid = 0;     // bytes of it inflated so far (prefix mode; the whole member otherwise)
    bool prefix_mode = false;   // inflate members only as far as the bytes asked for (svx_bam_seq_slices)
    bool whole = false;         // the member in `buf` has been inflated to its end and its CRC32 checked
    uint64_t n_hopped = 0;      // members passed (inflated or not)
    std::vector<uint64_t>* verified = nullptr;  // (walks) file offsets of the members inflated whole with their CRC32 checked
    std::vector<uint64_t>* touched = nullptr;   // (walks with the check deferred) file offsets of every member bytes were taken from
    std::vector<uint8_t> buf;

    Cursor(const File* file, Inflater* i) : f(file), inf(i) {}

    bool settle() {  // make (coff, uoff) canonical; false on malformed data
        for (;;) {
            const int rc = parse_block(f->map, f->fsize, coff, &blk);
            if (rc == 1) { eof = true; uoff = 0; return true; }
            if (rc < 0) { bad = true; return false; }
            if (uoff > blk.isize) { bad = true; return false; }
            if (uoff < blk.isize) { eof = false; return true; }
            coff += blk.bsize;
            uoff = 0;
            ++n_hopped;
        }
    }
    bool seek(VPos p) {
        coff = p.coff;
        uoff = p.uoff;
        return settle();
    }
    VPos tell() const { VPos p; p.coff = coff; p.uoff = uoff; return p; }
    bool ensure(uint32_t upto) {  // bytes [0, upto) of the current member are in `buf`
        if (buf.empty()) buf.resize(65536);
        if (buf_coff != coff) {
            buf_coff = ~0ull;
            if (!inf->begin(f->map + coff + blk.payload_off, blk.payload_len)) { bad = true; return false; }
            buf_valid = 0;
            whole = false;
            if (touched) touched->push_back(coff);
        } else if (buf_valid >= upto && (prefix_mode || whole)) {
            return true;
        }
        // prefix mode: inflate up to the last byte asked for; otherwise the whole member, CRC32 checked
        const uint32_t want = prefix_mode ? upto : blk.isize;
        uint32_t valid = 0;
        if (!inf->extend(buf.data(), buf_valid, want, blk.isize, blk.crc, &valid)) { bad = true; buf_coff = ~0ull; return false; }
        buf_coff = coff;
        buf_valid = valid;
        if (want == blk.isize && !whole && verified) verified->push_back(coff);
        whole = want == blk.isize;
        return true;
    }
    bool read(void* dst, size_t n) {
        uint8_t* d = static_cast<uint8_t*>(dst);
        while (n) {
            if (eof || bad) { bad = true; return false; }
            const size_t take = std::min<size_t>(n, blk.isize - uoff);
            if (!ensure(uoff + (uint32_t)take)) return false;
            memcpy(d, buf.data() + uoff, take);
            d += take;
            n -= take;
            uoff += (uint32_t)take;
            if (!settle()) return false;
        }
        return true;
    }
    bool skip(uint64_t n) {
        while (n) {
            if (eof || bad) { bad = true; return false; }
            const uint64_t take = std::min<uint64_t>(n, blk.isize - uoff);
            n -= take;
            uoff += (uint32_t)take;
            if (!settle()) return false;
        }
        return true;
    }
};

// ------------------------------------------------------------------ record columns
struct Chunk {
    std::vector<int32_t> tid, pos, l_seq, ref_len;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint32_t> n_cig, name_len, aux_len, sa_len;
    std::vector<int64_t> sa_off;  // relative to the record's aux start
    std::vector<uint64_t> voffset, seq_coff;
    std::vector<uint32_t> seq_uoff;
    std::vector<uint32_t> cigar;
    std::vector<char> names;
    std::vector<uint8_t> aux;
    uint64_t blocks_spanned = 0;
    std::vector<uint64_t> verified;  // members this walk inflated whole and CRC-checked (verifying ingest)
    std::vector<uint64_t> touched;   // members this walk took bytes from without checking them (svx_bam_set_defer_verify)
    std::string err;
};

// size of one aux value starting at p (type byte at p[0]); 0 on malformed data
size_t aux_value_size(const uint8_t* p, const uint8_t* end) {
    if (p >= end) return 0;
    switch (p[0]) {
        case 'A': case 'c': case 'C': return 2;
        case 's': case 'S': return 3;
        case 'i': case 'I': case 'f': return 5;
        case 'd': return 9;
        case 'Z': case 'H': {
            const uint8_t* q = p + 1;
            while (q < end && *q) ++q;
            return q < end ? (size_t)(q - p) + 1 : 0;
        }
        case 'B': {
            if (p + 6 > end) return 0;
            size_t es;
            switch (p[1]) {
                case 'c': case 'C': es = 1; break;
                case 's': case 'S': es = 2; break;
                case 'i': case 'I': case 'f': es = 4; break;
                default: return 0;
            }
            const uint64_t cnt = le32(p + 2);
            const uint64_t tot = 6 + cnt * es;
            return p + tot <= end ? (size_t)tot : 0;
        }
        default: return 0;
    }
}

const uint32_t kRefMask = 0x18D;  // M D N = X consume the reference (htslib bam_cigar2rlen)

// Walk records from `start` to `stop` (exclusive; canonical position) or to the end of the file.
bool walk_records(const File* f, VPos start, bool have_stop, VPos stop, Inflater* inf, Chunk* out, bool whole_members,
                  bool note_touched = false) {
    Cursor c(f, inf);
    // a record's head, name and CIGAR are followed by its SEQ bytes — the expensive kind to inflate (svx_inflate.h) —
    // which the walk hops over: stopping right behind the CIGAR is a third of the CPU time of inflating the member
    c.prefix_mode = !whole_members;
    if (whole_members) c.verified = &out->verified;
    else if (note_touched) c.touched = &out->touched;
    char msg[256];
    if (!c.seek(start)) { out->err = "malformed BGZF member at a walk start"; return false; }
    std::vector<uint8_t> body;
    std::vector<uint8_t> auxbuf;
    for (;;) {
        const VPos here = c.tell();
        if (have_stop) {
            if (here == stop) break;
            if (stop < here || c.eof) {
                out->err = "index does not match the file: a record walk passed an indexed record start";
                return false;
            }
        } else if (c.eof) {
            break;
        }
        uint8_t fix[36];
        if (!c.read(fix, 36)) { out->err = "truncated BAM record header"; return false; }
        const int32_t block_size = (int32_t)le32(fix);
        const int32_t tid = (int32_t)le32(fix + 4), pos = (int32_t)le32(fix + 8);
        const uint32_t l_rn = fix[12], mapq = fix[13];
        uint32_t n_cig = le16(fix + 16);
        const uint32_t flag = le16(fix + 18);
        const int32_t l_seq = (int32_t)le32(fix + 20);
        const uint64_t seq_bytes = l_seq > 0 ? ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq : 0;
        const uint64_t fixed = 32ull + l_rn + 4ull * n_cig + seq_bytes;
        if (block_size < 32 || l_seq < 0 || l_rn == 0 || (uint64_t)block_size < fixed) {
            snprintf(msg, sizeof(msg), "corrupt BAM record at virtual offset %llu:%u",
                     (unsigned long long)here.coff, here.uoff);
            out->err = msg;
            return false;
        }
        body.resize(l_rn + 4ull * n_cig);
        if (!c.read(body.data(), body.size())) { out->err = "truncated BAM record (name/CIGAR)"; return false; }
        const VPos seq_at = c.tell();
        if (!c.skip(seq_bytes)) { out->err = "truncated BAM record (SEQ/QUAL)"; return false; }
        const uint64_t aux_bytes = (uint64_t)block_size - fixed;
        auxbuf.resize(aux_bytes);
        if (aux_bytes && !c.read(auxbuf.data(), aux_bytes)) { out->err = "truncated BAM record (aux)"; return false; }

        size_t nl = 0;
        while (nl < l_rn && body[nl]) ++nl;
        // ---- aux walk: locate SA:Z and CG:B,I
        const uint8_t* ab = auxbuf.data();
        const uint8_t* ae = ab + aux_bytes;
        const uint8_t *sa = nullptr, *cg = nullptr;
        size_t sa_n = 0, cg_total = 0;
        for (const uint8_t* p = ab; p + 3 <= ae;) {
            const size_t vs = aux_value_size(p + 2, ae);
            if (!vs) {
                snprintf(msg, sizeof(msg), "malformed aux field in record '%.*s'", (int)nl, (const char*)body.data());
                out->err = msg;
                return false;
            }
            if (p[0] == 'S' && p[1] == 'A' && p[2] == 'Z' && !sa) { sa = p + 3; sa_n = vs - 2; }
            if (p[0] == 'C' && p[1] == 'G' && !cg) { cg = p; cg_total = 2 + vs; }
            p += 2 + vs;
        }
        // ---- long CIGAR (SAM spec §4.2.2; condition of htslib bam_tag2cigar)
        const uint8_t* cig_src = body.data() + l_rn;
        bool cg_used = false;
        if (cg && n_cig > 0 && tid >= 0 && pos >= 0 && cg[2] == 'B' && (cg[3] == 'I' || cg[3] == 'i')) {
            const uint32_t w0 = le32(cig_src);
            const uint32_t cg_len = le32(cg + 4);
            if ((w0 & 15) == 4 && (w0 >> 4) == (uint32_t)l_seq && cg_len >= n_cig && cg_len < (1u << 29)) {
                cig_src = cg + 8;
                n_cig = cg_len;
                cg_used = true;
            }
        }
        int64_t rlen = 0;
        const size_t c0 = out->cigar.size();
        out->cigar.resize(c0 + n_cig);
        for (uint32_t k = 0; k < n_cig; ++k) {
            const uint32_t w = le32(cig_src + 4 * k);
            out->cigar[c0 + k] = w;
            if ((kRefMask >> (w & 15)) & 1) rlen += w >> 4;
        }
        out->tid.push_back(tid);
        out->pos.push_back(pos);
        out->l_seq.push_back(l_seq);
        out->ref_len.push_back((int32_t)std::min<int64_t>(rlen, 0x7FFFFFFF));
        out->flag.push_back((uint16_t)flag);
        out->mapq.push_back((uint8_t)mapq);
        out->n_cig.push_back(n_cig);
        out->voffset.push_back((here.coff << 16) | here.uoff);
        out->seq_coff.push_back(seq_at.coff);
        out->seq_uoff.push_back(seq_at.uoff);
        out->name_len.push_back((uint32_t)nl);
        out->names.insert(out->names.end(), body.begin(), body.begin() + nl);
        // aux bytes, CG removed when it was moved into the CIGAR (pysam no longer shows the tag)
        const size_t a0 = out->aux.size();
        if (cg_used) {
            out->aux.insert(out->aux.end(), ab, cg);
            out->aux.insert(out->aux.end(), cg + cg_total, ae);
            if (sa && sa > cg) sa -= cg_total;
        } else {
            out->aux.insert(out->aux.end(), ab, ae);
        }
        out->aux_len.push_back((uint32_t)(out->aux.size() - a0));
        out->sa_off.push_back(sa ? (int64_t)(sa - ab) : -1);
        out->sa_len.push_back(sa ? (uint32_t)sa_n : 0);
    }
    out->blocks_spanned = c.n_hopped;
    return true;
}

// ------------------------------------------------------------------ .bai
struct RefIndex {
    std::vector<uint64_t> points;  // virtual offsets that are record boundaries (chunk begins/ends, linear index)
    uint64_t lo = ~0ull, hi = 0;   // [first chunk begin, last chunk end) as virtual offsets
};

bool read_file(const std::string& p, std::vector<uint8_t>* out) {
    FILE* fh = fopen(p.c_str(), "rb");
    if (!fh) return false;
    fseek(fh, 0, SEEK_END);
    const long n = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(out->data(), 1, out->size(), fh) == out->size();
    fclose(fh);
    return ok;
}

bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

bool parse_bai(const std::vector<uint8_t>& d, int32_t n_ref_expected, std::vector<RefIndex>* refs) {
    size_t p = 0;
    const size_t n = d.size();
    if (n < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    const int32_t n_ref = (int32_t)le32(d.data() + 4);
    if (n_ref != n_ref_expected) return false;
    p = 8;
    refs->assign(n_ref, RefIndex());
    for (int32_t r = 0; r < n_ref; ++r) {
        RefIndex& R = (*refs)[r];
        if (p + 4 > n) return false;
        const int32_t n_bin = (int32_t)le32(d.data() + p);
        p += 4;
        if (n_bin < 0) return false;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > n) return false;
            const uint32_t bin = le32(d.data() + p);
            const int32_t n_chunk = (int32_t)le32(d.data() + p + 4);
            p += 8;
            if (n_chunk < 0 || p + 16ull * n_chunk > n) return false;
            for (int32_t k = 0; k < n_chunk; ++k) {
                const uint64_t beg = le64(d.data() + p), end = le64(d.data() + p + 8);
                p += 16;
                if (bin == 37450) continue;  // metadata pseudo-bin (file range, mapped/unmapped counts)
                if (end < beg) return false;
                R.points.push_back(beg);
                R.points.push_back(end);
                R.lo = std::min(R.lo, beg);
                R.hi = std::max(R.hi, end);
            }
        }
        if (p + 4 > n) return false;
        const int32_t n_intv = (int32_t)le32(d.data() + p);
        p += 4;
        if (n_intv < 0 || p + 8ull * n_intv > n) return false;
        for (int32_t k = 0; k < n_intv; ++k) {
            const uint64_t v = le64(d.data() + p);
            p += 8;
            if (v) R.points.push_back(v);
        }
        std::sort(R.points.begin(), R.points.end());
        R.points.erase(std::unique(R.points.begin(), R.points.end()), R.points.end());
        // linear-index entries outside the chunk range would be inconsistent
        if (!R.points.empty() && (R.points.front() < R.lo || R.points.back() > R.hi)) return false;
    }
    return true;
}

// `.csi` (the coordinate-sorted index htslib writes with `samtools index -c`, needed for contigs above 512 Mbp; CSI v1
// specification): a BGZF file of its own — magic, min_shift, depth, auxiliary bytes, then per sequence the bins with
// a `loffset` each (no linear index).  Gives the same thing a `.bai` does here: record boundaries.
bool inflate_all(const std::vector<uint8_t>& raw, std::vector<uint8_t>* out) {
    Inflater inf;
    uint64_t coff = 0;
    out->clear();
    for (;;) {
        Blk blk;
        const int rc = parse_block(raw.data(), raw.size(), coff, &blk);
        if (rc == 1) return true;
        if (rc < 0 || out->size() + blk.isize > (512u << 20)) return false;
        const size_t at = out->size();
        out->resize(at + blk.isize);
        if (blk.isize && !inf.run(raw.data() + coff + blk.payload_off, blk.payload_len, out->data() + at, blk.isize, blk.crc))
            return false;
        coff += blk.bsize;
    }
}

bool parse_csi(const std::vector<uint8_t>& raw, int32_t n_ref_expected, std::vector<RefIndex>* refs) {
    std::vector<uint8_t> d;
    if (!inflate_all(raw, &d)) return false;
    const size_t n = d.size();
    if (n < 16 || memcmp(d.data(), "CSI\1", 4) != 0) return false;
    const int32_t min_shift = (int32_t)le32(d.data() + 4), depth = (int32_t)le32(d.data() + 8), l_aux = (int32_t)le32(d.data() + 12);
    if (min_shift < 0 || min_shift > 31 || depth < 0 || depth > 10 || l_aux < 0 || 16ull + (uint64_t)l_aux + 4 > n) return false;
    size_t p = 16 + (size_t)l_aux;
    const int32_t n_ref = (int32_t)le32(d.data() + p);
    p += 4;
    if (n_ref != n_ref_expected) return false;
    const uint64_t meta_bin = ((1ull << ((depth + 1) * 3)) - 1) / 7 + 1;  // metadata pseudo-bin
    refs->assign(n_ref, RefIndex());
    for (int32_t r = 0; r < n_ref; ++r) {
        RefIndex& R = (*refs)[r];
        if (p + 4 > n) return false;
        const int32_t n_bin = (int32_t)le32(d.data() + p);
        p += 4;
        if (n_bin < 0) return false;
        std::vector<uint64_t> first_overlaps;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 16 > n) return false;
            const uint32_t bin = le32(d.data() + p);
            const uint64_t loffset = le64(d.data() + p + 4);
            const int32_t n_chunk = (int32_t)le32(d.data() + p + 12);
            p += 16;
            if (n_chunk < 0 || p + 16ull * n_chunk > n) return false;
            for (int32_t k = 0; k < n_chunk; ++k) {
                const uint64_t beg = le64(d.data() + p), end = le64(d.data() + p + 8);
                p += 16;
                if (bin == meta_bin) continue;
                if (end < beg) return false;
                R.points.push_back(beg);
                R.points.push_back(end);
                R.lo = std::min(R.lo, beg);
                R.hi = std::max(R.hi, end);
            }
            if (bin != meta_bin && loffset) first_overlaps.push_back(loffset);
        }
        for (uint64_t v : first_overlaps)  // the first record overlapping a bin: a record boundary of this sequence
            if (v >= R.lo && v <= R.hi) R.points.push_back(v);
        std::sort(R.points.begin(), R.points.end());
        R.points.erase(std::unique(R.points.begin(), R.points.end()), R.points.end());
    }
    return true;
}

template <typename T>
T* dup_array(const std::vector<T>& v) {
    T* p = static_cast<T*>(malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

// Worker threads of one handle, started once and kept: svx_bam_load and svx_bam_seq_slices are 20-60 ms calls, and
// threads created at the start of such a call spend a good part of it where the kernel first put them — next to
// their parent — before the balancer spreads them over the node's cores (same slices, 64 threads: 17 ms or 90 ms
// from one call to the next with a thread per call).  Kept threads stay where the first call spread them.
class Pool {
  public:
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        work_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    // fn() on n threads at once; returns when all have returned.  One run at a time (a handle is used by one thread).
    void run(int n, const std::function<void()>& fn) {
        if (n <= 1) {
            fn();
            return;
        }
        std::unique_lock<std::mutex> g(m_);
        while ((int)th_.size() < n) {
            const int idx = (int)th_.size();
            th_.emplace_back([this, idx] { loop(idx); });
        }
        job_ = &fn;
        active_ = n;
        pending_ = n;
        ++generation_;
        work_.notify_all();
        done_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(int idx) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            work_.wait(g, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (idx >= active_) continue;
            const std::function<void()>* job = job_;
            g.unlock();
            (*job)();
            g.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable work_, done_;
    std::vector<std::thread> th_;
    const std::function<void()>* job_ = nullptr;
    int active_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// Device lanes: the streams (and page-locked staging rings) the reader uses on a device, ONE set per device for the
// whole process, brought up by a background thread that svx_bam_load starts before its record walk.  A stream's
// creation plus its first host-to-device copy cost 10-25 ms on gfx950 (rocprofv3 --hip-trace of the command: the first
// hipMemcpyAsync of a fresh stream returns after 12-15 ms) — per reader handle that was paid in every load; here it is
// paid once and hides behind the walk.
//   upload    the copy stream of the CIGAR pools' device copies (svx_bam_device_pool)
//   inflate   kInflateLanes streams, each with a ring of page-locked staging slots, for svx_bam_seq_slices' device leg
//             (svx_bam_set_device_inflate): one per call in flight — both haplotype BAMs decode at the same time
static svx_inflate_launch_fn g_inflate_launch = nullptr;  // svx_inflate.hip registers its launches when the library loads
static svx_gather_launch_fn g_gather_launch = nullptr;    // (a build of this file alone — the sanitizer tests — has none)
extern "C" void svx_bam_register_device_kernels(svx_inflate_launch_fn inflate, svx_gather_launch_fn gather) {
    g_inflate_launch = inflate;
    g_gather_launch = gather;
}

constexpr int kInflateLanes = 16;  // at most; a device gets g_inflate_lanes of them when they are brought up
// How many a device's readers get (svx_bam_set_inflate_lanes, before the first load with a device share).  Two serve a
// diploid sample's two readers; a process that decodes many samples at once (svim-asm-cohort) keeps more calls in flight —
// a call holds its lane for 40-60 ms, staging included, and with the walks' check on the leg the lanes, not the CPUs, were
// what its workers waited for (N = 24: 9.5 samples/s on two lanes in one process, 11.5 as two processes with two each).
static std::atomic<int> g_inflate_lanes{2};
extern "C" int svx_bam_set_inflate_lanes(int lanes) {
    if (lanes < 1 || lanes > kInflateLanes) return SVX_E_INVALID;
    g_inflate_lanes.store(lanes);
    return SVX_OK;
}
#ifndef SVX_STAGE_POLL_US
#define SVX_STAGE_POLL_US 20  // how long a staging thread sleeps between looks at its slot's turn
#endif
#ifndef SVX_RING_SLOTS
#define SVX_RING_SLOTS 8
#endif
#ifndef SVX_SLOT_MB
#define SVX_SLOT_MB 2
#endif
constexpr int kRingSlots = SVX_RING_SLOTS;
constexpr size_t kSlotBytes = (size_t)SVX_SLOT_MB << 20;
constexpr int kLegPhasesMax = 8;
// SVX_BAM_LEG_PHASES=N: the device leg stages and decodes its members in N phases, a phase's kernels (on a second stream)
// beside the next phase's copies.  Measured on the full-size sample with both readers' calls at once (tools/r06_leg_probe.py,
// profiles/r06_leg_phases.txt): 1 phase 55-81 ms, 2 phases 51-79, 3 phases 67-100, 4 phases 70-83 — both readers' copies
// already share the host link, and every set of launches costs a member's latency; so ONE phase, and this stays a switch.
static uint32_t leg_phases_asked() {
    static const uint32_t n = [] { const char* e = getenv("SVX_BAM_LEG_PHASES"); return e ? (uint32_t)std::max(1, atoi(e)) : 1u; }();
    return n;
}
struct InflateLane {
    hipStream_t stream = nullptr;     // the copies of the payloads
    hipStream_t kstream = nullptr;    // the kernels: a phase's members are decoded while the next phase's payloads travel
    uint8_t* ring = nullptr;          // kRingSlots * kSlotBytes, page-locked
    hipEvent_t slot_done[kRingSlots] = {};
    hipEvent_t staged[kLegPhasesMax] = {};  // behind the last copy of a phase
    hipEvent_t done = nullptr;
    std::mutex busy;                  // held by the call that uses the lane
};
struct DeviceLanes {
    std::once_flag once, inflate_once;
    std::mutex mu;
    std::condition_variable cv;
    bool upload_tried = false;
    hipStream_t upload = nullptr;
    std::atomic<bool> upload_up{false};
    InflateLane inflate[kInflateLanes];
    int n_inflate = 0;                    // lanes brought up (g_inflate_lanes at that time)
    std::atomic<bool> inflate_up{false};  // all inflate lanes are usable
    bool inflate_tried = false;
};
constexpr int kMaxLanes = 64;
static DeviceLanes g_lanes[kMaxLanes];

// ---- buffers kept between handles.  hipFree and hipHostFree wait for the device (every stream of the process) and unpin
// pages: a process that opens and closes readers all the time (svim-asm-cohort: 30-170 ms per sample inside close(), during
// which another worker's inflate kernels had to finish first) gives the buffers back here instead and the next handle of the
// same device takes one that fits.  At most kKeptMax of either kind, each taken only for a need of at least half its size.
struct KeptBuffer { void* p; size_t bytes; int device; };
static std::mutex g_kept_mu;
static std::vector<KeptBuffer> g_kept_dev, g_kept_host;
constexpr size_t kKeptMax = 12;
static bool kept_off() {
    static const bool off = getenv("SVX_BAM_NO_BUFFER_CACHE") != nullptr;
    return off;
}
static void* kept_take(std::vector<KeptBuffer>& v, int device, size_t need, size_t* got) {
    if (kept_off()) return nullptr;
    std::lock_guard<std::mutex> lock(g_kept_mu);
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].device == device && v[i].bytes >= need && v[i].bytes / 2 <= need + (4u << 20) && (best == v.size() || v[i].bytes < v[best].bytes))
            best = i;
    if (best == v.size()) return nullptr;
    void* p = v[best].p;
    *got = v[best].bytes;
    v.erase(v.begin() + (long)best);
    return p;
}
static bool kept_give(std::vector<KeptBuffer>& v, int device, void* p, size_t bytes) {
    if (kept_off() || !p || !bytes) return false;
    std::lock_guard<std::mutex> lock(g_kept_mu);
    if (v.size() >= kKeptMax) return false;
    v.push_back(KeptBuffer{p, bytes, device});
    return true;
}
// device memory of at least `need` bytes on the current device (`device`): *got = its size; nullptr: out of memory
static void* dev_buffer(int device, size_t need, size_t* got) {
    if (void* p = kept_take(g_kept_dev, device, need, got)) return p;
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = need;
    return p;
}
static void dev_buffer_done(int device, void* p, size_t bytes) {
    if (p && !kept_give(g_kept_dev, device, p, bytes)) (void)hipFree(p);
}
static void* host_buffer(int device, size_t need, size_t* got) {
    if (void* p = kept_take(g_kept_host, device, need, got)) return p;
    void* p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = need;
    return p;
}
static void host_buffer_done(int device, void* p, size_t bytes) {
    if (p && !kept_give(g_kept_host, device, p, bytes)) (void)hipHostFree(p);
}

// A new stream and one host-to-device copy of `bytes` through it (page-locked source `src`, or a buffer of its own): the
// first copy of a stream sets up its DMA path, and the first copy of this SIZE class does so once more (the command's
// trace: 6-9 ms for the first 512 KiB part of a pool on a stream that had only moved 4 KiB) — done here, beside the walk.
static bool warm_stream(hipStream_t* out, size_t bytes, void* src = nullptr) {
    hipStream_t st = nullptr;
    void *h = src, *d = nullptr;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (ok && !h) {
        ok = hipHostMalloc(&h, bytes, hipHostMallocDefault) == hipSuccess;
        if (ok) memset(h, 0, bytes);
    }
    ok = ok && hipMalloc(&d, bytes) == hipSuccess;
    ok = ok && hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (d) (void)hipFree(d);
    if (h && !src) (void)hipHostFree(h);
    if (!ok) {
        (void)hipGetLastError();
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
    }
    *out = st;
    return ok;
}

static void upload_lane_bring_up(int device) {
    DeviceLanes& L = g_lanes[device];
    hipStream_t up = nullptr;
    if (hipSetDevice(device) == hipSuccess) (void)warm_stream(&up, 1u << 20);
    else (void)hipGetLastError();
    {
        std::lock_guard<std::mutex> lock(L.mu);
        L.upload = up;
        L.upload_tried = true;
        L.upload_up.store(true);
    }
    L.cv.notify_all();
}

// the inflate lanes, each on a thread of its own: stream creations run side by side (each 10-25 ms), and a fresh
// process's first sequence-slice call comes only tens of milliseconds after its first load
static void inflate_lanes_bring_up(int device) {
    DeviceLanes& L = g_lanes[device];
    std::vector<std::thread> th;
    std::atomic<int> good(0);
    const bool want_inflate = g_inflate_launch && !getenv("SVX_BAM_NO_INFLATE_LANES");
    const int n_lanes = g_inflate_lanes.load();
    for (int k = 0; want_inflate && k < n_lanes; ++k)
        th.emplace_back([&L, &good, device, k] {
            InflateLane& I = L.inflate[k];
            void* ring = nullptr;
            bool ok = hipSetDevice(device) == hipSuccess && hipHostMalloc(&ring, kRingSlots * kSlotBytes, hipHostMallocDefault) == hipSuccess;
            if (ok) memset(ring, 0, kSlotBytes);
            I.ring = static_cast<uint8_t*>(ring);
            ok = ok && warm_stream(&I.stream, kSlotBytes, ring);
            if (leg_phases_asked() > 1) {  // (experiments: the leg's decode pipelined with its staging)
                ok = ok && hipStreamCreateWithFlags(&I.kstream, hipStreamNonBlocking) == hipSuccess;
                for (int q = 0; q < kLegPhasesMax && ok; ++q) ok = hipEventCreateWithFlags(&I.staged[q], hipEventDisableTiming) == hipSuccess;
            }
            // (blocking events: a thread that waits for a slot or for the leg sleeps — under a CPU quota a spinning wait
            //  spends the very seconds the leg is there to save)
            for (int q = 0; q < kRingSlots && ok; ++q)
                ok = hipEventCreateWithFlags(&I.slot_done[q], hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&I.done, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
            if (ok) good.fetch_add(1);
            else (void)hipGetLastError();
        });
    for (std::thread& t : th) t.join();
    {
        std::lock_guard<std::mutex> lock(L.mu);
        L.n_inflate = n_lanes;
        L.inflate_up.store(want_inflate && good.load() == n_lanes);
        L.inflate_tried = true;
    }
    L.cv.notify_all();
}

// start the lanes of `device` once per process (returns at once); the inflate lanes only for a reader that has a device
// share (svx_bam_set_device_inflate): a one-shot command without one does not pay for two more streams and 32 MB of ring
// The bring-up threads are detached and inside the HIP runtime for 10-50 ms: a process that leaves in order (exit(),
// not _exit) within that time must not tear the runtime's statics and g_lanes down under them.  An exit handler
// registered here — behind the runtime's own, so it runs before them — waits for the ones still running.
static std::atomic<int> g_bring_ups_running{0};
static void wait_for_bring_ups() {
    for (int i = 0; i < 4000 && g_bring_ups_running.load() > 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}
static void lanes_start(int device, bool with_inflate) {
    if (device < 0 || device >= kMaxLanes) return;
    static std::once_flag exit_hook;
    std::call_once(exit_hook, [] { (void)std::atexit(wait_for_bring_ups); });
    std::call_once(g_lanes[device].once, [device] {
        g_bring_ups_running.fetch_add(1);
        std::thread([device] { upload_lane_bring_up(device); g_bring_ups_running.fetch_sub(1); }).detach();
    });
    if (with_inflate)
        std::call_once(g_lanes[device].inflate_once, [device] {
            g_bring_ups_running.fetch_add(1);
            std::thread([device] { inflate_lanes_bring_up(device); g_bring_ups_running.fetch_sub(1); }).detach();
        });
}

static hipStream_t upload_stream(int device) {
    if (device < 0 || device >= kMaxLanes) return nullptr;
    lanes_start(device, false);
    DeviceLanes& L = g_lanes[device];
    std::unique_lock<std::mutex> lock(L.mu);
    L.cv.wait(lock, [&] { return L.upload_tried; });
    return L.upload;
}

struct svx_bam {
    int fd = -1;
    File file;
    int n_threads = 1;
    std::string path, err, text;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    VPos first_record;
    int index_state = 0;
    std::vector<RefIndex> refs;
    // loaded columns
    uint64_t n_records = 0;
    std::vector<int32_t> tid, pos, l_seq, ref_len;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint64_t> cigar_off, name_off, aux_off, voffset, seq_coff;
    std::vector<uint32_t> seq_uoff, sa_len;
    std::vector<int64_t> sa_off;
    uint32_t* cigar = nullptr;
    bool cigar_pinned = false;
    size_t cigar_pinned_bytes = 0;  // size of the page-locked buffer the pool lies in (>= the pool: it may come from the kept ones)
    std::vector<char> names;
    std::vector<uint8_t> aux;
    uint64_t blocks_inflated = 0, blocks_spanned = 0;
    std::vector<uint64_t> verified_members;  // sorted: members a record walk has inflated whole and CRC-checked — a sequence
                                             // slice that lands in one of them (a third do: the first member of a record's
                                             // SEQ bytes holds its head and CIGAR) needs only its prefix, not a second check
    int pin_device = -1;  // HIP device whose context page-locks the CIGAR pool; -1: pageable memory
    int inflate_pct = 0;  // share of a sequence-slice call whose members the device inflates (svx_bam_set_device_inflate)
    uint32_t inflate_min_members = 500;   // ... when that share holds at least this many members
    uint32_t inflate_wait_ms = 0;         // ... and how long a call waits for one of the device's inflate lanes to come free
    // svx_bam_set_defer_verify: the record walks take only the bytes they need of a member (as with the check off) and note
    // the member here; the next sequence-slice call's device leg checks these members whole beside its own, or the threads do
    // (at the end of that call, or in svx_bam_verify_pending).  Sorted, unique.
    bool defer_verify = false;
    std::vector<uint64_t> pending_members;
    uint8_t* d_inflate = nullptr;  // the device leg's buffer, kept between calls
    size_t d_inflate_cap = 0;
    // ... asked for ahead of the first sequence-slice call, on a thread beside whatever follows the record walk: a
    // hipMalloc of 0.7-1.1 GB takes 20-40 ms and holds the runtime's lock — the staging threads of the OTHER reader's leg
    // stood in hipMemcpyAsync for that long (8 x 25 ms of "enqueueing" in the leg's debug line, one reader of two)
    std::thread d_inflate_ahead;
    void* d_inflate_ahead_ptr = nullptr;
    size_t d_inflate_ahead_cap = 0;
    void take_inflate_ahead() {  // (the caller's thread, before it looks at d_inflate)
        if (!d_inflate_ahead.joinable()) return;
        d_inflate_ahead.join();
        if (d_inflate_ahead_ptr) {
            if (!d_inflate) { d_inflate = static_cast<uint8_t*>(d_inflate_ahead_ptr); d_inflate_cap = d_inflate_ahead_cap; }
            else dev_buffer_done(pin_device, d_inflate_ahead_ptr, d_inflate_ahead_cap);
        }
        d_inflate_ahead_ptr = nullptr; d_inflate_ahead_cap = 0;
    }
    uint64_t device_members = 0;   // members the device has inflated and verified for this handle
    bool verify = true;   // inflate whole members and check their CRC32 (svx_bam_set_verify); the default
    Pool pool;

    // copy of the page-locked pool in the pinned device's HBM, uploaded while svx_bam_load assembles the pool
    // (svx_bam_device_pool); the buffer lives as long as the handle, the stream is the process's (upload_stream)
    uint32_t* d_cigar = nullptr;
    uint64_t d_cap = 0, d_ops = 0;   // words allocated / words of the current pool
    hipStream_t up_stream = nullptr; // not owned
    hipEvent_t up_done = nullptr;    // recorded behind the pool's last part
    bool d_valid = false;
    int d_device = -1;

    void free_cigar() {
        if (d_valid) {  // the upload reads the page-locked pool: it must have ended before the pool goes
            (void)hipEventSynchronize(up_done);
            d_valid = false;
        }
        if (!cigar) return;
        if (cigar_pinned) host_buffer_done(pin_device, cigar, cigar_pinned_bytes);
        else free(cigar);
        cigar = nullptr;
        cigar_pinned = false;
        cigar_pinned_bytes = 0;
    }
    void free_device() {
        take_inflate_ahead();
        d_valid = false;
        if (d_inflate) dev_buffer_done(pin_device, d_inflate, d_inflate_cap);
        d_inflate = nullptr; d_inflate_cap = 0;
        if (d_cigar) dev_buffer_done(d_device, d_cigar, (size_t)d_cap * 4);
        if (up_done) (void)hipEventDestroy(up_done);
        d_cigar = nullptr; up_done = nullptr; up_stream = nullptr;
        d_cap = 0; d_ops = 0; d_device = -1;
    }
    static bool device_pool_off() {
        static const bool off = [] { const char* e = getenv("SVX_BAM_DEVICE_POOL"); return e && e[0] == '0'; }();
        return off;
    }
    // room for n_words of the pool on pin_device, the copy stream and an event; false: no device copy this time
    bool prepare_device(uint64_t n_words) {
        if (device_pool_off() || pin_device < 0) return false;
        if (d_device != pin_device) free_device();
        d_device = pin_device;
        up_stream = upload_stream(pin_device);
        bool ok = up_stream != nullptr;
        if (ok && !up_done) ok = hipEventCreateWithFlags(&up_done, hipEventDisableTiming) == hipSuccess;
        if (ok && d_cap < n_words) {
            if (d_cigar) dev_buffer_done(d_device, d_cigar, (size_t)d_cap * 4);
            d_cigar = nullptr; d_cap = 0;
            const uint64_t want = (n_words + (n_words >> 3) + 0x3FFFFu) & ~0x3FFFFull;  // 1 MiB steps, an eighth of slack
            size_t got = 0;
            void* p = dev_buffer(pin_device, (size_t)want * 4, &got);
            ok = p != nullptr;
            if (ok) { d_cigar = static_cast<uint32_t*>(p); d_cap = got / 4; }
        }
        if (!ok) {
            (void)hipGetLastError();
            free_device();
        }
        return ok;
    }
};

static int fail(svx_bam* b, int rc, const std::string& msg) {
    if (b) b->err = msg;
    return rc;
}

extern "C" int svx_bam_open(const char* path, int n_threads, svx_bam** out, char* err, size_t err_cap) {
    if (out) *out = nullptr;
    auto report = [&](const std::string& m, int rc) {
        if (err && err_cap) snprintf(err, err_cap, "%s", m.c_str());
        return rc;
    };
    if (!path || !out) return report("null argument", SVX_E_INVALID);
    svx_bam* b = new (std::nothrow) svx_bam();
    if (!b) return report("out of memory", SVX_E_NOMEM);
    b->path = path;
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(64u, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
    b->n_threads = n_threads;
    {
        const char* v = getenv("SVX_BAM_VERIFY");
        b->verify = !(v && v[0] == '0');  // default on: what htslib does under the reference
    }
    b->fd = open(path, O_RDONLY);
    struct stat st;
    if (b->fd < 0 || fstat(b->fd, &st) != 0) {
        const std::string m = std::string("cannot open ") + path;
        svx_bam_close(b);
        return report(m, SVX_E_INVALID);
    }
    b->file.fsize = (uint64_t)st.st_size;
    if (b->file.fsize) {
        void* m = mmap(nullptr, b->file.fsize, PROT_READ, MAP_PRIVATE, b->fd, 0);
        if (m == MAP_FAILED) {
            svx_bam_close(b);
            return report("mmap failed", SVX_E_NOMEM);
        }
        b->file.map = static_cast<const uint8_t*>(m);
    }
    // ---- header
    Inflater inf;
    Cursor c(&b->file, &inf);
    uint8_t head[12];
    VPos zero;
    bool ok = c.seek(zero) && c.read(head, 8) && memcmp(head, "BAM\1", 4) == 0;
    int32_t l_text = 0, n_ref = 0;
    if (ok) {
        l_text = (int32_t)le32(head + 4);
        ok = l_text >= 0;
    }
    if (ok) {
        b->text.resize((size_t)l_text);
        ok = (l_text == 0 || c.read(&b->text[0], (size_t)l_text)) && c.read(head, 4);
        const size_t z = b->text.find('\0');
        if (z != std::string::npos) b->text.resize(z);
    }
    if (ok) {
        n_ref = (int32_t)le32(head);
        ok = n_ref >= 0;
    }
    for (int32_t r = 0; ok && r < n_ref; ++r) {
        ok = c.read(head, 4);
        const int32_t l_name = ok ? (int32_t)le32(head) : 0;
        ok = ok && l_name > 0 && l_name < (1 << 20);
        std::string name;
        if (ok) {
            name.resize((size_t)l_name);
            ok = c.read(&name[0], (size_t)l_name) && c.read(head, 4);
            name.resize(strlen(name.c_str()));
        }
        if (ok) {
            b->ref_names.push_back(name);
            b->ref_lens.push_back((int32_t)le32(head));
        }
    }
    if (!ok) {
        const std::string m = std::string(path) + " is not a BAM file (bad magic, truncated header or malformed BGZF)";
        svx_bam_close(b);
        return report(m, SVX_E_INVALID);
    }
    b->first_record = c.tell();
    b->blocks_inflated += inf.n_blocks;
    // ---- index
    // <file>.bai, <file without .bam>.bai, then the same two names with .csi (htslib's order)
    std::string stem = b->path;
    const size_t dot = stem.rfind('.');
    if (dot != std::string::npos) stem = stem.substr(0, dot);
    std::string bai, csi;
    for (const std::string& cand : {b->path + ".bai", stem + ".bai"})
        if (bai.empty() && file_exists(cand)) bai = cand;
    for (const std::string& cand : {b->path + ".csi", stem + ".csi"})
        if (csi.empty() && file_exists(cand)) csi = cand;
    if (!bai.empty() || !csi.empty()) {
        b->index_state = 2;
        std::vector<uint8_t> d;
        if (!bai.empty() ? (read_file(bai, &d) && parse_bai(d, n_ref, &b->refs))
                         : (read_file(csi, &d) && parse_csi(d, n_ref, &b->refs))) {
            // consistent with the file?  every point must address a member inside the file, and
            // the first placed record of the file must be the lowest indexed position
            bool good = true, any = false;
            VPos lowest;
            lowest.coff = ~0ull;
            {   // every index point against the member header it names: ~10^5 page touches, spread over the threads
                std::vector<uint64_t> pts;
                for (const RefIndex& R : b->refs) pts.insert(pts.end(), R.points.begin(), R.points.end());
                std::atomic<bool> bad(false);
                auto check = [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i) {
                        Blk blk;
                        const uint64_t v = pts[i], co = v >> 16;
                        const int rc = parse_block(b->file.map, b->file.fsize, co, &blk);
                        if (rc < 0 || (rc == 1 && (v & 0xFFFF)) || (rc == 0 && (v & 0xFFFF) > blk.isize)) bad.store(true);
                    }
                };
                const size_t nt = pts.size() < 4096 ? 1 : (size_t)std::max(1, std::min(b->n_threads, 16));
                if (nt == 1) {
                    check(0, pts.size());
                } else {
                    std::atomic<size_t> turn(0);  // (these threads then serve svx_bam_load and svx_bam_seq_slices)
                    b->pool.run((int)nt, [&] {
                        const size_t t = turn.fetch_add(1);
                        check(pts.size() * t / nt, pts.size() * (t + 1) / nt);
                    });
                }
                good = !bad.load();
            }
            for (const RefIndex& R : b->refs) {
                if (!R.points.empty()) {
                    any = true;
                    Cursor k(&b->file, &inf);
                    VPos p;
                    p.coff = R.lo >> 16;
                    p.uoff = (uint32_t)(R.lo & 0xFFFF);
                    if (!good || !k.seek(p)) { good = false; break; }
                    if (k.tell() < lowest) lowest = k.tell();
                }
            }
            if (good && any) good = (lowest == b->first_record);
            if (good && !any) {
                // an index without any chunk is only right for a file without placed records
                Cursor k(&b->file, &inf);
                good = k.seek(b->first_record);
                if (good && !k.eof) {
                    uint8_t fix[8];
                    good = k.read(fix, 8) && (int32_t)le32(fix + 4) < 0;
                }
            }
            if (good) b->index_state = 1;
        }
    } else if (file_exists(b->path + ".csi")) {
        b->index_state = 2;
    }
    *out = b;
    return SVX_OK;
}

extern "C" void svx_bam_close(svx_bam* b) {
    if (!b) return;
    b->free_cigar();
    b->free_device();
    if (b->file.map) munmap(const_cast<uint8_t*>(b->file.map), b->file.fsize);
    if (b->fd >= 0) close(b->fd);
    delete b;
}

extern "C" const char* svx_bam_last_error(const svx_bam* b) { return b ? b->err.c_str() : "null handle"; }

extern "C" int svx_bam_header(const svx_bam* b, const char** text, uint64_t* l_text, int32_t* n_ref) {
    if (!b) return SVX_E_INVALID;
    if (text) *text = b->text.data();
    if (l_text) *l_text = b->text.size();
    if (n_ref) *n_ref = (int32_t)b->ref_names.size();
    return SVX_OK;
}

extern "C" int svx_bam_reference(const svx_bam* b, int32_t tid, const char** name, int32_t* length) {
    if (!b || tid < 0 || (size_t)tid >= b->ref_names.size()) return SVX_E_INVALID;
    if (name) *name = b->ref_names[tid].c_str();
    if (length) *length = b->ref_lens[tid];
    return SVX_OK;
}

extern "C" int svx_bam_index_state(const svx_bam* b) { return b ? b->index_state : 0; }

extern "C" int svx_bam_set_verify(svx_bam* b, int on) {
    if (!b) return SVX_E_INVALID;
    b->verify = on != 0;
    return SVX_OK;
}

extern "C" int svx_bam_set_device_inflate(svx_bam* b, int percent) {
    if (!b || percent < 0 || percent > 100) return SVX_E_INVALID;
    b->inflate_pct = percent;
    return SVX_OK;
}

extern "C" int svx_bam_set_device_inflate_min(svx_bam* b, uint32_t members) {
    if (!b) return SVX_E_INVALID;
    b->inflate_min_members = members;
    return SVX_OK;
}

extern "C" int svx_bam_set_defer_verify(svx_bam* b, int on) {
    if (!b) return SVX_E_INVALID;
    b->defer_verify = on != 0;
    return SVX_OK;
}

// the members in `coffs` inflated whole and checked (CRC32, ISIZE) by the handle's threads; true: all good (they then count
// as verified)
static bool verify_members_on_host(svx_bam* b, const std::vector<uint64_t>& coffs) {
    if (coffs.empty()) return true;
    std::atomic<size_t> next(0);
    std::atomic<bool> bad(false);
    std::atomic<uint64_t> n_inflated(0);
    auto check = [&]() {
        Inflater inf;
        std::vector<uint8_t> buf(65536);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= coffs.size() || bad.load()) break;
            Blk blk;
            if (parse_block(b->file.map, b->file.fsize, coffs[i], &blk) != 0 ||
                !inf.run(b->file.map + coffs[i] + blk.payload_off, blk.payload_len, buf.data(), blk.isize, blk.crc))
                bad.store(true);
        }
        n_inflated.fetch_add(inf.n_blocks);
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)b->n_threads, coffs.size() / 16 + 1));
    if (nt <= 1) check();
    else b->pool.run(nt, check);
    b->blocks_inflated += n_inflated.load();
    if (bad.load()) return false;
    std::vector<uint64_t> all;
    all.reserve(b->verified_members.size() + coffs.size());
    std::merge(b->verified_members.begin(), b->verified_members.end(), coffs.begin(), coffs.end(), std::back_inserter(all));
    all.erase(std::unique(all.begin(), all.end()), all.end());
    b->verified_members.swap(all);
    return true;
}

extern "C" int svx_bam_verify_pending(svx_bam* b) {
    if (!b) return SVX_E_INVALID;
    if (b->pending_members.empty()) return SVX_OK;
    std::vector<uint64_t> todo;
    todo.swap(b->pending_members);
    if (!verify_members_on_host(b, todo)) return fail(b, SVX_E_INVALID, "svx_bam_verify_pending: a BGZF member a record walk took bytes from is malformed or fails its CRC32");
    return SVX_OK;
}

extern "C" uint64_t svx_bam_pending_members(const svx_bam* b) { return b ? (uint64_t)b->pending_members.size() : 0; }

extern "C" int svx_bam_set_device_inflate_wait(svx_bam* b, uint32_t milliseconds) {
    if (!b) return SVX_E_INVALID;
    b->inflate_wait_ms = milliseconds;
    return SVX_OK;
}

extern "C" uint64_t svx_bam_device_members(const svx_bam* b) { return b ? b->device_members : 0; }

extern "C" int svx_bam_set_pinned_device(svx_bam* b, int device) {
    if (!b) return SVX_E_INVALID;
    b->pin_device = device;
    return SVX_OK;
}

extern "C" int svx_bam_contig_spans(const svx_bam* b, uint64_t* span) {
    if (!b || !span) return SVX_E_INVALID;
    if (b->index_state != 1) return SVX_E_INVALID;
    for (size_t r = 0; r < b->refs.size(); ++r) {
        const RefIndex& R = b->refs[r];
        // +1: a contig whose records all sit inside one BGZF member still counts as work
        span[r] = R.points.empty() ? 0 : ((R.hi >> 16) - (R.lo >> 16)) + 1;
    }
    return SVX_OK;
}

namespace {

struct Piece {
    VPos start, stop;
    bool have_stop = true;
};

// canonical position of a virtual offset (false: malformed)
bool canonical(const File* f, uint64_t v, VPos* out) {
    Inflater none;
    Cursor c(f, &none);
    VPos p;
    p.coff = v >> 16;
    p.uoff = (uint32_t)(v & 0xFFFF);
    if (!c.seek(p)) return false;
    *out = c.tell();
    return true;
}

}  // namespace

extern "C" int svx_bam_load(svx_bam* b, const int32_t* tids, int32_t n_tids) {
    if (!b || (tids == nullptr && n_tids > 0) || n_tids < 0) return SVX_E_INVALID;
    const int32_t n_ref = (int32_t)b->ref_names.size();
    std::vector<char> want;
    if (tids) {
        want.assign((size_t)n_ref, 0);
        for (int32_t i = 0; i < n_tids; ++i) {
            if (tids[i] < 0 || tids[i] >= n_ref) return fail(b, SVX_E_INVALID, "svx_bam_load: contig index out of range");
            want[tids[i]] = 1;
        }
    }
    // the device lanes (the pool's upload stream first) come up beside the walk: first load of the process only
    if (b->pin_device >= 0 && !svx_bam::device_pool_off()) lanes_start(b->pin_device, b->inflate_pct > 0);
    else if (b->pin_device >= 0 && b->inflate_pct > 0) lanes_start(b->pin_device, true);
    // ---- cut the requested ranges into pieces
    std::vector<Piece> pieces;
    bool filter_after = false;
    if (b->index_state == 1 && tids) {
        // boundaries of the requested contigs, in file order
        uint64_t total = 0;
        std::vector<std::pair<VPos, VPos>> ranges;  // canonical [lo, hi) per contig
        std::vector<std::vector<VPos>> pts;
        for (int32_t r = 0; r < n_ref; ++r) {
            if (!want[r] || b->refs[r].points.empty()) continue;
            std::vector<VPos> v;
            for (uint64_t x : b->refs[r].points) {
                VPos p;
                if (!canonical(&b->file, x, &p)) return fail(b, SVX_E_INVALID, "index points at a malformed BGZF member");
                v.push_back(p);
            }
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            total += v.back().coff - v.front().coff + 1;
            pts.push_back(v);
        }
        const uint64_t target = std::max<uint64_t>(1 << 16, total / ((uint64_t)b->n_threads * 8) + 1);
        for (const std::vector<VPos>& v : pts) {
            size_t s = 0;
            for (size_t k = 1; k < v.size(); ++k) {
                if (k + 1 == v.size() || v[k].coff - v[s].coff >= target) {
                    Piece pc;
                    pc.start = v[s];
                    pc.stop = v[k];
                    pieces.push_back(pc);
                    s = k;
                }
            }
        }
    } else if (b->index_state == 1 && !tids) {
        // every contig through the index (parallel), then whatever follows the last indexed
        // record (unplaced reads) sequentially
        std::vector<VPos> v;
        for (int32_t r = 0; r < n_ref; ++r)
            for (uint64_t x : b->refs[r].points) {
                VPos p;
                if (!canonical(&b->file, x, &p)) return fail(b, SVX_E_INVALID, "index points at a malformed BGZF member");
                v.push_back(p);
            }
        v.push_back(b->first_record);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        const uint64_t total = v.back().coff - v.front().coff + 1;
        const uint64_t target = std::max<uint64_t>(1 << 16, total / ((uint64_t)b->n_threads * 8) + 1);
        size_t s = 0;
        for (size_t k = 1; k < v.size(); ++k) {
            if (k + 1 == v.size() || v[k].coff - v[s].coff >= target) {
                Piece pc;
                pc.start = v[s];
                pc.stop = v[k];
                pieces.push_back(pc);
                s = k;
            }
        }
        Piece tail;
        tail.start = v.back();
        tail.have_stop = false;
        pieces.push_back(tail);
    } else {
        Piece pc;
        pc.start = b->first_record;
        pc.have_stop = false;
        pieces.push_back(pc);
        filter_after = tids != nullptr;
    }
    // ---- walk
    std::vector<Chunk> chunks(pieces.size());
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    std::atomic<uint64_t> inflated(0);
    const int nt = (int)std::min<size_t>((size_t)b->n_threads, pieces.size());
    // The page-locked pool (3-5 ms of page pinning for a genome's 12 MB) and the room for its device copy are made
    // BESIDE the walk's last pieces: the first worker that finds no piece left — every piece is handed out, all but at
    // most nt - 1 are done — extrapolates the pool's size from the finished pieces (equal compressed spans) plus an
    // eighth, and allocates; a pool that turns out larger is allocated again behind the walk, as before.
    struct Early {
        void* pinned = nullptr;
        uint64_t words = 0;
        size_t bytes = 0;
        int device = -1;
        ~Early() { if (pinned) host_buffer_done(device, pinned, bytes); }
    } early;
    early.device = b->pin_device;
    std::atomic<uint64_t> cig_done(0), pieces_done(0);
    std::atomic<bool> early_claimed(false);
    const bool early_ok = b->pin_device >= 0 && nt > 1 && !filter_after && !getenv("SVX_BAM_LATE_POOL");
    // the check of the members these walks touch, deferred to the device (svx_bam_set_defer_verify): only where a device leg
    // can take it — otherwise the walks check as they go, as ever
    const bool deferred = b->verify && b->defer_verify && b->inflate_pct > 0 && b->pin_device >= 0 && b->pin_device < kMaxLanes &&
                          g_inflate_launch && g_gather_launch;
    auto worker = [&]() {
        Inflater inf;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size() || failed.load()) break;
            if (!walk_records(&b->file, pieces[i].start, pieces[i].have_stop, pieces[i].stop, &inf, &chunks[i], b->verify && !deferred, deferred))
                failed.store(true);
            cig_done.fetch_add(chunks[i].cigar.size());
            pieces_done.fetch_add(1);
        }
        inflated.fetch_add(inf.n_blocks);
        if (early_ok && !failed.load() && !early_claimed.exchange(true)) {
            const uint64_t done = pieces_done.load(), words = cig_done.load();
            if (done * 4 >= pieces.size() * 3 && words) {
                const uint64_t est = words * pieces.size() / done;
                const uint64_t want = est + (est >> 3) + (64u << 10);
                if (hipSetDevice(b->pin_device) == hipSuccess && (early.pinned = host_buffer(b->pin_device, (size_t)want * 4, &early.bytes)) != nullptr) {
                    early.words = early.bytes / 4;
                    if (b->d_valid) (void)hipEventSynchronize(b->up_done);  // (the previous pool's upload, long over)
                    // from here on the device buffer may be freed, re-allocated or re-filled: it no longer holds the
                    // previous load's pool, whatever becomes of this load (a failing piece returns with the old columns
                    // still in place — svx_bam_device_pool must not hand out a buffer that merely has their op count)
                    b->d_valid = false;
                    (void)b->prepare_device(want);
                } else {
                    early.pinned = nullptr;
                }
            }
        }
    };
    if (getenv("SVX_BAM_DEBUG")) fprintf(stderr, "svx_bam_load: %zu pieces, %d threads\n", pieces.size(), nt);
    if (nt <= 1) {
        worker();
    } else {
        b->pool.run(nt, worker);
    }
    b->blocks_inflated += inflated.load();
    if (failed.load()) {
        std::string m = "BAM record walk failed";
        for (const Chunk& ch : chunks)
            if (!ch.err.empty()) { m = ch.err; break; }
        if (b->index_state == 1) {
            // a stale or foreign index: fall back to the sequential walk once
            b->index_state = 2;
            const int rc = svx_bam_load(b, tids, n_tids);
            if (rc == SVX_OK) b->err = "index ignored (" + m + ")";
            return rc;
        }
        return fail(b, SVX_E_INVALID, m);
    }
    // ---- concatenate
    uint64_t n = 0, n_cig = 0, n_name = 0, n_aux = 0, spanned = 0;
    {
        size_t add = 0;
        for (const Chunk& ch : chunks) add += ch.verified.size();
        if (add) {
            b->verified_members.reserve(b->verified_members.size() + add);
            for (const Chunk& ch : chunks) b->verified_members.insert(b->verified_members.end(), ch.verified.begin(), ch.verified.end());
            std::sort(b->verified_members.begin(), b->verified_members.end());
            b->verified_members.erase(std::unique(b->verified_members.begin(), b->verified_members.end()), b->verified_members.end());
        }
        add = 0;
        for (const Chunk& ch : chunks) add += ch.touched.size();
        if (add) {
            std::vector<uint64_t>& pm = b->pending_members;
            pm.reserve(pm.size() + add);
            for (const Chunk& ch : chunks) pm.insert(pm.end(), ch.touched.begin(), ch.touched.end());
            std::sort(pm.begin(), pm.end());
            pm.erase(std::unique(pm.begin(), pm.end()), pm.end());
            if (!b->verified_members.empty())
                pm.erase(std::remove_if(pm.begin(), pm.end(), [&](uint64_t c) {
                             return std::binary_search(b->verified_members.begin(), b->verified_members.end(), c); }), pm.end());
        }
    }
    for (const Chunk& ch : chunks) {
        spanned += ch.blocks_spanned;
        for (size_t i = 0; i < ch.tid.size(); ++i) {
            if (filter_after && (ch.tid[i] < 0 || !want[ch.tid[i]])) continue;
            ++n;
            n_cig += ch.n_cig[i];
            n_name += ch.name_len[i];
            n_aux += ch.aux_len[i];
        }
    }
    b->free_cigar();
    const size_t cig_bytes = std::max<uint64_t>(4, n_cig * 4);
    void* pinned = nullptr;
    if (early.pinned && early.words >= std::max<uint64_t>(1, n_cig)) {  // allocated beside the walk
        b->cigar = static_cast<uint32_t*>(early.pinned);
        b->cigar_pinned = true;
        b->cigar_pinned_bytes = early.bytes;
        early.pinned = nullptr;
    } else if (b->pin_device >= 0 && hipSetDevice(b->pin_device) == hipSuccess &&
               (pinned = host_buffer(b->pin_device, cig_bytes, &b->cigar_pinned_bytes)) != nullptr) {
        b->cigar = static_cast<uint32_t*>(pinned);
        b->cigar_pinned = true;
    } else {
        (void)hipGetLastError();
        b->cigar = static_cast<uint32_t*>(malloc(cig_bytes));
        if (!b->cigar) return fail(b, SVX_E_NOMEM, "out of memory (CIGAR pool)");
    }
    b->n_records = n;
    b->tid.clear(); b->pos.clear(); b->l_seq.clear(); b->ref_len.clear(); b->flag.clear(); b->mapq.clear();
    b->voffset.clear(); b->seq_coff.clear(); b->seq_uoff.clear(); b->sa_len.clear(); b->sa_off.clear();
    b->cigar_off.assign(1, 0); b->name_off.assign(1, 0); b->aux_off.assign(1, 0);
    b->names.clear(); b->aux.clear();
    b->tid.reserve(n); b->pos.reserve(n); b->l_seq.reserve(n); b->ref_len.reserve(n); b->flag.reserve(n);
    b->mapq.reserve(n); b->voffset.reserve(n); b->seq_coff.reserve(n); b->seq_uoff.reserve(n);
    b->sa_len.reserve(n); b->sa_off.reserve(n); b->cigar_off.reserve(n + 1); b->name_off.reserve(n + 1);
    b->aux_off.reserve(n + 1); b->names.reserve(n_name); b->aux.reserve(n_aux);
    // The pool's device copy travels while the pool is assembled: every kUploadWords of finished pool are handed to
    // the copy stream (DMA out of the page-locked pool, beside this thread's memcpy of the next chunk), so the last
    // part is on its way when the loop ends — svx_collect_batch finds the pool in HBM (svx_collect_in.part_dev)
    // instead of uploading it between the walk and the kernels.
    constexpr uint64_t kUploadWords = 128u << 10;  // 512 KiB: ~10 us of DMA behind the last chunk
    bool up = b->cigar_pinned && n_cig && b->prepare_device(n_cig);
    uint64_t up_at = 0;
    auto upload_to = [&](uint64_t end) {
        if (up && end > up_at &&
            hipMemcpyAsync(b->d_cigar + up_at, b->cigar + up_at, (end - up_at) * 4, hipMemcpyHostToDevice, b->up_stream) != hipSuccess) {
            (void)hipGetLastError();
            up = false;
        }
        up_at = end;
    };
    uint64_t cw = 0;
    for (const Chunk& ch : chunks) {
        size_t co = 0, no = 0, ao = 0;
        if (cw - up_at >= kUploadWords) upload_to(cw);
        for (size_t i = 0; i < ch.tid.size(); ++i) {
            const bool keep = !(filter_after && (ch.tid[i] < 0 || !want[ch.tid[i]]));
            if (keep) {
                b->tid.push_back(ch.tid[i]); b->pos.push_back(ch.pos[i]); b->l_seq.push_back(ch.l_seq[i]);
                b->ref_len.push_back(ch.ref_len[i]); b->flag.push_back(ch.flag[i]); b->mapq.push_back(ch.mapq[i]);
                b->voffset.push_back(ch.voffset[i]); b->seq_coff.push_back(ch.seq_coff[i]);
                b->seq_uoff.push_back(ch.seq_uoff[i]);
                if (ch.n_cig[i])  // (a chunk of records without CIGARs — unplaced reads — has no CIGAR buffer at all)
                    memcpy(b->cigar + cw, ch.cigar.data() + co, (size_t)ch.n_cig[i] * 4);
                cw += ch.n_cig[i];
                b->cigar_off.push_back(cw);
                b->names.insert(b->names.end(), ch.names.begin() + no, ch.names.begin() + no + ch.name_len[i]);
                b->name_off.push_back(b->names.size());
                b->sa_off.push_back(ch.sa_off[i] < 0 ? -1 : (int64_t)b->aux.size() + ch.sa_off[i]);
                b->sa_len.push_back(ch.sa_len[i]);
                b->aux.insert(b->aux.end(), ch.aux.begin() + ao, ch.aux.begin() + ao + ch.aux_len[i]);
                b->aux_off.push_back(b->aux.size());
            }
            co += ch.n_cig[i];
            no += ch.name_len[i];
            ao += ch.aux_len[i];
        }
    }
    upload_to(cw);
    if (up && hipEventRecord(b->up_done, b->up_stream) == hipSuccess) {
        b->d_valid = true;
        b->d_ops = n_cig;
    } else if (b->up_stream) {  // a part failed: nothing of the copy is handed out (and none of it is still reading the pool)
        (void)hipGetLastError();
        (void)hipStreamSynchronize(b->up_stream);
    }
    b->blocks_spanned = spanned;
    // the device leg's buffer for the sequence-slice call that follows a load of a whole file: its members' payloads are
    // about half of the file, their output twice that (1.6 x the share of the file), the token arena of the two-pass
    // kernels (a list per member, a member per 32 KiB of the file at most, SVX_INFLATE_ARENA_MEMBERS at most) and the
    // tables (a call that needs more allocates again)
    if (!tids && b->inflate_pct > 0 && b->pin_device >= 0 && b->pin_device < kMaxLanes && g_inflate_launch && !b->d_inflate &&
        !b->d_inflate_ahead.joinable() && b->file.fsize > (64u << 20)) {
        const double share_bytes = (double)b->file.fsize * b->inflate_pct / 100.0;
        const size_t lists = std::min<size_t>((size_t)(share_bytes / 32768.0) + 512, SVX_INFLATE_ARENA_MEMBERS);
        // (with the walks' check deferred to the leg, svx_bam_set_defer_verify: + the members the walks touched — 2.0 x)
        const size_t want = (size_t)(share_bytes * (b->defer_verify ? 2.0 : 1.6)) + lists * SVX_INFLATE_TOK_STRIDE * 8 + (32u << 20);
        svx_bam* h = b;
        b->d_inflate_ahead = std::thread([h, want] {
            if (hipSetDevice(h->pin_device) != hipSuccess) { (void)hipGetLastError(); return; }
            h->d_inflate_ahead_ptr = dev_buffer(h->pin_device, want, &h->d_inflate_ahead_cap);
        });
    }
    return SVX_OK;
}

extern "C" int svx_bam_device_pool(svx_bam* b, const uint32_t** d_cigar, uint64_t* n_ops, void** ready) {
    if (!b) return SVX_E_INVALID;
    if (d_cigar) *d_cigar = b->d_valid ? b->d_cigar : nullptr;
    if (n_ops) *n_ops = b->d_valid ? b->d_ops : 0;
    if (ready) *ready = b->d_valid ? static_cast<void*>(b->up_done) : nullptr;
    return SVX_OK;
}

extern "C" int svx_bam_device_pool_wait(svx_bam* b, double* waited_us) {
    if (!b) return SVX_E_INVALID;
    if (waited_us) *waited_us = 0.0;
    if (!b->d_valid) return SVX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipEventSynchronize(b->up_done) != hipSuccess) {
        (void)hipGetLastError();
        b->d_valid = false;
        return fail(b, SVX_E_HIP, "the CIGAR pool's upload failed");
    }
    if (waited_us) *waited_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return SVX_OK;
}

extern "C" int svx_bam_get_columns(const svx_bam* b, svx_bam_columns* o) {
    if (!b || !o) return SVX_E_INVALID;
    memset(o, 0, sizeof(*o));
    o->n_records = b->n_records;
    o->tid = b->tid.data(); o->pos = b->pos.data(); o->l_seq = b->l_seq.data(); o->ref_len = b->ref_len.data();
    o->flag = b->flag.data(); o->mapq = b->mapq.data();
    o->cigar_off = b->cigar_off.data(); o->cigar = b->cigar;
    o->name_off = b->name_off.data(); o->names = b->names.data();
    o->aux_off = b->aux_off.data(); o->aux = b->aux.data();
    o->sa_off = b->sa_off.data(); o->sa_len = b->sa_len.data();
    o->voffset = b->voffset.data();
    o->blocks_inflated = b->blocks_inflated;
    o->blocks_spanned = b->blocks_spanned;
    o->cigar_pinned = b->cigar_pinned ? 1 : 0;
    o->n_threads = b->n_threads;
    return SVX_OK;
}

extern "C" int svx_bam_seq_slices(svx_bam* b, const uint32_t* rec, const uint32_t* begin, const uint32_t* end,
                                  uint32_t n, const uint64_t* out_off, uint8_t* out) {
    if (!b) return SVX_E_INVALID;
    if (n == 0) return SVX_OK;
    if (!rec || !begin || !end || !out_off || !out) return SVX_E_INVALID;
    for (uint32_t i = 0; i < n; ++i)
        if (rec[i] >= b->n_records) return fail(b, SVX_E_INVALID, "svx_bam_seq_slices: record index out of range");
    static const char kLut[17] = "=ACMGRSVTWYHKDBN";
    // verify (svx_bam_set_verify / SVX_BAM_VERIFY=1): inflate every member a slice touches completely and check its
    // CRC32 (htslib's behaviour) instead of stopping at the last byte needed (a member inflated in part cannot be checked)
    const bool verify_all = b->verify;
    std::atomic<bool> failed(false);
    std::atomic<uint64_t> inflated(0), n_jobs(0), n_known(0);
    // A run of slices in three steps: (1) locate — walk the member headers (nothing is inflated) from each record's
    // SEQ start to the bytes of its slices: which members, and how far into each; (2) inflate those members two at a
    // time (Inflater::run_two: the decoder's rounds are latency-bound, two streams side by side cost 1.3x one);
    // (3) unpack the 4-bit codes of every piece from the two buffers.
    struct Job {   // one member and the prefix of it that is needed
        uint64_t coff;
        Blk blk;
        uint32_t upto;
    };
    struct Piece {  // the part of slice `slice` that lies in member `job`: packed bytes [uoff, uoff + n) of the member
        uint32_t slice, job, uoff, n;
        uint64_t first;  // index of its first packed byte within the record's SEQ field
    };
    struct State {
        Inflater inf[2];
        std::vector<uint8_t> buf[2];
        std::vector<Job> jobs;
        std::vector<Piece> pieces;
        uint32_t cur_rec = ~0u;
        uint64_t cur_byte = 0;  // bytes of the record's SEQ field already passed by the cursor
    };
    // the 4-bit codes of piece p (packed bytes at `packed`) → the ASCII bases of its slice
    auto unpack_piece = [&](const Piece& p, const uint8_t* packed) {
        const uint32_t i = p.slice;
        const uint32_t L = (uint32_t)b->l_seq[rec[i]];
        const uint32_t a = std::min(begin[i], L), e = std::max(a, std::min(end[i], L));
        const uint64_t k0 = std::max<uint64_t>(a, 2 * p.first), k1 = std::min<uint64_t>(e, 2 * (p.first + p.n));
        uint8_t* dst = out + out_off[i] + (k0 - a);
        for (uint64_t k = k0; k < k1; ++k) {
            const uint8_t by = packed[(k >> 1) - p.first];
            *dst++ = (uint8_t)kLut[(k & 1) ? (by & 15) : (by >> 4)];
        }
    };
    auto locate = [&](Cursor& c, State& st, uint32_t lo, uint32_t hi) -> bool {
        st.jobs.clear();
        st.pieces.clear();
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t r = rec[i];
            const uint32_t L = (uint32_t)b->l_seq[r];
            const uint32_t a = std::min(begin[i], L), e = std::max(a, std::min(end[i], L));
            if (out_off[i + 1] - out_off[i] != (uint64_t)(e - a)) { failed.store(true); return false; }
            if (e == a) continue;
            const uint64_t b0 = a >> 1, b1 = ((uint64_t)e + 1) >> 1;
            if (r != st.cur_rec || b0 < st.cur_byte) {
                VPos p;
                p.coff = b->seq_coff[r];
                p.uoff = b->seq_uoff[r];
                if (!c.seek(p)) { failed.store(true); return false; }
                st.cur_rec = r;
                st.cur_byte = 0;
            }
            if (!c.skip(b0 - st.cur_byte)) { failed.store(true); return false; }
            for (uint64_t at = b0; at < b1;) {
                if (c.eof || c.bad) { failed.store(true); return false; }
                const uint32_t take = (uint32_t)std::min<uint64_t>(b1 - at, c.blk.isize - c.uoff);
                if (st.jobs.empty() || st.jobs.back().coff != c.coff) st.jobs.push_back(Job{c.coff, c.blk, 0});
                Job& j = st.jobs.back();
                j.upto = std::max(j.upto, c.uoff + take);
                st.pieces.push_back(Piece{i, (uint32_t)(st.jobs.size() - 1), c.uoff, take, at});
                at += take;
                if (!c.skip(take)) { failed.store(true); return false; }
            }
            st.cur_byte = b1;
        }
        return true;
    };
    auto work_on = [&](Cursor& c, State& st, uint32_t lo, uint32_t hi) {
        if (!locate(c, st, lo, hi)) return;
        size_t pc = 0;
        for (size_t j = 0; j < st.jobs.size() && !failed.load(); j += 2) {
            const size_t nj = std::min<size_t>(2, st.jobs.size() - j);
            const uint8_t* in[2] = {nullptr, nullptr};
            size_t in_len[2] = {0, 0}, isize[2] = {0, 0}, want[2] = {0, 0};
            uint32_t crc[2] = {0, 0};
            for (size_t k = 0; k < nj; ++k) {
                const Job& jb = st.jobs[j + k];
                if (st.buf[k].empty()) st.buf[k].resize(65536);
                in[k] = b->file.map + jb.coff + jb.blk.payload_off;
                in_len[k] = jb.blk.payload_len;
                isize[k] = jb.blk.isize;
                // (a member a record walk has already inflated whole and checked is not checked a second time)
                const bool known = verify_all && std::binary_search(b->verified_members.begin(), b->verified_members.end(), jb.coff);
                want[k] = (verify_all && !known) ? jb.blk.isize : jb.upto;
                n_jobs.fetch_add(1, std::memory_order_relaxed);
                if (known) n_known.fetch_add(1, std::memory_order_relaxed);
                crc[k] = jb.blk.crc;
            }
            if (!Inflater::run_two(st.inf, in, in_len, st.buf, isize, want, crc, nj)) { failed.store(true); return; }
            for (; pc < st.pieces.size() && st.pieces[pc].job < j + nj; ++pc) {
                const Piece& p = st.pieces[pc];
                unpack_piece(p, st.buf[p.job - j].data() + p.uoff);
            }
        }
    };
    auto work = [&](uint32_t lo, uint32_t hi) {
        Inflater none;
        Cursor c(&b->file, &none);
        State st;
        for (uint32_t at = lo; at < hi && !failed.load(); at += 32) work_on(c, st, at, std::min(hi, at + 32));
        inflated.fetch_add(st.inf[0].n_blocks + st.inf[1].n_blocks);
    };
    const uint32_t nt = (uint32_t)std::max(1, std::min<int>(b->n_threads, (int)(n / 16 + 1)));
    bool pending_on_leg = false;  // the device leg has checked the members the record walks left unchecked (svx_bam_set_defer_verify)
    if (nt <= 1) {
        work(0, n);
    } else {
        // runs of 32 slices handed out on demand: the node is shared, and with one fixed share per thread the call
        // lasts as long as the unluckiest thread (same slices, 64 threads: 17 to 88 ms from one call to the next)
        constexpr uint32_t kRun = 32;
        std::atomic<uint32_t> next(0);
        const bool debug = getenv("SVX_BAM_DEBUG") != nullptr;
        std::atomic<int64_t> cpu_sum(0), cpu_max(0), wall_max(0), start_max(0);
        const auto t_call = std::chrono::steady_clock::now();
        auto since_call_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };

        // ---- the device leg (svx_bam_set_device_inflate): the members of the first n_g slices are inflated and verified
        // by svx_inflate.hip's kernels (a wave per member: 13 ms for a full-size call's 14 k members) while the handle's
        // threads take the other slices; the members the record walks left unchecked ride along (svx_bam_set_defer_verify).
        // The leg takes its share in one set of launches and is collected when the threads are through.  In order: (1) the threads locate the leg's runs (member headers only), (2) the payloads
        // travel through a ring of page-locked slots (the threads fill, the lane's stream copies), (3) inflate kernel,
        // a gather of the slices' packed bytes, (4) the threads' own share, (5) read-back, statuses, unpacking.
        struct RunPlan {
            std::vector<Job> jobs;
            std::vector<Piece> pieces;
        };
        uint32_t n_g = 0;
        InflateLane* lane = nullptr;
        std::unique_lock<std::mutex> lane_lock;
        const bool leg_wanted = verify_all && b->inflate_pct > 0 && b->pin_device >= 0 && b->pin_device < kMaxLanes && n >= 2048 &&
                                g_inflate_launch && g_gather_launch;
        if (leg_wanted && !g_lanes[b->pin_device].inflate_up.load()) {
            // a fresh process: the lanes were started beside the first load and may be a few milliseconds away — worth a
            // short wait (the leg takes ~40 % of the call's CPU seconds), not a long one
            DeviceLanes& L = g_lanes[b->pin_device];
            lanes_start(b->pin_device, true);
            std::unique_lock<std::mutex> lock(L.mu);
            L.cv.wait_for(lock, std::chrono::milliseconds(30), [&] { return L.inflate_tried; });
        }
        if (leg_wanted && g_lanes[b->pin_device].inflate_up.load()) {
            // one of the device's lanes; all taken (more calls in flight than lanes: a process that handles several
            // samples at once): the threads take everything — or, where the caller would rather wait than spend the CPU
            // seconds (svx_bam_set_device_inflate_wait), a sleeping wait for the first lane that comes free
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(b->inflate_wait_ms);
            for (;;) {
                for (int k = 0; k < g_lanes[b->pin_device].n_inflate && !lane; ++k) {
                    std::unique_lock<std::mutex> l(g_lanes[b->pin_device].inflate[k].busy, std::try_to_lock);
                    if (l.owns_lock()) {
                        lane = &g_lanes[b->pin_device].inflate[k];
                        lane_lock = std::move(l);
                    }
                }
                if (lane || std::chrono::steady_clock::now() >= give_up) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (lane) n_g = (uint32_t)((uint64_t)n * (uint64_t)b->inflate_pct / 100 / kRun * kRun);
        }
        std::vector<RunPlan> plans(n_g / kRun);
        std::vector<uint64_t> g_src_off, g_dst_off;   // per piece of the leg: where its packed bytes lie in d_out / in the read-back
        std::vector<uint32_t> g_len;
        std::vector<uint32_t> g_status;
        std::vector<uint8_t> g_packed;
        uint32_t g_members = 0, g_extra = 0;  // (g_extra: members of the deferred check among them, svx_bam_set_defer_verify)
        bool leg_running = false;
        uint64_t leg_o_status = 0, leg_o_packed = 0;  // where the leg's statuses and packed bytes lie in d_inflate
        double t_located = 0, t_staged = 0, t_launched = 0;
        if (n_g) {
            bool ok = hipSetDevice(b->pin_device) == hipSuccess;
            // (1) locate
            std::atomic<uint32_t> next_run(0);
            auto locate_runs = [&]() {
                Inflater none;
                Cursor c(&b->file, &none);
                State st;
                for (;;) {
                    const uint32_t r = next_run.fetch_add(1);
                    if (r >= plans.size() || failed.load()) break;
                    if (!locate(c, st, r * kRun, (r + 1) * kRun)) break;
                    plans[r].jobs = st.jobs;
                    plans[r].pieces = st.pieces;
                }
            };
            b->pool.run((int)nt, locate_runs);
            t_located = since_call_ms();
            ok = ok && !failed.load();
            // member and piece tables; device memory
            std::vector<uint64_t> in_off, out_off_m;
            std::vector<uint32_t> in_len, isz, crc;
            std::vector<const uint8_t*> src;
            uint64_t in_bytes = 0, out_bytes = 0, packed_bytes = 0;
            for (const RunPlan& pl : plans) {
                const uint32_t base = (uint32_t)in_off.size();
                for (const Job& jb : pl.jobs) {
                    in_off.push_back(in_bytes);
                    in_len.push_back(jb.blk.payload_len);
                    isz.push_back(jb.blk.isize);
                    crc.push_back(jb.blk.crc);
                    out_off_m.push_back(out_bytes);
                    src.push_back(b->file.map + jb.coff + jb.blk.payload_off);
                    in_bytes += ((uint64_t)jb.blk.payload_len + 3) & ~3ull;
                    out_bytes += ((uint64_t)jb.blk.isize + 8 + 15) & ~15ull;
                }
                for (const Piece& pc : pl.pieces) {
                    g_src_off.push_back(out_off_m[base + pc.job] + pc.uoff);
                    g_len.push_back(pc.n);
                    g_dst_off.push_back(packed_bytes);
                    packed_bytes += pc.n;
                }
            }
            // the members the record walks took bytes from without checking them (svx_bam_set_defer_verify) ride along: whole
            // members with no piece to gather — those that are not among the leg's own already
            if (!b->pending_members.empty()) {
                std::vector<uint64_t> own;
                own.reserve(in_off.size());
                for (const RunPlan& pl : plans)
                    for (const Job& jb : pl.jobs) own.push_back(jb.coff);
                std::sort(own.begin(), own.end());
                for (const uint64_t coff : b->pending_members) {
                    if (std::binary_search(own.begin(), own.end(), coff)) continue;
                    Blk blk;
                    if (parse_block(b->file.map, b->file.fsize, coff, &blk) != 0) { failed.store(true); ok = false; break; }
                    in_off.push_back(in_bytes);
                    in_len.push_back(blk.payload_len);
                    isz.push_back(blk.isize);
                    crc.push_back(blk.crc);
                    out_off_m.push_back(out_bytes);
                    src.push_back(b->file.map + coff + blk.payload_off);
                    in_bytes += ((uint64_t)blk.payload_len + 3) & ~3ull;
                    out_bytes += ((uint64_t)blk.isize + 8 + 15) & ~15ull;
                    ++g_extra;
                }
            }
            g_members = (uint32_t)in_off.size();
            // A set of launches costs one member's latency on the device (3-4 ms) and the leg its staging: below a few
            // hundred members the threads are through sooner (svx_bam_set_device_inflate_min, default 500; config 5's
            // 1 200-member calls: the same wall-clock on the device for 0.6 of 1.0 CPU-seconds).
            if (g_members < b->inflate_min_members) ok = false;
            const uint32_t n_pc = (uint32_t)g_len.size();
            auto up256 = [](uint64_t x) { return (x + 255) & ~255ull; };
            const uint64_t o_in = 0, o_out = up256(in_bytes + 8), o_tab = o_out + up256(out_bytes + 16);
            const uint64_t tab_bytes = (uint64_t)g_members * 28 + (uint64_t)n_pc * 20 + 256;
            // the two-pass kernels' token lists: an arena for SVX_INFLATE_ARENA_MEMBERS members at a time (svx_inflate_dev.h)
            const uint32_t arena_members = std::min<uint32_t>(g_members, SVX_INFLATE_ARENA_MEMBERS);
            const uint64_t o_status = o_tab + up256(tab_bytes), o_packed = o_status + up256((uint64_t)g_members * 4);
            const uint64_t o_ntok = o_packed + up256(packed_bytes + 8), o_tok = o_ntok + up256((uint64_t)g_members * 4);
            const uint64_t need = o_tok + up256((uint64_t)arena_members * SVX_INFLATE_TOK_STRIDE * 8);
            if (ok && g_members && n_pc) {
                b->take_inflate_ahead();
                if (b->d_inflate_cap < need) {
                    if (b->d_inflate) dev_buffer_done(b->pin_device, b->d_inflate, b->d_inflate_cap);
                    b->d_inflate = nullptr;
                    b->d_inflate_cap = 0;
                    size_t got = 0;
                    void* pdev = dev_buffer(b->pin_device, need + (need >> 3), &got);
                    ok = pdev != nullptr;
                    if (ok) { b->d_inflate = static_cast<uint8_t*>(pdev); b->d_inflate_cap = got; }
                }
            } else {
                ok = false;
            }
            // the tables as one blob: in_off | out_off | src_off | dst_off (u64) | in_len | isize | crc | len (u32)
            std::vector<uint8_t> blob;
            uint64_t t_in_off = 0, t_out_off = 0, t_src_off = 0, t_dst_off = 0, t_in_len = 0, t_isz = 0, t_crc = 0, t_len = 0;
            if (ok) {
                auto put = [&](const void* ptr, size_t bytes) { const uint64_t at = blob.size(); blob.insert(blob.end(), (const uint8_t*)ptr, (const uint8_t*)ptr + bytes); return at; };
                t_in_off = put(in_off.data(), (size_t)g_members * 8);
                t_out_off = put(out_off_m.data(), (size_t)g_members * 8);
                t_src_off = put(g_src_off.data(), (size_t)n_pc * 8);
                t_dst_off = put(g_dst_off.data(), (size_t)n_pc * 8);
                t_in_len = put(in_len.data(), (size_t)g_members * 4);
                t_isz = put(isz.data(), (size_t)g_members * 4);
                t_crc = put(crc.data(), (size_t)g_members * 4);
                t_len = put(g_len.data(), (size_t)n_pc * 4);
                ok = blob.size() <= tab_bytes &&
                     hipMemcpyAsync(b->d_inflate + o_tab, blob.data(), blob.size(), hipMemcpyHostToDevice, lane->stream) == hipSuccess;
            }
            // (2) payloads through the ring: batch i = members [cut[i], cut[i + 1]) in slot i % kRingSlots
            std::vector<uint32_t> cut(1, 0);
            for (uint32_t m = 0; ok && m < g_members; ++m) {
                const uint64_t end_m = (m + 1 < g_members ? in_off[m + 1] : in_bytes);
                if (end_m - in_off[cut.back()] > kSlotBytes) cut.push_back(m);
            }
            cut.push_back(g_members);
            std::atomic<uint32_t> next_batch(0), phase_end(0), slot_gen[kRingSlots];
            for (int q = 0; q < kRingSlots; ++q) slot_gen[q].store(0);
            std::atomic<bool> stage_failed(false);
            std::atomic<int64_t> st_turn_us(0), st_event_us(0), st_copy_us(0), st_enqueue_us(0);  // (SVX_BAM_DEBUG: where staging goes)
            auto stage = [&]() {
                if (hipSetDevice(b->pin_device) != hipSuccess) { stage_failed.store(true); return; }
                auto now_us = [] { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
                for (;;) {
                    const uint32_t i = next_batch.fetch_add(1);
                    if (i >= phase_end.load()) break;
                    const uint32_t q = i % kRingSlots, gen = i / kRingSlots;
                    const int64_t t0 = debug ? now_us() : 0;
                    while (slot_gen[q].load(std::memory_order_acquire) != gen)  // batch i - kRingSlots has enqueued its copy
                        std::this_thread::sleep_for(std::chrono::microseconds(SVX_STAGE_POLL_US));
                    if (stage_failed.load()) { slot_gen[q].store(gen + 1, std::memory_order_release); continue; }
                    const int64_t t1 = debug ? now_us() : 0;
                    bool good = gen == 0 || hipEventSynchronize(lane->slot_done[q]) == hipSuccess;  // ... and the copy has read the slot
                    const int64_t t2 = debug ? now_us() : 0;
                    uint8_t* slot = lane->ring + (size_t)q * kSlotBytes;
                    const uint64_t b0 = in_off[cut[i]];
                    const uint64_t b1 = cut[i + 1] < g_members ? in_off[cut[i + 1]] : in_bytes;
                    for (uint32_t m = cut[i]; good && m < cut[i + 1]; ++m) memcpy(slot + (in_off[m] - b0), src[m], in_len[m]);
                    const int64_t t3 = debug ? now_us() : 0;
                    good = good && hipMemcpyAsync(b->d_inflate + o_in + b0, slot, b1 - b0, hipMemcpyHostToDevice, lane->stream) == hipSuccess &&
                           hipEventRecord(lane->slot_done[q], lane->stream) == hipSuccess;
                    if (debug) {
                        const int64_t t4 = now_us();
                        st_turn_us.fetch_add(t1 - t0); st_event_us.fetch_add(t2 - t1); st_copy_us.fetch_add(t3 - t2); st_enqueue_us.fetch_add(t4 - t3);
                    }
                    if (!good) stage_failed.store(true);
                    slot_gen[q].store(gen + 1, std::memory_order_release);
                }
            };
            // (2 + 3) the payloads through the ring, then the kernels, the gather and the event behind it — in one phase, or
            // (leg_phases_asked) in several with a phase's kernels on the lane's second stream beside the next phase's copies
            const uint32_t n_batches = (uint32_t)cut.size() - 1;
            const uint32_t n_phases = lane && lane->kstream ? std::max(1u, std::min(std::min<uint32_t>(leg_phases_asked(), kLegPhasesMax), n_batches)) : 1u;
            const hipStream_t ks = lane ? (n_phases > 1 ? lane->kstream : lane->stream) : nullptr;
            uint8_t* const d = b->d_inflate;
            const uint8_t* const tab = d + o_tab;
            for (uint32_t ph = 0; ok && ph < n_phases; ++ph) {
                const uint32_t b0 = (uint32_t)((uint64_t)n_batches * ph / n_phases), b1 = (uint32_t)((uint64_t)n_batches * (ph + 1) / n_phases);
                next_batch.store(b0);
                phase_end.store(b1);
                // as many threads as the ring has slots: each batch finds its slot free or about to be — more threads would
                // only wait for their turn (spinning through the CPU quota: measured, 3.4 instead of 2.9 CPU-seconds per run)
                b->pool.run((int)std::min<uint32_t>(std::min<uint32_t>(nt, (uint32_t)kRingSlots), b1 - b0), stage);
                ok = !stage_failed.load();
                const uint32_t m0 = cut[b0], m1 = cut[b1];
                if (n_phases > 1)
                    ok = ok && hipEventRecord(lane->staged[ph], lane->stream) == hipSuccess && hipStreamWaitEvent(ks, lane->staged[ph], 0) == hipSuccess;
                ok = ok &&
                     g_inflate_launch(ks, d + o_in, (const uint64_t*)(tab + t_in_off) + m0, (const uint32_t*)(tab + t_in_len) + m0,
                                      (const uint32_t*)(tab + t_isz) + m0, (const uint32_t*)(tab + t_crc) + m0, m1 - m0, d + o_out,
                                      (const uint64_t*)(tab + t_out_off) + m0, (uint32_t*)(d + o_status) + m0, (uint32_t*)(d + o_ntok) + m0,
                                      d + o_tok, arena_members) == 0;
                if (ph == 0) t_staged = since_call_ms();  // (the first phase's: what the kernels wait for at least)
            }
            if (debug)
                fprintf(stderr, "svx_bam_seq_slices: staging %zu batches in %u phase(s), %.1f MB: thread-ms waiting for the slot's turn %.1f, for its last copy %.1f, "
                        "copying members in %.1f, enqueueing %.1f\n", cut.size() - 1, n_phases, in_bytes / 1e6, st_turn_us.load() / 1e3, st_event_us.load() / 1e3,
                        st_copy_us.load() / 1e3, st_enqueue_us.load() / 1e3);
            if (ok) {
                ok = g_gather_launch(ks, d + o_out, (const uint64_t*)(tab + t_src_off), (const uint32_t*)(tab + t_len),
                                     (const uint64_t*)(tab + t_dst_off), n_pc, d + o_packed) == 0 &&
                     hipEventRecord(lane->done, ks) == hipSuccess;
            }
            t_launched = since_call_ms();
            if (ok) {
                leg_running = true;
                pending_on_leg = true;  // (taken back below if the leg fails behind its launch)
                g_status.resize(g_members);
                g_packed.resize(packed_bytes + 8);
                next.store(n_g);  // the threads take the slices behind the leg's
            } else {
                (void)hipGetLastError();
                if (lane) { (void)hipStreamSynchronize(lane->stream); if (lane->kstream) (void)hipStreamSynchronize(lane->kstream); }  // nothing of a failed leg is still at work
                n_g = 0;      // the threads take everything
                g_members = 0;
            }
            leg_o_status = o_status;
            leg_o_packed = o_packed;
        }
        uint32_t pull_hi = n;  // the threads take the slices [next, pull_hi)
        auto pull = [&]() {
            timespec c0;
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
            const auto w0 = std::chrono::steady_clock::now();
            Inflater none;
            Cursor c(&b->file, &none);  // only walks member headers here
            State st;
            for (;;) {
                const uint32_t lo = next.fetch_add(kRun);
                if (lo >= pull_hi || failed.load()) break;
                work_on(c, st, lo, std::min(pull_hi, lo + kRun));
            }
            inflated.fetch_add(st.inf[0].n_blocks + st.inf[1].n_blocks);
            if (debug) {  // wall time far above CPU time: the thread was waiting for a core, not working
                timespec c1;
                clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
                const int64_t cpu = (c1.tv_sec - c0.tv_sec) * 1000000 + (c1.tv_nsec - c0.tv_nsec) / 1000;
                const int64_t wall = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - w0).count();
                const int64_t late = std::chrono::duration_cast<std::chrono::microseconds>(w0 - t_call).count();
                cpu_sum.fetch_add(cpu);
                for (int64_t v = cpu_max.load(); cpu > v && !cpu_max.compare_exchange_weak(v, cpu);) {}
                for (int64_t v = wall_max.load(); wall > v && !wall_max.compare_exchange_weak(v, wall);) {}
                for (int64_t v = start_max.load(); late > v && !start_max.compare_exchange_weak(v, late);) {}
            }
        };
        b->pool.run((int)nt, pull);
        // (5) the leg's results
        if (leg_running) {
            const double t_cpu_done = since_call_ms();
            bool ok = hipEventSynchronize(lane->done) == hipSuccess;
            const double t_dev_done = since_call_ms();
            ok = ok && hipMemcpy(g_status.data(), b->d_inflate + leg_o_status, (size_t)g_members * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(g_packed.data(), b->d_inflate + leg_o_packed, g_packed.size() - 8, hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) {
                // a device error BEHIND the launch (every failure in front of it falls back to the threads above): the host
                // decoder takes the leg's slices now — a transient device error must not fail a run the host can finish,
                // and the results never depend on the share
                (void)hipGetLastError();
                (void)hipStreamSynchronize(lane->stream);  // nothing of the failed leg is still reading the ring
                if (lane->kstream) (void)hipStreamSynchronize(lane->kstream);
                (void)hipGetLastError();
                next.store(0);
                pull_hi = n_g;
                b->pool.run((int)nt, pull);
                g_members = 0;
                pending_on_leg = false;
            }
            for (uint32_t m = 0; m < g_members; ++m)
                if (g_status[m] != 0) failed.store(true);  // malformed stream, wrong length or CRC32: as the host decoder judges
            if (ok && !failed.load()) {
                // unpack: piece k of the leg (plans in order) lies at g_dst_off[k] of the read-back
                std::vector<uint32_t> first_piece(plans.size() + 1, 0);
                for (size_t r = 0; r < plans.size(); ++r) first_piece[r + 1] = first_piece[r] + (uint32_t)plans[r].pieces.size();
                std::atomic<uint32_t> next_plan(0);
                auto unpack = [&]() {
                    for (;;) {
                        const uint32_t r = next_plan.fetch_add(1);
                        if (r >= plans.size()) break;
                        for (size_t k = 0; k < plans[r].pieces.size(); ++k)
                            unpack_piece(plans[r].pieces[k], g_packed.data() + g_dst_off[first_piece[r] + k]);
                    }
                };
                b->pool.run((int)std::min<uint32_t>(nt, (uint32_t)plans.size()), unpack);
            }
            inflated.fetch_add(g_members);
            b->device_members += g_members;
            if (debug)
                fprintf(stderr, "svx_bam_seq_slices: device leg: %u of %u slices, %u members (%u of them for the record walks' check), %.1f MB packed; located +%.1f ms, staged +%.1f, "
                        "launched +%.1f, threads done +%.1f, device done +%.1f, all +%.1f\n", n_g, n, g_members, g_extra, g_packed.size() / 1e6,
                        t_located, t_staged, t_launched, t_cpu_done, t_dev_done, since_call_ms());
        }
        if (debug)
            fprintf(stderr, "svx_bam_seq_slices: %u slices, %u threads: call %.1f ms; per thread: last start +%.1f ms, longest wall %.1f ms, "
                    "longest cpu %.1f ms, cpu sum %.1f ms\n", n, nt,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(),
                    start_max.load() / 1e3, wall_max.load() / 1e3, cpu_max.load() / 1e3, cpu_sum.load() / 1e3);
    }
    b->blocks_inflated += inflated.load();
    if (!failed.load() && !b->pending_members.empty()) {
        std::vector<uint64_t> todo;
        todo.swap(b->pending_members);
        if (pending_on_leg) {  // every status of the leg was 0: they count as verified
            std::vector<uint64_t> all;
            all.reserve(b->verified_members.size() + todo.size());
            std::merge(b->verified_members.begin(), b->verified_members.end(), todo.begin(), todo.end(), std::back_inserter(all));
            all.erase(std::unique(all.begin(), all.end()), all.end());
            b->verified_members.swap(all);
        } else if (!verify_members_on_host(b, todo)) {
            failed.store(true);
        }
    }
    if (getenv("SVX_BAM_DEBUG"))
        fprintf(stderr, "svx_bam_seq_slices: %llu member inflations, %llu of them of members a record walk had verified (prefix only); "
                "%zu members verified by walks\n", (unsigned long long)n_jobs.load(), (unsigned long long)n_known.load(),
                b->verified_members.size());
    if (failed.load()) return fail(b, SVX_E_INVALID, "svx_bam_seq_slices: bad slice bounds or malformed BGZF data");
    return SVX_OK;
}

extern "C" int svx_inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, const uint64_t* stops,
                               uint32_t n_stops, uint64_t* n_out) {
    if ((!in && in_len) || (!out && cap) || (!stops && n_stops) || !n_out) return SVX_E_INVALID;
    *n_out = 0;
    std::unique_ptr<svx_inflate::Stream> st(new svx_inflate::Stream());
    st->begin(in, in_len);
    for (uint32_t i = 0; i < n_stops; ++i) {
        const bool ok = st->run(out, cap, (size_t)std::min<uint64_t>(stops[i], cap), false);
        *n_out = st->produced();
        if (!ok) return SVX_E_INVALID;
    }
    const bool ok = st->run(out, cap, 0, true);
    *n_out = st->produced();
    return ok ? SVX_OK : SVX_E_INVALID;
}

extern "C" int svx_inflate_raw_pair(const uint8_t* in_a, size_t in_len_a, uint8_t* out_a, size_t cap_a, uint64_t stop_a,
                                    uint64_t* n_out_a, int* rc_a, const uint8_t* in_b, size_t in_len_b, uint8_t* out_b,
                                    size_t cap_b, uint64_t stop_b, uint64_t* n_out_b, int* rc_b) {
    if ((!in_a && in_len_a) || (!out_a && cap_a) || (!in_b && in_len_b) || (!out_b && cap_b) || !n_out_a || !n_out_b ||
        !rc_a || !rc_b)
        return SVX_E_INVALID;
    std::unique_ptr<svx_inflate::Stream> a(new svx_inflate::Stream()), b(new svx_inflate::Stream());
    a->begin(in_a, in_len_a);
    b->begin(in_b, in_len_b);
    bool ok_a = false, ok_b = false;
    const bool end_a = stop_a == ~0ull, end_b = stop_b == ~0ull;
    svx_inflate::Stream::run_pair(*a, out_a, cap_a, end_a ? 0 : (size_t)std::min<uint64_t>(stop_a, cap_a), end_a, &ok_a,
                                  *b, out_b, cap_b, end_b ? 0 : (size_t)std::min<uint64_t>(stop_b, cap_b), end_b, &ok_b);
    *n_out_a = a->produced();
    *n_out_b = b->produced();
    *rc_a = ok_a ? SVX_OK : SVX_E_INVALID;
    *rc_b = ok_b ? SVX_OK : SVX_E_INVALID;
    return SVX_OK;
}
