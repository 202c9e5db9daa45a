// svx_text.cpp — native text side of libsvx.so (include/svx_text.h): indexed-FASTA batch fetch and the VCF
// record lines.  Host C++ only.  Restates pysam.FastaFile.fetch as the reference uses it and the formatters of
// SVCandidate.py get_vcf_entry* / SVIM_COMBINE.py:428-477 (file:line cited per function below); written from
// the behaviour of those functions, one batch call over candidate columns instead of one Python call per record.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "svx.h"
#include "svx_text.h"
#include <sys/mman.h>

// ------------------------------------------------------------------------------------------ FASTA
struct svx_fasta {
    int fd = -1;
    // The file is read through a private read-only mapping (pread() where mapping fails).  Sixteen threads calling
    // pread() on one descriptor spend most of their time on the reference count of the one `struct file` they share:
    // 63 k intervals cost 0.27-0.31 CPU-seconds that way and 0.04-0.05 through the mapping (GPU box, full-size sample).
    const uint8_t* map = nullptr;
    size_t size = 0;
    std::vector<int64_t> length, offset;
    std::vector<int32_t> line_bases, line_width;
};

static void set_err(char* err, size_t cap, const char* fmt, const char* a) {
    if (err && cap) snprintf(err, cap, fmt, a);
}

extern "C" int svx_fasta_open(const char* path, int32_t n_refs, const int64_t* length, const int64_t* offset,
                              const int32_t* line_bases, const int32_t* line_width, svx_fasta** out, char* err,
                              size_t err_cap) {
    if (out) *out = nullptr;
    if (!path || !out || n_refs < 0 || (n_refs && (!length || !offset || !line_bases || !line_width))) {
        set_err(err, err_cap, "%s", "svx_fasta_open: bad argument");
        return SVX_E_INVALID;
    }
    svx_fasta* fa = new (std::nothrow) svx_fasta();
    if (!fa) return SVX_E_NOMEM;
    fa->fd = open(path, O_RDONLY);
    if (fa->fd < 0) {
        set_err(err, err_cap, "cannot open %s", path);
        delete fa;
        return SVX_E_INVALID;
    }
    struct stat st;
    if (fstat(fa->fd, &st) != 0) {
        set_err(err, err_cap, "cannot stat %s", path);
        close(fa->fd);
        delete fa;
        return SVX_E_INVALID;
    }
    fa->size = (size_t)st.st_size;
    if (fa->size && !getenv("SVX_FASTA_PREAD")) {  // (SVX_FASTA_PREAD=1: the pread() path, for A/B and tests)
        void* m = mmap(nullptr, fa->size, PROT_READ, MAP_PRIVATE, fa->fd, 0);
        if (m != MAP_FAILED) fa->map = static_cast<const uint8_t*>(m);
    }
    try {
        fa->length.assign(length, length + n_refs);
        fa->offset.assign(offset, offset + n_refs);
        fa->line_bases.assign(line_bases, line_bases + n_refs);
        fa->line_width.assign(line_width, line_width + n_refs);
    } catch (...) {
        svx_fasta_close(fa);
        return SVX_E_NOMEM;
    }
    *out = fa;
    return SVX_OK;
}

extern "C" void svx_fasta_close(svx_fasta* fa) {
    if (!fa) return;
    if (fa->map) munmap(const_cast<uint8_t*>(fa->map), fa->size);
    if (fa->fd >= 0) close(fa->fd);
    delete fa;
}

static inline uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// bases [start, end) of one sequence into dst (end already clipped to the sequence length): ONE pread of the byte
// range that holds them (line ends included), then the lines' bases are moved to dst
static bool fetch_one(const svx_fasta* fa, int32_t ref, int64_t start, int64_t end, bool upper, uint8_t* dst,
                      std::vector<uint8_t>& raw) {
    const int64_t lb = fa->line_bases[ref], lw = fa->line_width[ref], off = fa->offset[ref];
    if (lb <= 0 || lw < lb || off < 0) return false;
    const int64_t byte0 = off + (start / lb) * lw + start % lb;
    const int64_t byte1 = off + ((end - 1) / lb) * lw + (end - 1) % lb + 1;
    if (byte0 < 0 || byte1 < byte0 || (uint64_t)byte1 > fa->size) return false;
    const size_t n = (size_t)(byte1 - byte0);
    if (raw.size() < n) raw.resize(n);
    if (fa->map) {
        memcpy(raw.data(), fa->map + byte0, n);
    } else {
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fa->fd, raw.data() + got, n - got, (off_t)(byte0 + (int64_t)got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
    }
    if (upper)  // one pass over everything read (a line end stays what it is), instead of a short loop per line
        for (size_t i = 0; i < n; ++i) raw[i] = ascii_upper(raw[i]);
    const uint8_t* src = raw.data();
    int64_t left = end - start;
    int64_t take = std::min<int64_t>(lb - start % lb, left);  // the rest of the first line, then whole lines
    for (;;) {
        memcpy(dst, src, (size_t)take);
        dst += take;
        left -= take;
        if (left <= 0) break;
        src += take + (lw - lb);  // over the line end
        take = std::min<int64_t>(lb, left);
    }
    return true;
}

extern "C" int svx_fasta_fetch_batch(const svx_fasta* fa, const int32_t* ref, const int64_t* start, const int64_t* end,
                                     uint32_t n, int upper, const uint64_t* out_off, uint8_t* out, int n_threads) {
    if (!fa || (n && (!ref || !start || !end || !out_off))) return SVX_E_INVALID;
    const int32_t n_refs = (int32_t)fa->length.size();
    // validate everything before a byte is written
    for (uint32_t i = 0; i < n; ++i) {
        if (ref[i] < 0 || ref[i] >= n_refs || start[i] < 0 || end[i] < start[i]) return SVX_E_INVALID;
        const int64_t e = std::min(end[i], fa->length[ref[i]]);
        const int64_t len = e > start[i] ? e - start[i] : 0;
        if (out_off[i + 1] < out_off[i] || (int64_t)(out_off[i + 1] - out_off[i]) != len) return SVX_E_INVALID;
    }
    if (n && out_off[n] && !out) return SVX_E_INVALID;
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(16u, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
    if (n < 256) n_threads = 1;
    std::atomic<uint32_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        std::vector<uint8_t> raw;
        for (;;) {
            const uint32_t lo = next.fetch_add(512);
            if (lo >= n) break;
            const uint32_t hi = std::min<uint32_t>(n, lo + 512);
            for (uint32_t i = lo; i < hi; ++i) {
                const int64_t e = std::min(end[i], fa->length[ref[i]]);
                if (e > start[i] && !fetch_one(fa, ref[i], start[i], e, upper != 0, out + out_off[i], raw)) bad.store(1);
            }
        }
    };
    if (n_threads == 1) {
        work();
    } else {
        try {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
            for (std::thread& t : th) t.join();
        } catch (...) {
            return SVX_E_NOMEM;
        }
    }
    return bad.load() ? SVX_E_INVALID : SVX_OK;
}

// -------------------------------------------------------------------------------------------- VCF
namespace {

struct Out {
    std::string s;
    void num(int64_t v) {
        char buf[24];
        int n = snprintf(buf, sizeof(buf), "%lld", (long long)v);
        s.append(buf, (size_t)n);
    }
    void lit(const char* t) { s.append(t); }
    void bytes(const void* p, size_t n) {
        if (n) s.append((const char*)p, n);
    }
    void ch(char c) { s.push_back(c); }
};

struct Entry {
    int32_t rank;
    int64_t start, end;
    uint32_t idx;
};

inline void pool_str(Out& o, const char* pool, const int64_t* off, int64_t i) {
    o.bytes(pool + off[i], (size_t)(off[i + 1] - off[i]));
}

inline uint8_t complement_upper(uint8_t c) {
    // SVCandidate.py:106 — complement.get(base.upper(), base.upper()); the slice is upper-cased already
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        default: return c;
    }
}

// label index: DEL, INV, INS, DUP_TANDEM, DUP_INT, BND (SVIM_COMBINE.py:431-464, third tuple member)
const char* const LABELS[6] = {"DEL", "INV", "INS", "DUP_TANDEM", "DUP_INT", "BND"};
const int KIND_LABEL[9] = {0, 1, 2, 2, 3, 2, 4, 5, 5};

}  // namespace

namespace {

// every byte of [p, p + n) at file offset `at` (pwrite may write less than asked)
int pwrite_all(int fd, const char* p, size_t n, uint64_t at) {
    while (n) {
        const ssize_t k = pwrite(fd, p, n, (off_t)at);
        if (k < 0) {
            if (errno == EINTR) continue;
            return SVX_E_INVALID;
        }
        p += k; n -= (size_t)k; at += (uint64_t)k;
    }
    return SVX_OK;
}

// The record lines of write_final_vcf, formatted by a few threads over contiguous chunks of the sorted entries.
// fd < 0: joined into one malloc'ed buffer (*text).  fd >= 0: every thread writes its own chunk at its place in the
// file (pwrite at `file_at` + the sizes of the chunks in front of it) — no joined copy, and the copies into the page
// cache run side by side.
int vcf_format_impl(const svx_vcf_in* in, char** text, int fd, uint64_t file_at, uint64_t* n_bytes, uint64_t* n_lines) {
    if (!in || !n_bytes || (fd < 0 && !text)) return SVX_E_INVALID;
    if (text) *text = nullptr;
    *n_bytes = 0;
    if (n_lines) *n_lines = 0;
    const uint32_t ne = in->n_entries;
    if (ne && (!in->kind || !in->row || !in->sc || !in->ss || !in->se || !in->dc || !in->ds || !in->de || !in->flag ||
               !in->copies || !in->gt || !in->contigs || !in->contig_off || !in->contig_rank || !in->genotypes ||
               !in->genotype_off))
        return SVX_E_INVALID;
    if (ne && in->read_names && (!in->r_off || (!in->r_flat && in->r_off[in->n_rows]) || !in->names || !in->name_off))
        return SVX_E_INVALID;
    if (ne && in->sequence_alleles && (!in->b_off || !in->b_len || !in->q_len || (in->bases_bytes && !in->bases) ||
                                       (in->seqs_bytes && !in->seqs)))
        return SVX_E_INVALID;
    try {
        const bool dbg = getenv("SVX_VCF_DEBUG") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_begin = now();
        // ---- sort keys: ((contig, start, end)) per entry, SVIM_COMBINE.py:431-464; natural contig order :369-376
        std::vector<Entry> ent(ne);
        for (uint32_t e = 0; e < ne; ++e) {
            const uint32_t r = in->row[e];
            const uint8_t k = in->kind[e];
            if (r >= in->n_rows || k > SVX_VCF_BND_REV) return SVX_E_INVALID;
            int32_t contig;
            int64_t a, b;
            switch (k) {
                case SVX_VCF_DEL: contig = in->sc[r]; a = std::max<int64_t>(1, in->ss[r]); b = in->se[r]; break;
                case SVX_VCF_INS: case SVX_VCF_DUPINT_INS:
                    contig = in->dc[r]; a = std::max<int64_t>(1, in->ds[r]); b = in->de[r]; break;
                case SVX_VCF_BND: contig = in->sc[r]; a = in->ss[r] + 1; b = in->ss[r] + 2; break;
                case SVX_VCF_BND_REV: contig = in->dc[r]; a = in->ds[r] + 1; b = in->ds[r] + 2; break;
                default: contig = in->sc[r]; a = in->ss[r] + 1; b = in->se[r]; break;  // INV, DUP_TAN (both), DUP_INT as dup
            }
            if (contig < 0 || (uint32_t)contig >= in->n_contigs) return SVX_E_INVALID;
            if (in->gt[r] >= in->n_genotypes) return SVX_E_INVALID;
            // every slice the formatter will read stays inside its pool
            if (in->sequence_alleles) {
                if (in->b_off[e] < 0 || in->b_len[e] < 0 || (uint64_t)in->b_off[e] + (uint64_t)in->b_len[e] > in->bases_bytes) return SVX_E_INVALID;
                if (k == SVX_VCF_DUPINT_INS && in->b2_off && in->b2_len &&
                    (in->b2_off[e] < 0 || in->b2_len[e] < 0 || (uint64_t)in->b2_off[e] + (uint64_t)in->b2_len[e] > in->bases_bytes))
                    return SVX_E_INVALID;
                if (k == SVX_VCF_INS && in->q_len[r] > 0 &&
                    (!in->q_off || in->q_off[r] < 0 || (uint64_t)in->q_off[r] + (uint64_t)in->q_len[r] > in->seqs_bytes))
                    return SVX_E_INVALID;
            }
            if (in->sequence_alleles && k == SVX_VCF_DUPTAN_INS && in->copies[r] > 0 &&
                (uint64_t)in->b_len[e] * ((uint64_t)in->copies[r] + 1) > (1ull << 31))
                return SVX_E_TOO_LARGE;  // ALT = REF x (copies + 1) (:209-214): a line of more than 2 GiB is a damaged input
            if (in->read_names) {
                if (in->r_off[r] < 0 || in->r_off[r + 1] < in->r_off[r]) return SVX_E_INVALID;
                for (int64_t j = in->r_off[r]; j < in->r_off[r + 1]; ++j)
                    if (in->r_flat[j] < 0 || (uint64_t)in->r_flat[j] >= in->n_names) return SVX_E_INVALID;
            }
            ent[e] = Entry{in->contig_rank[contig], a, b, e};
        }
        const double t_keys = now();
        auto by_pos = [](const Entry& x, const Entry& y) { return x.start != y.start ? x.start < y.start : x.end < y.end; };
        uint32_t split_from = 32768;  // (SVX_VCF_SORT_SPLIT: tests run small tables through the split form)
        if (const char* v = getenv("SVX_VCF_SORT_SPLIT")) split_from = (uint32_t)std::max(0, atoi(v));
        if (ne < split_from) {
            std::stable_sort(ent.begin(), ent.end(), [&](const Entry& x, const Entry& y) {
                return x.rank != y.rank ? x.rank < y.rank : by_pos(x, y);
            });
        } else {
            // many entries: one stable counting pass by contig rank, then the contigs' stretches sorted side by side
            int32_t max_rank = 0;
            for (const Entry& e : ent) {
                if (e.rank < 0) return SVX_E_INVALID;
                max_rank = std::max(max_rank, e.rank);
            }
            if ((uint64_t)max_rank > 4u * (uint64_t)in->n_contigs + 16u) return SVX_E_INVALID;  // (ranks are ranks of the contigs)
            std::vector<uint32_t> first((size_t)max_rank + 2, 0);
            for (const Entry& e : ent) ++first[(size_t)e.rank + 1];
            for (size_t r = 1; r < first.size(); ++r) first[r] += first[r - 1];
            std::vector<Entry> byrank(ne);
            {
                std::vector<uint32_t> at(first.begin(), first.end() - 1);
                for (const Entry& e : ent) byrank[at[(size_t)e.rank]++] = e;
            }
            ent.swap(byrank);
            std::atomic<size_t> next(0);
            auto work = [&] {
                for (size_t r = next.fetch_add(1); r + 1 < first.size(); r = next.fetch_add(1))
                    std::stable_sort(ent.begin() + first[r], ent.begin() + first[r + 1], by_pos);
            };
            const unsigned n_sort = std::min<unsigned>(8u, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
            std::vector<std::thread> th;
            for (unsigned t = 1; t < n_sort; ++t) th.emplace_back(work);
            work();
            for (std::thread& t : th) t.join();
        }
        const double t_sort = now();
        // ---- IDs numbered per label in sorted order (:470-475): serial, one counter per label
        std::vector<int64_t> id_no(ne);
        {
            int64_t counter[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t q = 0; q < ne; ++q) id_no[q] = ++counter[KIND_LABEL[in->kind[ent[q].idx]]];
        }
        // ---- lines in sorted order: contiguous chunks of entries formatted by a few threads, then joined
        std::atomic<int> status(SVX_OK);
        auto format_range = [&](uint32_t q_lo, uint32_t q_hi, Out& o) -> int {
        for (uint32_t q = q_lo; q < q_hi; ++q) {
            const uint32_t e = ent[q].idx;
            const uint32_t r = in->row[e];
            const uint8_t k = in->kind[e];
            const bool seq = in->sequence_alleles != 0;
            const uint8_t fl = in->flag[r];
            const uint8_t* ref_b = seq ? in->bases + in->b_off[e] : nullptr;
            const int64_t ref_n = seq ? in->b_len[e] : 0;
            const bool is_src = !(k == SVX_VCF_INS || k == SVX_VCF_DUPINT_INS || k == SVX_VCF_BND_REV);
            const int32_t contig = is_src ? in->sc[r] : in->dc[r];
            // CHROM POS ID
            pool_str(o, in->contigs, in->contig_off, contig);
            o.ch('\t');
            o.num(ent[q].start);
            o.lit("\tsvim_asm.");
            const int lab = KIND_LABEL[k];
            o.lit(LABELS[lab]);
            o.ch('.');
            o.num(id_no[q]);
            o.ch('\t');
            // REF ALT
            const int64_t ss = in->ss[r], se = in->se[r], ds = in->ds[r], de = in->de[r];
            switch (k) {
                case SVX_VCF_DEL:  // SVCandidate.py:56-62: ref = fetch(max(0, start-1), end), alt = fetch(max(0, start-1), start)
                    if (seq) {
                        o.bytes(ref_b, (size_t)ref_n);
                        o.ch('\t');
                        const int64_t first = ss > 0 ? ss - 1 : 0;
                        o.bytes(ref_b, (size_t)std::min<int64_t>(ref_n, ss - first));
                    } else {
                        o.lit("N\t<DEL>");
                    }
                    break;
                case SVX_VCF_INV:  // :102-109
                    if (seq) {
                        o.bytes(ref_b, (size_t)ref_n);
                        o.ch('\t');
                        for (int64_t i = ref_n - 1; i >= 0; --i) o.ch((char)complement_upper(ref_b[i]));
                    } else {
                        o.lit("N\t<INV>");
                    }
                    break;
                case SVX_VCF_INS:  // :154-160: ref = fetch(max(0, start-1), start), alt = ref + sequence
                    if (seq) {
                        o.bytes(ref_b, (size_t)ref_n);
                        o.ch('\t');
                        o.bytes(ref_b, (size_t)ref_n);
                        if (in->q_len[r] > 0) o.bytes(in->seqs + in->q_off[r], (size_t)in->q_len[r]);
                    } else {
                        o.lit("N\t<INS>");
                    }
                    break;
                case SVX_VCF_DUPTAN_INS:  // :209-214: alt = ref * (copies + 1)
                    if (seq) {
                        o.bytes(ref_b, (size_t)ref_n);
                        o.ch('\t');
                        for (int64_t c = 0; c < in->copies[r] + 1; ++c) o.bytes(ref_b, (size_t)ref_n);
                    } else {
                        o.lit("N\t<INS>");
                    }
                    break;
                case SVX_VCF_DUPTAN_DUP: o.lit("N\t<DUP:TANDEM>"); break;
                case SVX_VCF_DUPINT_INS:  // :299-305: alt = ref + fetch(source interval)
                    if (seq) {
                        o.bytes(ref_b, (size_t)ref_n);
                        o.ch('\t');
                        o.bytes(ref_b, (size_t)ref_n);
                        if (in->b2_off && in->b2_len && in->b2_len[e] > 0) o.bytes(in->bases + in->b2_off[e], (size_t)in->b2_len[e]);
                    } else {
                        o.lit("N\t<INS>");
                    }
                    break;
                case SVX_VCF_DUPINT_DUP: o.lit("N\t<DUP:INT>"); break;
                default: {  // breakends :389-443
                    const bool s_rev = (fl & 2) != 0, d_rev = (fl & 4) != 0;
                    const bool fwd_entry = k == SVX_VCF_BND;
                    const int32_t mate_contig = fwd_entry ? in->dc[r] : in->sc[r];
                    const int64_t mate_pos = (fwd_entry ? ds : ss) + 1;
                    if (mate_contig < 0 || (uint32_t)mate_contig >= in->n_contigs) return SVX_E_INVALID;
                    // bracket form: 0 "N[c:p[", 1 "N]c:p]", 2 "]c:p]N", 3 "[c:p[N"
                    int form;
                    if (!s_rev && d_rev) form = 1;
                    else if (s_rev && !d_rev) form = 3;
                    else if (!s_rev) form = fwd_entry ? 0 : 2;   // fwd/fwd
                    else form = fwd_entry ? 2 : 0;               // rev/rev
                    o.lit("N\t");
                    const char open = (form == 0 || form == 3) ? '[' : ']';
                    if (form <= 1) o.ch('N');
                    o.ch(open);
                    pool_str(o, in->contigs, in->contig_off, mate_contig);
                    o.ch(':');
                    o.num(mate_pos);
                    o.ch(open);
                    if (form >= 2) o.ch('N');
                    break;
                }
            }
            // QUAL FILTER
            o.lit("\t.\t");
            if (k == SVX_VCF_INV && !(fl & 1)) o.lit("incomplete_inversion");
            else if ((k == SVX_VCF_DUPTAN_INS || k == SVX_VCF_DUPTAN_DUP) && !(fl & 1)) o.lit("not_fully_covered");
            else o.lit("PASS");
            // INFO
            o.lit("\tSVTYPE=");
            switch (k) {
                case SVX_VCF_DEL: o.lit("DEL;END="); o.num(se); o.lit(";SVLEN="); o.num(ss - se); break;
                case SVX_VCF_INV: o.lit("INV;END="); o.num(se); break;
                case SVX_VCF_INS: o.lit("INS;END="); o.num(ds); o.lit(";SVLEN="); o.num(de - ds); break;
                case SVX_VCF_DUPTAN_INS: o.lit("INS;END="); o.num(se); o.lit(";SVLEN="); o.num((se - ss) * in->copies[r]); break;
                case SVX_VCF_DUPTAN_DUP: o.lit("DUP:TANDEM;END="); o.num(se); o.lit(";SVLEN="); o.num(se - ss); break;
                case SVX_VCF_DUPINT_INS:
                    o.lit("INS;"); if (fl & 1) o.lit("CUTPASTE;");
                    o.lit("END="); o.num(ds); o.lit(";SVLEN="); o.num(de - ds); break;
                case SVX_VCF_DUPINT_DUP:
                    o.lit("DUP:INT;"); if (fl & 1) o.lit("CUTPASTE;");
                    o.lit("END="); o.num(se); o.lit(";SVLEN="); o.num(se - ss); break;
                default: o.lit("BND"); break;
            }
            if (in->read_names) {
                o.lit(";READS=");
                for (int64_t j = in->r_off[r]; j < in->r_off[r + 1]; ++j) {
                    if (j > in->r_off[r]) o.ch(',');
                    pool_str(o, in->names, in->name_off, in->r_flat[j]);
                }
            }
            // FORMAT sample
            if (k == SVX_VCF_DUPTAN_DUP) {
                o.lit("\tGT:CN\t");
                pool_str(o, in->genotypes, in->genotype_off, in->gt[r]);
                o.ch(':');
                o.num(in->copies[r] + 1);
            } else {
                o.lit("\tGT\t");
                pool_str(o, in->genotypes, in->genotype_off, in->gt[r]);
            }
            o.ch('\n');
        }
        return SVX_OK;
        };
        // (8 threads; 16 for the ~10^5 records of a crowded sample.  SVX_VCF_THREADS overrides — measurements)
        unsigned n_thr = std::min<unsigned>(ne >= 65536 ? 16u : 8u, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
        if (const char* v = getenv("SVX_VCF_THREADS")) n_thr = (unsigned)std::max(1, std::min(64, atoi(v)));
        if (ne < 4096) n_thr = 1;
        std::vector<Out> parts(n_thr);
        auto run = [&](unsigned t) {
            const uint32_t q_lo = (uint32_t)((uint64_t)ne * t / n_thr), q_hi = (uint32_t)((uint64_t)ne * (t + 1) / n_thr);
            size_t guess = (size_t)(q_hi - q_lo) * 96;
            if (in->sequence_alleles)
                for (uint32_t q = q_lo; q < q_hi; ++q) guess += (size_t)in->b_len[ent[q].idx] * 2;
            try {
                parts[t].s.reserve(guess);
                const int rc = format_range(q_lo, q_hi, parts[t]);
                if (rc != SVX_OK) status.store(rc);
            } catch (const std::bad_alloc&) {
                status.store(SVX_E_NOMEM);
            } catch (...) {
                status.store(SVX_E_INVALID);
            }
        };
        if (n_thr == 1) {
            run(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(run, t);
            for (std::thread& t : th) t.join();
        }
        if (status.load() != SVX_OK) return status.load();
        const double t_fmt = now();
        if (dbg) fprintf(stderr, "svx_vcf: %u entries, %u threads: keys %.1f ms, sort %.1f ms, format %.1f ms\n", ne, n_thr,
                         (t_keys - t_begin) * 1e3, (t_sort - t_keys) * 1e3, (t_fmt - t_sort) * 1e3);
        size_t total = 0;
        for (const Out& part : parts) total += part.s.size();
        if (fd >= 0) {
            std::vector<size_t> at(n_thr);
            size_t pos = 0;
            for (unsigned t = 0; t < n_thr; ++t) { at[t] = pos; pos += parts[t].s.size(); }
            auto put = [&](unsigned t) {
                if (parts[t].s.size() && pwrite_all(fd, parts[t].s.data(), parts[t].s.size(), file_at + at[t]) != SVX_OK)
                    status.store(SVX_E_INVALID);
            };
            if (n_thr == 1) {
                put(0);
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(put, t);
                for (std::thread& t : th) t.join();
            }
            if (status.load() != SVX_OK) return status.load();
            *n_bytes = total;
            if (n_lines) *n_lines = ne;
            return SVX_OK;
        }
        char* buf = (char*)malloc(total ? total : 1);
        if (!buf) return SVX_E_NOMEM;
        {
            std::vector<size_t> at(n_thr);
            size_t pos = 0;
            for (unsigned t = 0; t < n_thr; ++t) { at[t] = pos; pos += parts[t].s.size(); }
            auto copy = [&](unsigned t) { if (parts[t].s.size()) memcpy(buf + at[t], parts[t].s.data(), parts[t].s.size()); };
            if (n_thr == 1) {
                copy(0);
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(copy, t);
                for (std::thread& t : th) t.join();
            }
        }
        *text = buf;
        *n_bytes = total;
        if (n_lines) *n_lines = ne;
        return SVX_OK;
    } catch (const std::bad_alloc&) {
        return SVX_E_NOMEM;
    } catch (...) {
        return SVX_E_INVALID;
    }
}

}  // namespace

extern "C" int svx_vcf_format(const svx_vcf_in* in, char** text, uint64_t* n_bytes, uint64_t* n_lines) {
    if (!text) return SVX_E_INVALID;
    return vcf_format_impl(in, text, -1, 0, n_bytes, n_lines);
}

extern "C" int svx_vcf_write(const svx_vcf_in* in, int fd, uint64_t* n_bytes, uint64_t* n_lines) {
    if (fd < 0) return SVX_E_INVALID;
    const int fl = fcntl(fd, F_GETFL);
    if (fl < 0 || (fl & O_APPEND)) return SVX_E_INVALID;  // (pwrite on an append-mode descriptor ignores its offset)
    const off_t here = lseek(fd, 0, SEEK_CUR);  // behind what the caller has written (the header lines)
    if (here < 0) return SVX_E_INVALID;
    uint64_t n = 0;
    const int rc = vcf_format_impl(in, nullptr, fd, (uint64_t)here, &n, n_lines);
    if (n_bytes) *n_bytes = n;
    if (rc != SVX_OK) return rc;
    return lseek(fd, here + (off_t)n, SEEK_SET) < 0 ? SVX_E_INVALID : SVX_OK;
}

extern "C" void svx_vcf_free(char* text) { free(text); }
