// bam_sanitize.cpp — the native BAM reader (svim_asm_amd/csrc/svx_bam.cpp, the C-ABI of include/svx_bam.h) under
// AddressSanitizer / UBSan on the CPU: every entry point on well-formed files, then on damaged copies of them
// (flipped bytes, truncations, overwritten BGZF / record length fields).  A damaged file may be refused with an
// error or read as whatever it now says; the reader must not touch memory it does not own.
// Test infrastructure (tests/test_bam_sanitizers.py builds and runs it); not part of the product.
//   bam_sanitize <scratch-dir> <mutations-per-file> <file.bam>...
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "svx_bam.h"
#include "svx_inflate_dev.h"

// Stand-ins for the device launches (svx_inflate.hip is not in this build): they fail like everything else that needs a
// device here, but with them registered the reader takes the paths of a build that has kernels — the record walks defer
// their check (svx_bam_set_defer_verify) and the threads have to catch up on it.
static int no_inflate(void*, const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint8_t*,
                      const uint64_t*, uint32_t*, uint32_t*, void*, uint32_t) { return 1; }
static int no_gather(void*, const uint8_t*, const uint64_t*, const uint32_t*, const uint64_t*, uint32_t, uint8_t*) { return 1; }
extern "C" void svx_bam_register_device_kernels(svx_inflate_launch_fn, svx_gather_launch_fn);

static uint64_t g_sum = 0;  // keeps the reads of every column alive

static int walk(const char* path, int threads, bool per_contig) {
    char err[256] = {0};
    svx_bam* b = nullptr;
    if (svx_bam_open(path, threads, &b, err, sizeof err) != 0 || !b) return 1;  // refused: fine
    static unsigned turn = 0;
    (void)svx_bam_set_verify(b, (int)(++turn & 1));  // whole members + CRC32 and inflate-what-is-needed in turn
    // every third handle asks for a page-locked pool, its device copy and a device share of the sequence slices: there is
    // no device here, so every HIP call fails — the walkers' early-pool claim, the lanes' bring-up threads and the
    // fall-backs to pageable memory and to the host's decoder run under the sanitizers
    if (turn % 3 == 0) {
        (void)svx_bam_set_pinned_device(b, 0);
        (void)svx_bam_set_device_inflate(b, 50);
        (void)svx_bam_set_device_inflate_min(b, 0);
        static const bool registered = (svx_bam_register_device_kernels(&no_inflate, &no_gather), true);
        (void)registered;
        (void)svx_bam_set_defer_verify(b, (int)((turn / 3) & 1));  // every other one of them leaves the walks' check pending
    }
    const char* text = nullptr;
    uint64_t l_text = 0;
    int32_t n_ref = 0;
    if (svx_bam_header(b, &text, &l_text, &n_ref) == 0)
        for (uint64_t i = 0; i < l_text; ++i) g_sum += (uint8_t)text[i];
    for (int32_t t = 0; t < n_ref; ++t) {
        const char* name = nullptr;
        int32_t len = 0;
        if (svx_bam_reference(b, t, &name, &len) == 0 && name) g_sum += strlen(name) + (uint32_t)len;
    }
    std::vector<uint64_t> span((size_t)(n_ref > 0 ? n_ref : 1));
    if (svx_bam_index_state(b) == 1 && n_ref > 0) (void)svx_bam_contig_spans(b, span.data());
    for (int pass = 0; pass < (per_contig ? 2 : 1); ++pass) {
        int rc;
        if (pass == 0) {
            rc = svx_bam_load(b, nullptr, 0);
        } else {
            std::vector<int32_t> tids;
            for (int32_t t = 0; t < n_ref; t += 2) tids.push_back(t);
            rc = svx_bam_load(b, tids.data(), (int32_t)tids.size());
        }
        if (rc != 0) { g_sum += strlen(svx_bam_last_error(b)); continue; }
        svx_bam_columns c;
        if (svx_bam_get_columns(b, &c) != 0) continue;
        for (uint64_t r = 0; r < c.n_records; ++r) {
            g_sum += (uint32_t)c.tid[r] + (uint32_t)c.pos[r] + (uint32_t)c.l_seq[r] + (uint32_t)c.ref_len[r] + c.flag[r] + c.mapq[r] + c.voffset[r];
            for (uint64_t k = c.cigar_off[r]; k < c.cigar_off[r + 1]; ++k) g_sum += c.cigar[k];
            for (uint64_t k = c.name_off[r]; k < c.name_off[r + 1]; ++k) g_sum += (uint8_t)c.names[k];
            for (uint64_t k = c.aux_off[r]; k < c.aux_off[r + 1]; ++k) g_sum += c.aux[k];
            if (c.sa_off[r] >= 0)
                for (uint32_t k = 0; k < c.sa_len[r]; ++k) g_sum += c.aux[(uint64_t)c.sa_off[r] + k];
        }
        // bases: the whole read of the first records, then ragged and out-of-range slices
        const uint32_t n = (uint32_t)(c.n_records < 64 ? c.n_records : 64);
        std::vector<uint32_t> rec, begin, end;
        std::vector<uint64_t> off(1, 0);
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t l = c.l_seq[r] > 0 ? (uint32_t)c.l_seq[r] : 0u;
            const uint32_t cases[3][2] = {{0, l}, {l / 3, l / 2 + 1}, {l, l + 100}};
            for (auto& cs : cases) {
                rec.push_back(r); begin.push_back(cs[0]); end.push_back(cs[1]);
                const uint32_t b0 = cs[0] < l ? cs[0] : l, e0 = cs[1] < l ? cs[1] : l;
                off.push_back(off.back() + (e0 > b0 ? e0 - b0 : 0));
            }
        }
        std::vector<uint8_t> out(off.back() + 1);
        g_sum += svx_bam_pending_members(b);
        if (pass == 0 && (turn & 2)) g_sum += (uint64_t)svx_bam_verify_pending(b);  // (sometimes before, sometimes inside the slices' call)
        if (!rec.empty() && svx_bam_seq_slices(b, rec.data(), begin.data(), end.data(), (uint32_t)rec.size(), off.data(), out.data()) == 0)
            for (uint8_t v : out) g_sum += v;
        g_sum += (uint64_t)svx_bam_verify_pending(b) + svx_bam_pending_members(b);
    }
    svx_bam_close(b);
    return 0;
}

static std::vector<uint8_t> slurp(const std::string& p) {
    std::vector<uint8_t> v;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? (size_t)n : 0);
    if (n > 0 && fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}

static void spill(const std::string& p, const std::vector<uint8_t>& v) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) return;
    if (!v.empty()) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string scratch = argv[1];
    const int n_mut = atoi(argv[2]);
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int refused = 0, read = 0;
    for (int a = 3; a < argc; ++a) {
        const std::string path = argv[a];
        if (walk(path.c_str(), 3, true) != 0) { fprintf(stderr, "well-formed file refused: %s\n", path.c_str()); return 1; }
        if (walk(path.c_str(), 1, false) != 0) return 1;
        // (a file that comes with a .csi instead of a .bai: the same treatment for that index)
        const bool has_bai = !slurp(path + ".bai").empty();
        const std::string ext = has_bai ? ".bai" : ".csi";
        const std::vector<uint8_t> good = slurp(path), bai = slurp(path + ext);
        for (int m = 0; m < n_mut && !good.empty(); ++m) {
            std::vector<uint8_t> bad = good;
            const uint64_t kind = next() % 5;
            if (kind == 0) {                       // a few flipped bytes anywhere
                for (int k = 0; k < 1 + (int)(next() % 4); ++k) bad[next() % bad.size()] ^= (uint8_t)(1u << (next() % 8));
            } else if (kind == 1) {                // truncated
                bad.resize(next() % bad.size());
            } else if (kind == 2) {                // a BGZF header field (BSIZE / ISIZE area of the first blocks)
                const size_t at = next() % (bad.size() < 64 ? bad.size() : 64);
                bad[at] = (uint8_t)next();
            } else if (kind == 3) {                // a run of random bytes
                const size_t at = next() % bad.size(), len = 1 + next() % 32;
                for (size_t k = at; k < at + len && k < bad.size(); ++k) bad[k] = (uint8_t)next();
            } else {                               // the tail (EOF marker, last block)
                const size_t len = 1 + next() % 40;
                for (size_t k = bad.size() > len ? bad.size() - len : 0; k < bad.size(); ++k) bad[k] = (uint8_t)next();
            }
            const std::string p = scratch + "/mut.bam";
            spill(p, bad);
            std::vector<uint8_t> bad_bai = bai;   // the index next to it: intact, damaged or absent
            const uint64_t ik = next() % 3;
            if (ik == 1 && !bad_bai.empty()) for (int k = 0; k < 3; ++k) bad_bai[next() % bad_bai.size()] = (uint8_t)next();
            remove((p + (has_bai ? ".csi" : ".bai")).c_str());
            if (ik == 2 || bad_bai.empty()) remove((p + ext).c_str()); else spill(p + ext, bad_bai);
            (walk(p.c_str(), 1 + (int)(next() % 4), (next() & 1) != 0) ? refused : read)++;
        }
    }
    printf("bam_sanitize ok: %d damaged files refused, %d read, checksum %llu\n", refused, read, (unsigned long long)g_sum);
    return 0;
}
