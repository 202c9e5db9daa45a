// text_sanitize.cpp — the native text side (svim_asm_amd/csrc/svx_text.cpp, the C-ABI of include/svx_text.h) under
// AddressSanitizer / UBSan / ThreadSanitizer on the CPU: svx_fasta_fetch_batch on a wrapped FASTA with valid, clipped and
// invalid intervals, and svx_vcf_format / svx_vcf_write on random candidate columns — well-formed ones (every kind, every option) and
// ones whose indices, offsets or lengths were damaged.  A damaged input may be refused or formatted as what it now says;
// the library must not touch memory it does not own.  Test infrastructure (tests/test_text_sanitizers.py builds and runs
// it); not part of the product.
//   text_sanitize <scratch-dir> <rounds>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "svx.h"
#include "svx_text.h"

static uint64_t g_sum = 0;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    const int rounds = atoi(argv[2]);
    std::mt19937_64 rng(12345);
    auto rnd = [&](uint64_t n) { return n ? rng() % n : 0; };
    // ---- a FASTA with three sequences, 60 bases per line
    const char* names[3] = {"chr1", "chr2", "short"};
    const int64_t lens[3] = {30011, 9000, 17};
    std::vector<std::string> seqs(3);
    std::vector<int64_t> length, offset;
    std::vector<int32_t> lb, lw;
    const std::string path = dir + "/ref.fa";
    FILE* fh = fopen(path.c_str(), "wb");
    if (!fh) return 2;
    for (int k = 0; k < 3; ++k) {
        fprintf(fh, ">%s\n", names[k]);
        offset.push_back(ftell(fh));
        length.push_back(lens[k]);
        lb.push_back(60);
        lw.push_back(61);
        for (int64_t i = 0; i < lens[k]; ++i) {
            const char c = "ACGTacgtNn"[rnd(10)];
            seqs[k].push_back(c);
            fputc(c, fh);
            if ((i + 1) % 60 == 0 || i + 1 == lens[k]) fputc('\n', fh);
        }
    }
    fclose(fh);
    char err[256];
    svx_fasta* fa = nullptr;
    if (svx_fasta_open(path.c_str(), 3, length.data(), offset.data(), lb.data(), lw.data(), &fa, err, sizeof err) != 0) return 3;
    int fetched = 0, refused = 0, formatted = 0, rejected = 0;
    for (int round = 0; round < rounds; ++round) {
        // ---- fetch batches: n intervals, some clipped, occasionally an invalid one
        const uint32_t n = (uint32_t)(1 + rnd(round % 7 == 0 ? 3000 : 40));
        std::vector<int32_t> ref(n);
        std::vector<int64_t> start(n), end(n);
        std::vector<uint64_t> off(n + 1, 0);
        const bool poison = rnd(4) == 0;
        bool valid = true;
        for (uint32_t i = 0; i < n; ++i) {
            ref[i] = (int32_t)rnd(3);
            start[i] = (int64_t)rnd((uint64_t)lens[ref[i]] + 30);
            end[i] = start[i] + (int64_t)rnd(rnd(3) ? 200 : 40000);
            if (poison && i == n / 2) {
                switch (rnd(4)) {
                    case 0: ref[i] = 7; break;
                    case 1: start[i] = -5; break;
                    case 2: end[i] = start[i] - 1; break;
                    default: ref[i] = -1; break;
                }
                valid = false;
            }
            int64_t len = 0;
            if (ref[i] >= 0 && ref[i] < 3 && start[i] >= 0 && end[i] >= start[i]) {
                const int64_t e = end[i] < lens[ref[i]] ? end[i] : lens[ref[i]];
                len = e > start[i] ? e - start[i] : 0;
            }
            off[i + 1] = off[i] + (uint64_t)len;
        }
        if (poison && valid == true) {}  // (never)
        std::vector<uint8_t> out(off[n] + 1);
        const int upper = (int)rnd(2);
        const int rc = svx_fasta_fetch_batch(fa, ref.data(), start.data(), end.data(), n, upper, off.data(), out.data(), (int)rnd(5));
        if (valid) {
            if (rc != 0) return 4;
            for (uint32_t i = 0; i < n; ++i)
                for (uint64_t j = off[i]; j < off[i + 1]; ++j) {
                    char c = seqs[ref[i]][(size_t)(start[i] + (int64_t)(j - off[i]))];
                    if (upper && c >= 'a' && c <= 'z') c = (char)(c - 32);
                    if ((char)out[j] != c) return 5;
                }
            ++fetched;
        } else {
            if (rc == 0) return 6;
            ++refused;
        }
        // a wrong output layout must be refused as well
        if (valid && n > 1 && off[n] > 0) {
            std::vector<uint64_t> bad = off;
            bad[n] += 1;
            if (svx_fasta_fetch_batch(fa, ref.data(), start.data(), end.data(), n, upper, bad.data(), out.data(), 1) == 0) return 7;
        }

        // ---- VCF body: random candidate columns and entries
        const uint32_t n_rows = (uint32_t)(1 + rnd(round % 5 == 0 ? 9000 : 60));
        const uint32_t n_contigs = 3, n_gt = 3, n_names = 5;
        std::vector<int32_t> sc(n_rows), dc(n_rows);
        std::vector<int64_t> ss(n_rows), se(n_rows), ds(n_rows), de(n_rows), copies(n_rows), q_off(n_rows), q_len(n_rows), r_off(n_rows + 1, 0), r_flat;
        std::vector<uint8_t> flag(n_rows), gt(n_rows);
        std::vector<uint8_t> pool_seqs(4000);
        for (auto& c : pool_seqs) c = (uint8_t)"ACGTN"[rnd(5)];
        for (uint32_t r = 0; r < n_rows; ++r) {
            sc[r] = (int32_t)rnd(n_contigs); dc[r] = (int32_t)rnd(n_contigs);
            ss[r] = (int64_t)rnd(20000); se[r] = ss[r] + (int64_t)rnd(3000);
            ds[r] = (int64_t)rnd(20000); de[r] = ds[r] + (int64_t)rnd(3000);
            copies[r] = (int64_t)rnd(6); flag[r] = (uint8_t)rnd(8); gt[r] = (uint8_t)rnd(n_gt);
            q_len[r] = (int64_t)rnd(500); q_off[r] = (int64_t)rnd(pool_seqs.size() - (size_t)q_len[r]);
            const uint32_t nr = (uint32_t)rnd(4);
            for (uint32_t j = 0; j < nr; ++j) r_flat.push_back((int64_t)rnd(n_names));
            r_off[r + 1] = (int64_t)r_flat.size();
        }
        const uint32_t ne = (uint32_t)rnd(2 * n_rows + 1);
        std::vector<uint8_t> kind(ne);
        std::vector<uint32_t> row(ne);
        std::vector<int64_t> b_off(ne), b_len(ne), b2_off(ne), b2_len(ne);
        std::vector<uint8_t> bases(20000);
        for (auto& c : bases) c = (uint8_t)"ACGTN"[rnd(5)];
        for (uint32_t e = 0; e < ne; ++e) {
            kind[e] = (uint8_t)rnd(9); row[e] = (uint32_t)rnd(n_rows);
            b_len[e] = (int64_t)rnd(400); b_off[e] = (int64_t)rnd(bases.size() - (size_t)b_len[e]);
            b2_len[e] = (int64_t)rnd(400); b2_off[e] = (int64_t)rnd(bases.size() - (size_t)b2_len[e]);
        }
        const std::string names_pool = "r0read1rd2x3longername4";
        const int64_t name_off[6] = {0, 2, 7, 10, 12, 23};
        const std::string contig_pool = "chr1chr2short";
        const int64_t contig_off[4] = {0, 4, 8, 13};
        const int32_t contig_rank[3] = {0, 1, 2};
        const std::string gt_pool = "1/11/00/1";
        const int64_t gt_off[4] = {0, 3, 6, 9};
        svx_vcf_in in;
        memset(&in, 0, sizeof in);
        in.n_rows = n_rows; in.sc = sc.data(); in.ss = ss.data(); in.se = se.data(); in.dc = dc.data(); in.ds = ds.data(); in.de = de.data();
        in.flag = flag.data(); in.copies = copies.data(); in.gt = gt.data(); in.q_off = q_off.data(); in.q_len = q_len.data();
        in.r_off = r_off.data(); in.r_flat = r_flat.empty() ? nullptr : r_flat.data();
        in.seqs = pool_seqs.data(); in.seqs_bytes = pool_seqs.size(); in.names = names_pool.data(); in.name_off = name_off; in.n_names = n_names;
        in.contigs = contig_pool.data(); in.contig_off = contig_off; in.contig_rank = contig_rank; in.n_contigs = n_contigs;
        in.genotypes = gt_pool.data(); in.genotype_off = gt_off; in.n_genotypes = n_gt;
        in.n_entries = ne; in.kind = kind.data(); in.row = row.data();
        in.bases = bases.data(); in.bases_bytes = bases.size(); in.b_off = b_off.data(); in.b_len = b_len.data();
        in.b2_off = b2_off.data(); in.b2_len = b2_len.data();
        in.sequence_alleles = (int)rnd(2); in.read_names = (int)rnd(2);
        // damage one thing every other round
        const int damage = round % 2 ? (int)(1 + rnd(9)) : 0;
        if (ne) {
            const uint32_t e = (uint32_t)rnd(ne), r = row[e];
            switch (damage) {
                case 1: row[e] = n_rows + (uint32_t)rnd(1000); break;
                case 2: kind[e] = 9 + (uint8_t)rnd(200); break;
                case 3: sc[r] = dc[r] = 3 + (int32_t)rnd(100); break;
                case 4: sc[r] = dc[r] = -2; break;
                case 5: gt[r] = 200; break;
                case 6: b_off[e] = (int64_t)bases.size() - 3; b_len[e] = 40; break;
                case 7: q_off[r] = (int64_t)pool_seqs.size() - 1; q_len[r] = 4000000000ll; break;
                case 8: if (!r_flat.empty()) r_flat[rnd(r_flat.size())] = 77; break;
                case 9: copies[r] = (int64_t)1 << 40; break;
                default: break;
            }
        }
        char* text = nullptr;
        uint64_t n_bytes = 0, n_lines = 0;
        const int vrc = svx_vcf_format(&in, &text, &n_bytes, &n_lines);
        if (vrc == 0) {
            if (n_lines != ne) return 8;
            uint64_t nl = 0;
            for (uint64_t i = 0; i < n_bytes; ++i) { g_sum += (uint8_t)text[i]; nl += text[i] == '\n'; }
            if (nl != ne) return 9;
            // the same lines through svx_vcf_write (every formatting thread writes its stretch with pwrite behind what the
            // caller wrote) and, every other round, with the entries sorted contig by contig (SVX_VCF_SORT_SPLIT=0)
            {
                if (round & 1) setenv("SVX_VCF_SORT_SPLIT", "0", 1); else unsetenv("SVX_VCF_SORT_SPLIT");
                const std::string vpath = dir + "/out.vcf";
                FILE* vf = fopen(vpath.c_str(), "wb");
                if (!vf) return 12;
                fputs("#header\n", vf);
                fflush(vf);
                uint64_t wb = 0, wl = 0;
                const int wrc = svx_vcf_write(&in, fileno(vf), &wb, &wl);
                fclose(vf);
                unsetenv("SVX_VCF_SORT_SPLIT");
                if (wrc != 0 || wb != n_bytes || wl != n_lines) return 13;
                vf = fopen(vpath.c_str(), "rb");
                if (!vf) return 12;
                std::vector<char> back(n_bytes + 8);
                const size_t got = fread(back.data(), 1, back.size(), vf);
                const bool at_end = fgetc(vf) == EOF;
                fclose(vf);
                if (got != n_bytes + 8 || !at_end || memcmp(back.data(), "#header\n", 8) != 0 ||
                    (n_bytes && memcmp(back.data() + 8, text, n_bytes) != 0))
                    return 14;
            }
            svx_vcf_free(text);
            ++formatted;
        } else {
            if (!damage) return 10;  // a well-formed input must format
            if (text) return 11;
            ++rejected;
        }
    }
    svx_fasta_close(fa);
    printf("text_sanitize ok: %d fetch batches read, %d refused; %d VCF bodies formatted, %d rejected (checksum %llu)\n", fetched,
           refused, formatted, rejected, (unsigned long long)g_sum);
    return 0;
}
