// svx_postpass_dev.h — device routine of the per-read post-passes of analyze_read_segments (SVIM_inter.py:260-338),
// shared by k_segments_post (svx_postpass.hip) and the fused split-segment chain of svx_collect_batch_dev
// (svx_cigar.hip).  See svx_postpass.hip for the procedure.
#pragma once
#include "svx_internal.h"
#include "svx_linkage_dev.h"

namespace svx_post_dev {

struct PostArgs {
    const svx_raw* raw;
    const uint32_t* read_off;
    uint32_t n_reads;
    const int32_t* contig_rank;
    uint32_t n_contigs;
    int32_t min_sv, max_sv;
    svx_post* out;
    const uint64_t* out_off;
    uint32_t* out_cnt;
    char* scratch;
    const uint64_t* scratch_off;
    uint64_t scratch_stride;  // != 0: every read's slice has this size (slice r at r * stride; no offset table)
};

// scratch of a read with s slots: 5 int arrays (rank, ref, start, end, side) + labels + the condensed
// distance vector + the linkage state
__host__ __device__ constexpr size_t post_scratch_bytes(uint32_t s) {
    return ((size_t)6 * s * 4 + 7) / 8 * 8 + (size_t)8 * s * (s ? s - 1 : 0) / 2 + (svx_link_bytes(s) + 7) / 8 * 8 + 16;
}

__device__ __forceinline__ svx_post make_post(int32_t kind, int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4,
                                              int32_t a5) {
    svx_post r;
    r.kind = kind; r.a0 = a0; r.a1 = a1; r.a2 = a2; r.a3 = a3; r.a4 = a4; r.a5 = a5; r.pad = 0;
    return r;
}

__device__ __forceinline__ int64_t abs64(int64_t v) { return v < 0 ? -v : v; }

// SVIM_inter.py:19-39 on the float64 rows [start, end, 0 (left) / 1 (right)]
__device__ __forceinline__ double reciprocal_overlap_distance(int32_t s1, int32_t e1, int32_t d1, int32_t s2, int32_t e2,
                                                              int32_t d2) {
    if (d1 == d2 || s2 >= e1 || s1 >= e2) return 1.0;
    const double overlap = (double)((e1 < e2 ? e1 : e2) - (s2 >= s1 ? s2 : s1));
    const double r1 = overlap / (double)(e1 - s1), r2 = overlap / (double)(e2 - s2);
    return 1.0 - (r1 < r2 ? r1 : r2);
}

// The three post-passes of read r, by one lane.
__device__ __forceinline__ void post_one_read(const PostArgs& p, const uint32_t r) {
    const uint32_t b = p.read_off[r], e = p.read_off[r + 1];
    svx_post* out = p.out + p.out_off[r];
    uint32_t n_out = 0;

    // ---- 1. tandem duplications
    {
        bool have = false, fully = false;
        int32_t chrom = 0, cnt = 0, first_dir = 0;
        int64_t S = 0, E = 0;
        for (uint32_t i = b; i < e; ++i) {
            const svx_raw t = p.raw[i];
            if (t.kind != SVX_RAW_TANDEM) continue;
            if (!have) {
                have = true;
                chrom = t.a0; S = t.a1; E = t.a2; cnt = 1; fully = t.a3 != 0; first_dir = t.a4;
            } else if (chrom == t.a0 && abs64(S - (int64_t)t.a1 * cnt) < 20ll * cnt &&
                       abs64(E - (int64_t)t.a2 * cnt) < 20ll * cnt && first_dir == t.a4) {
                S += t.a1; E += t.a2; ++cnt; fully = fully || t.a3 != 0;
            } else {
                out[n_out++] = make_post(SVX_POST_TANDEM, chrom, (int32_t)(S / cnt), (int32_t)(E / cnt), cnt, fully ? 1 : 0, 0);
                chrom = t.a0; S = t.a1; E = t.a2; cnt = 1; fully = t.a3 != 0;
            }
        }
        if (have) out[n_out++] = make_post(SVX_POST_TANDEM, chrom, (int32_t)(S / cnt), (int32_t)(E / cnt), cnt, fully ? 1 : 0, 0);
    }

    // ---- 2. interspersed duplications from pairs of breakends
    // BND record: a0 chr1, a1 pos1, a2 dir1, a3 chr2, a4 pos2, a5 dir2 (dir 0 'fwd', 1 'rev')
    for (uint32_t ti = b; ti < e; ++ti) {
        const svx_raw t = p.raw[ti];
        if (t.kind != SVX_RAW_BND) continue;
        for (uint32_t bi = b; bi < ti; ++bi) {
            const svx_raw q = p.raw[bi];
            if (q.kind != SVX_RAW_BND) continue;
            const int32_t near = q.a1 > t.a4 ? q.a1 - t.a4 : t.a4 - q.a1;
            if (!(q.a2 == t.a5 && q.a5 == t.a2 && q.a0 == t.a3 && near < 20 && q.a3 == t.a0 && q.a5 == q.a2)) continue;
            if (q.a2 == 0) {
                const int64_t length = (int64_t)t.a1 + 1 - q.a4;
                if (p.min_sv <= length && length <= p.max_sv) {
                    const int64_t mid = ((int64_t)q.a1 + 1 + t.a4) / 2;
                    out[n_out++] = make_post(SVX_POST_DUP_INT, q.a3, q.a4, t.a1 + 1, q.a0, (int32_t)mid, (int32_t)(mid + length));
                }
            } else {
                const int64_t length = (int64_t)q.a4 + 1 - t.a1;
                if (p.min_sv <= length && length <= p.max_sv) {
                    const int64_t mid = ((int64_t)q.a1 + t.a4 + 1) / 2;
                    out[n_out++] = make_post(SVX_POST_DUP_INT, q.a3, t.a1, q.a4 + 1, q.a0, (int32_t)mid, (int32_t)(mid + length));
                }
            }
        }
    }

    // ---- 3. inversions
    {
        const uint32_t s = e - b;
        char* mem = p.scratch + (p.scratch_stride ? (uint64_t)r * p.scratch_stride : p.scratch_off[r]);
        int32_t* rk = reinterpret_cast<int32_t*>(mem);
        int32_t* rf = rk + s;
        int32_t* st = rf + s;
        int32_t* en = st + s;
        int32_t* sd = en + s;
        uint32_t* lab = reinterpret_cast<uint32_t*>(sd + s);
        double* cond = reinterpret_cast<double*>(mem + ((size_t)6 * s * 4 + 7) / 8 * 8);
        char* link = reinterpret_cast<char*>(cond + (size_t)s * (s ? s - 1 : 0) / 2);
        uint32_t n = 0;
        for (uint32_t i = b; i < e; ++i) {  // stable insertion sort by (name rank, start, end)
            const svx_raw t = p.raw[i];
            if (t.kind != SVX_RAW_INV) continue;
            const int32_t rank = (uint32_t)t.a0 < p.n_contigs ? p.contig_rank[t.a0] : t.a0;
            uint32_t j = n++;
            while (j > 0 && (rk[j - 1] > rank || (rk[j - 1] == rank && (st[j - 1] > t.a1 || (st[j - 1] == t.a1 && en[j - 1] > t.a2))))) {
                rk[j] = rk[j - 1]; rf[j] = rf[j - 1]; st[j] = st[j - 1]; en[j] = en[j - 1]; sd[j] = sd[j - 1];
                --j;
            }
            rk[j] = rank; rf[j] = t.a0; st[j] = t.a1; en[j] = t.a2; sd[j] = t.a3 >= 2 ? 1 : 0;  // left_* 0, right_* 1
        }
        uint32_t g0 = 0, g1 = 0;  // active group [g0, g1) of the sorted list
        int32_t max_end = 0;
        auto flush = [&]() {
            const uint32_t m = g1 - g0;
            if (m == 0) return;
            if (m == 1) {
                out[n_out++] = make_post(SVX_POST_INV, rf[g0], st[g0], en[g0], 0, 0, 0);
                return;
            }
            size_t c = 0;
            for (uint32_t i = g0; i + 1 < g1; ++i)
                for (uint32_t j = i + 1; j < g1; ++j)
                    cond[c++] = reciprocal_overlap_distance(st[i], en[i], sd[i], st[j], en[j], sd[j]);
            svx_linkage_cut_one(m, cond, 0.3, lab, link);
            uint32_t n_clusters = 0;
            for (uint32_t i = 0; i < m; ++i) n_clusters = lab[i] > n_clusters ? lab[i] : n_clusters;
            for (uint32_t l = 1; l <= n_clusters; ++l) {
                bool first = true;
                int32_t chrom = 0, hi_start = 0, lo_end = 0;
                uint32_t members = 0;
                for (uint32_t i = 0; i < m; ++i) {
                    if (lab[i] != l) continue;
                    const uint32_t k = g0 + i;
                    if (first) { chrom = rf[k]; hi_start = st[k]; lo_end = en[k]; first = false; }
                    else { hi_start = st[k] > hi_start ? st[k] : hi_start; lo_end = en[k] < lo_end ? en[k] : lo_end; }
                    ++members;
                }
                out[n_out++] = make_post(SVX_POST_INV, chrom, hi_start, lo_end, members > 1 ? 1 : 0, 0, 0);
            }
        };
        for (uint32_t k = 0; k < n; ++k) {
            if (g1 == g0) {
                g0 = k; g1 = k + 1; max_end = en[k];
            } else if (rf[k] == rf[g1 - 1] && st[k] < max_end) {
                g1 = k + 1;
                max_end = en[k] > max_end ? en[k] : max_end;
            } else {
                flush();
                g0 = g1 = k + 1;  // the breakpoint that closes a group is dropped (:334-336)
            }
        }
        flush();
    }
    p.out_cnt[r] = n_out;
}


}  // namespace svx_post_dev
