// svx_pairhost.cpp — host arithmetic of the PAIR step that feeds the device (include/svx.h): the haplotype recipes of
// compute_distance (reference SVIM_COMBINE.py:43-100) for every cross-haplotype job of a step, from the candidate
// COLUMNS.  The reference builds two Python strings per pair out of six `reference.fetch` calls; here a haplotype is
// three pieces of one byte pool (svx_hap_piece: the partition's reference window before the variant, the middle, the
// window behind it), assembled and aligned on the device (svx_haplotype_distance_batch*).  Host code (threads), no GPU.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "svx.h"

namespace {

enum { T_DEL = 0, T_INV = 1, T_INS = 2, T_DUP_TAN = 3, T_DUP_INT = 4 };  // svim_asm_amd/table.py TYPE_ORDER

inline void put(svx_hap_piece* p, int64_t off, int64_t len, int64_t rep, uint16_t flags) {
    if (len > 0 && rep > 0) {
        p->off = (uint64_t)off;
        p->len = (uint32_t)len;
        p->repeat = (uint16_t)rep;
        p->flags = flags;
    } else {
        p->off = 0; p->len = 0; p->repeat = 0; p->flags = 0;
    }
}

}  // namespace

extern "C" int svx_pair_recipes(const svx_recipe_in* in, svx_hap_piece* pieces, int64_t* seq_range, int64_t* worst_copies) {
    if (!in || !seq_range) return SVX_E_INVALID;
    seq_range[0] = seq_range[1] = seq_range[2] = seq_range[3] = 0;
    if (worst_copies) *worst_copies = 0;
    const uint64_t J = in->n_jobs;
    if (J == 0) return SVX_OK;
    if (!pieces || !in->type || !in->ss || !in->se || !in->ds || !in->q_off || !in->q_len || !in->copies || !in->job_a ||
        !in->job_b || !in->job_part || !in->part_type || !in->part_len || !in->win_base || !in->win_lo)
        return SVX_E_INVALID;
    // ---- what is indexed with: rows and partitions inside their tables
    for (uint64_t j = 0; j < J; ++j) {
        const int64_t a = in->job_a[j], b = in->job_b[j], p = in->job_part[j];
        if (a < 0 || b < 0 || p < 0 || (uint64_t)a >= in->n_rows || (uint64_t)b >= in->n_rows || (uint64_t)p >= in->n_parts)
            return SVX_E_INVALID;
    }
    // ---- the stretches of the sequence pool the jobs' inserted alleles lie in: one per haplotype table (their
    // sequences lie one behind the other in the pool, `seq_split` is where the second one's begin)
    const bool split = in->seq_split >= 0;
    int64_t lo[2] = {INT64_MAX, INT64_MAX}, hi[2] = {INT64_MIN, INT64_MIN};
    int64_t worst = 0;
    for (uint64_t j = 0; j < J; ++j) {
        const int64_t typ = in->part_type[in->job_part[j]];
        const int64_t rows[2] = {in->job_a[j], in->job_b[j]};
        if (typ == T_INS) {
            for (int h = 0; h < 2; ++h) {
                const int64_t r = rows[h];
                if (in->q_len[r] <= 0) continue;
                const int side = (split && in->q_off[r] >= in->seq_split) ? 1 : 0;
                lo[side] = std::min(lo[side], in->q_off[r]);
                hi[side] = std::max(hi[side], in->q_off[r] + in->q_len[r]);
            }
        } else if (typ == T_DUP_TAN) {
            worst = std::max(worst, std::max(in->copies[rows[0]], in->copies[rows[1]]));
        }
    }
    if (worst_copies) *worst_copies = worst;
    if (worst + 1 > 0xFFFF) return SVX_E_TOO_LARGE;  // a piece repeats at most 65 535 times
    int64_t shift[2] = {0, 0};
    int64_t at = (int64_t)in->extra_at;
    for (int side = 0; side < 2; ++side) {
        if (lo[side] == INT64_MAX) continue;
        seq_range[2 * side] = lo[side];
        seq_range[2 * side + 1] = hi[side];
        shift[side] = lo[side] - at;
        at += hi[side] - lo[side];
    }
    // ---- the six pieces of every job
    const uint16_t up = SVX_PIECE_UPPER;
    auto fill = [&](uint64_t j0, uint64_t j1) {
        for (uint64_t j = j0; j < j1; ++j) {
            const int64_t part = in->job_part[j];
            const int64_t typ = in->part_type[part], L = in->part_len[part], base = in->win_base[part], wlo = in->win_lo[part];
            const int64_t rows[2] = {in->job_a[j], in->job_b[j]};
            int64_t s[2], e[2];
            for (int h = 0; h < 2; ++h) {
                const int64_t r = rows[h];
                const uint8_t t = in->type[r];
                const bool src_like = t == T_DEL || t == T_INV || t == T_DUP_TAN;  // (:47,:60,:71: the source interval;
                s[h] = src_like ? in->ss[r] : in->ds[r];                            //  INS, DUP_INT: the destination START
                e[h] = src_like ? in->se[r] : in->ds[r];                            //  twice, :71,:83)
            }
            const int64_t rs = std::max<int64_t>(0, std::min(s[0], s[1]) - 100);
            const int64_t re = std::min(L, std::max(e[0], e[1]) + 100);
            const bool tan = typ == T_DUP_TAN, inv = typ == T_INV;
            for (int h = 0; h < 2; ++h) {
                svx_hap_piece* p = pieces + (j * 2 + (uint64_t)h) * 3;
                const int64_t r = rows[h];
                // reference[region_start : start] and reference[end : region_end], with fetch()'s clamping of the end
                put(p + 0, base + rs - wlo, std::min(s[h], L) - rs, 1, up);
                put(p + 2, base + e[h] - wlo, re - e[h], 1, up);
                int64_t m_off = 0, m_len = 0;
                if (inv || tan) {
                    m_off = base + s[h] - wlo;
                    m_len = std::min(e[h], L) - s[h];
                } else if (typ == T_INS) {
                    const int side = (split && in->q_off[r] >= in->seq_split) ? 1 : 0;
                    m_off = in->q_off[r] - shift[side];
                    m_len = in->q_len[r];
                } else if (typ == T_DUP_INT && in->mid_off && in->mid_len) {
                    m_off = in->mid_off[j * 2 + (uint64_t)h];
                    m_len = in->mid_len[j * 2 + (uint64_t)h];
                }
                if (typ == T_DEL) m_len = 0;
                const int64_t m_rep = tan ? in->copies[r] + 1 : 1;
                const uint16_t m_flg = inv ? (uint16_t)(up | SVX_PIECE_REVCOMP) : ((tan || typ == T_DUP_INT) ? up : (uint16_t)0);
                put(p + 1, m_off, m_len, m_rep, m_flg);
            }
        }
    };
    const unsigned n_thr = J < 8192 ? 1u : std::min<unsigned>(8u, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
    if (n_thr == 1) {
        fill(0, J);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(fill, J * t / n_thr, J * (t + 1) / n_thr);
        for (std::thread& t : th) t.join();
    }
    return SVX_OK;
}
