/*
 * svx_oracle.c — CPU restatement of the SVIM-asm hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (svim_asm_amd/) never does.  Every function cites the
 * reference lines (under /root/reference/src/svim_asm) it restates.  Written from
 * the behaviour described in SURVEY.md Appendix A; no reference source is copied.
 *
 * Pinning: checked against the reference's own known-answer vectors
 * (tests/test_intra.py:8-22, tests/test_inter.py:8-11) and against golden vectors
 * produced by importing the reference in the build container
 * (tests/golden/, generator oracle/make_golden.py).  See tests/test_oracle_pins.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- a1: analyze_cigar_indel (SVIM_intra.py:8-30), batch form with the
 *      ref_start add of analyze_alignment_indel (SVIM_intra.py:36-43) ---------- */
uint64_t orc_cigar_extract(const uint32_t* cigar, const uint64_t* aln_off, uint32_t n_aln,
                           const int32_t* ref_start, uint32_t min_len, uint32_t* o_aln,
                           uint32_t* o_ref, uint32_t* o_read, uint32_t* o_len, uint8_t* o_type,
                           uint64_t cap) {
    uint64_t n = 0;
    for (uint32_t a = 0; a < n_aln; ++a) {
        uint32_t pos_ref = 0, pos_read = 0;                      /* :10-11 */
        const uint32_t rs = ref_start ? (uint32_t)ref_start[a] : 0u;
        for (uint64_t i = aln_off[a]; i < aln_off[a + 1]; ++i) { /* :13 */
            const uint32_t op = cigar[i] & 15u, len = cigar[i] >> 4;
            switch (op) {
                case 0: /* :14-16 */
                case 7: /* :27-29 */
                case 8:
                    pos_ref += len;
                    pos_read += len;
                    break;
                case 1: /* :17-20 */
                    if (len >= min_len) {
                        if (n < cap) {
                            o_aln[n] = a; o_ref[n] = rs + pos_ref; o_read[n] = pos_read;
                            o_len[n] = len; o_type[n] = 0;
                        }
                        ++n;
                    }
                    pos_read += len;
                    break;
                case 2: /* :21-24 */
                    if (len >= min_len) {
                        if (n < cap) {
                            o_aln[n] = a; o_ref[n] = rs + pos_ref; o_read[n] = pos_read;
                            o_len[n] = len; o_type[n] = 1;
                        }
                        ++n;
                    }
                    pos_ref += len;
                    break;
                case 4: /* :25-26 */
                    pos_read += len;
                    break;
                default: /* N, H, P, B and undefined codes: no branch in :14-29 */
                    break;
            }
        }
    }
    return n;
}

/* Count only (used to size outputs and as the timed CPU baseline body). */
uint64_t orc_cigar_count(const uint32_t* cigar, uint64_t n_ops, uint32_t min_len) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_ops; ++i) {
        const uint32_t op = cigar[i] & 15u, len = cigar[i] >> 4;
        n += ((op == 1u || op == 2u) && len >= min_len);
    }
    return n;
}

/* pysam / htslib per-alignment quantities:
 * reference_end - reference_start, query_alignment_start/end, infer_read_length,
 * hard-clipped bases (get_cigar_stats()[0][5], SVIM_COLLECT.py:11). */
void orc_cigar_stats(const uint32_t* cigar, const uint64_t* aln_off, uint32_t n_aln,
                     uint32_t* ref_len, uint32_t* q_start, uint32_t* q_end, uint32_t* read_len,
                     uint32_t* n_hard) {
    for (uint32_t a = 0; a < n_aln; ++a) {
        uint32_t rl = 0, qs = 0, qa = 0, il = 0, h = 0;
        int lead = 1;
        for (uint64_t i = aln_off[a]; i < aln_off[a + 1]; ++i) {
            const uint32_t op = cigar[i] & 15u, len = cigar[i] >> 4;
            if (lead) {
                if (op == 4u) qs += len;
                else if (op != 5u) lead = 0;
            }
            if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) rl += len;
            if (op == 0u || op == 1u || op == 7u || op == 8u) qa += len;
            if (op == 0u || op == 1u || op == 4u || op == 5u || op == 7u || op == 8u) il += len;
            if (op == 5u) h += len;
        }
        ref_len[a] = rl; q_start[a] = qs; q_end[a] = qs + qa; read_len[a] = il; n_hard[a] = h;
    }
}

/* ---- a3: adjacent-pair decision tree of analyze_read_segments (SVIM_inter.py:62-258) */
typedef struct { int32_t q_start, q_end, ref_id, ref_start, ref_end, is_reverse; } orc_seg;
typedef struct { int32_t min_sv, max_sv, qgt, qot, rgt, rot; } orc_params;
typedef struct { int32_t kind, a0, a1, a2, a3, a4, a5, pad; } orc_raw;
enum { R_NONE = 0, R_INS = 1, R_DEL = 2, R_BND = 3, R_TANDEM = 4, R_INV = 5 };
enum { FWD = 0, REV = 1 };

static orc_raw mk(int kind, int a0, int a1, int a2, int a3, int a4, int a5) {
    orc_raw r = {kind, a0, a1, a2, a3, a4, a5, 0};
    return r;
}

static orc_raw classify_pair(const orc_seg* c, const orc_seg* n, int32_t read_len,
                             const orc_params* o) {
    const int32_t d_read = n->q_start - c->q_end; /* :95 */
    if (c->ref_id == n->ref_id) {                 /* :98 */
        const int chr = c->ref_id;
        if (c->is_reverse == n->is_reverse) { /* :101 */
            const int32_t d_ref = c->is_reverse ? c->ref_start - n->ref_end   /* :104 */
                                                : n->ref_start - c->ref_end;  /* :106 */
            if (d_read >= -o->qot) {                                           /* :108 */
                if (d_ref >= -o->rot) {                                        /* :110 */
                    const int32_t dev = d_read - d_ref;                        /* :111 */
                    if (dev >= o->min_sv) {                                    /* :113 */
                        if (d_ref <= o->rgt) {                                 /* :115 */
                            if (!c->is_reverse)                                /* :116-118 */
                                return mk(R_INS, chr, c->ref_end, c->ref_end + dev, c->q_end, dev, 0);
                            else                                               /* :120-121 */
                                return mk(R_INS, chr, c->ref_start, c->ref_start + dev,
                                          read_len - n->q_start, dev, 0);
                        }
                    } else if (-o->max_sv <= dev && dev <= -o->min_sv) { /* :123 */
                        if (d_read <= o->qgt) {                          /* :125 */
                            if (!c->is_reverse) return mk(R_DEL, chr, c->ref_end, c->ref_end - dev, 0, 0, 0);
                            else return mk(R_DEL, chr, n->ref_end, n->ref_end - dev, 0, 0, 0);
                        }
                    } else if (dev < -o->max_sv) { /* :131 */
                        if (d_read <= o->qgt) {    /* :133 */
                            if (!c->is_reverse) return mk(R_BND, chr, c->ref_end - 1, FWD, chr, n->ref_start, FWD);
                            else return mk(R_BND, chr, c->ref_start, REV, chr, n->ref_end - 1, REV);
                        }
                    }
                } else {                        /* :141 overlap on reference */
                    if (d_read <= o->qgt) {     /* :143 */
                        const int32_t dev = d_read - d_ref;
                        if (dev >= o->min_sv) { /* :146 */
                            if (!c->is_reverse) {
                                if (n->ref_end > c->ref_start)       /* :149 */
                                    return mk(R_TANDEM, chr, n->ref_start, n->ref_start + dev, 1, 1, 0);
                                else if (d_ref >= -o->max_sv)        /* :152 */
                                    return mk(R_TANDEM, chr, n->ref_start, n->ref_start + dev, 0, 1, 0);
                                else                                 /* :155-157 */
                                    return mk(R_BND, chr, c->ref_end - 1, FWD, chr, n->ref_start, FWD);
                            } else {
                                if (n->ref_start < c->ref_end)       /* :160 */
                                    return mk(R_TANDEM, chr, c->ref_start, c->ref_start + dev, 1, 0, 0);
                                else if (d_ref >= -o->max_sv)        /* :163 */
                                    return mk(R_TANDEM, chr, c->ref_start, c->ref_start + dev, 0, 0, 0);
                                else                                 /* :166-168 */
                                    return mk(R_BND, chr, c->ref_start, REV, chr, n->ref_end - 1, REV);
                            }
                        }
                    }
                }
            }
        } else if (!c->is_reverse && n->is_reverse) { /* :172 */
            const int32_t d_ref = n->ref_end - c->ref_end;
            const int32_t dev = d_read - d_ref;
            if (-o->qot <= d_read && d_read <= o->qgt) {       /* :175 */
                if (n->ref_start - c->ref_end >= -o->rot) {    /* :176 case 1 */
                    if (o->min_sv <= -dev && -dev <= o->max_sv)
                        return mk(R_INV, chr, c->ref_end, c->ref_end - dev, 0, 0, 0); /* left_fwd */
                    return mk(R_BND, chr, c->ref_end - 1, FWD, chr, n->ref_end - 1, REV);
                } else if (c->ref_start - n->ref_end >= -o->rot) { /* :185 case 3 */
                    if (o->min_sv <= dev && dev <= o->max_sv)
                        return mk(R_INV, chr, n->ref_end, n->ref_end + dev, 1, 0, 0); /* left_rev */
                    return mk(R_BND, chr, c->ref_end - 1, FWD, chr, n->ref_end - 1, REV);
                }
            }
        } else { /* :198 reverse → forward */
            const int32_t d_ref = n->ref_start - c->ref_start;
            const int32_t dev = d_read - d_ref;
            if (-o->qot <= d_read && d_read <= o->qgt) {       /* :201 */
                if (n->ref_start - c->ref_end >= -o->rot) {    /* :202 case 2 */
                    if (o->min_sv <= -dev && -dev <= o->max_sv)
                        return mk(R_INV, chr, c->ref_start, c->ref_start - dev, 2, 0, 0); /* right_fwd */
                    return mk(R_BND, chr, c->ref_start, REV, chr, n->ref_start, FWD);
                } else if (c->ref_start - n->ref_end >= -o->rot) { /* :211 case 4 */
                    if (o->min_sv <= dev && dev <= o->max_sv)
                        return mk(R_INV, chr, n->ref_start, n->ref_start + dev, 3, 0, 0); /* right_rev */
                    return mk(R_BND, chr, c->ref_start, REV, chr, n->ref_start, FWD);
                }
            }
        }
    } else { /* :224 different contigs */
        if (d_read >= -o->qot && d_read <= o->qgt) { /* :230-232, :246-248 */
            if (c->is_reverse == n->is_reverse) {
                if (!c->is_reverse) return mk(R_BND, c->ref_id, c->ref_end - 1, FWD, n->ref_id, n->ref_start, FWD);
                else return mk(R_BND, c->ref_id, c->ref_start, REV, n->ref_id, n->ref_end - 1, REV);
            } else {
                if (!c->is_reverse) return mk(R_BND, c->ref_id, c->ref_end - 1, FWD, n->ref_id, n->ref_end - 1, REV);
                else return mk(R_BND, c->ref_id, c->ref_start, REV, n->ref_id, n->ref_start, FWD);
            }
        }
    }
    return mk(R_NONE, 0, 0, 0, 0, 0, 0);
}

void orc_segments_classify(const orc_seg* segs, const uint32_t* read_off, uint32_t n_reads,
                           const int32_t* read_len, const orc_params* params, orc_raw* out) {
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint32_t b = read_off[r], e = read_off[r + 1], k = e - b;
        if (k == 0) continue;
        orc_seg* s = (orc_seg*)malloc(sizeof(orc_seg) * k);
        memcpy(s, segs + b, sizeof(orc_seg) * k);
        /* stable insertion sort by (q_start, q_end) — SVIM_inter.py:83 */
        for (uint32_t i = 1; i < k; ++i) {
            orc_seg x = s[i];
            uint32_t j = i;
            while (j > 0 && (s[j - 1].q_start > x.q_start ||
                             (s[j - 1].q_start == x.q_start && s[j - 1].q_end > x.q_end))) {
                s[j] = s[j - 1];
                --j;
            }
            s[j] = x;
        }
        for (uint32_t i = 0; i + 1 < k; ++i) out[b + i] = classify_pair(&s[i], &s[i + 1], read_len[r], params);
        out[e - 1] = mk(R_NONE, 0, 0, 0, 0, 0, 0);
        free(s);
    }
}

/* ---- a5+a6: form_partitions (SVIM_COMBINE.py:15-32) on packed keys ----------------- */
typedef struct { uint64_t key; uint32_t idx; } orc_kv;
static int kv_cmp(const void* a, const void* b) {
    const orc_kv* x = (const orc_kv*)a; const orc_kv* y = (const orc_kv*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* stable: ties keep input order (:17) */
}
uint32_t orc_pair_partition(const uint64_t* keys, uint32_t n, uint32_t max_dist, uint32_t* perm,
                            uint32_t* part_id) {
    if (n == 0) return 0;
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * n);
    for (uint32_t i = 0; i < n; ++i) { kv[i].key = keys[i]; kv[i].idx = i; }
    qsort(kv, n, sizeof(orc_kv), kv_cmp);
    uint32_t pid = 0;
    for (uint32_t j = 0; j < n; ++j) {
        if (j > 0) {
            const uint64_t a = kv[j - 1].key, b = kv[j].key;
            const uint32_t pa = (uint32_t)a, pb = (uint32_t)b;
            const uint32_t d = pa > pb ? pa - pb : pb - pa;
            if ((a >> 32) != (b >> 32) || d > max_dist) ++pid; /* :24-26 */
        }
        perm[j] = kv[j].idx;
        part_id[j] = pid;
    }
    free(kv);
    return pid + 1;
}

/* ---- a7: exact global unit-cost edit distance (edlib.align default mode "NW",
 *      call sites SVIM_COMBINE.py:50,64,76,88,100) — textbook two-row DP ------------- */
uint32_t orc_edit_distance(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    if (la == 0) return lb;
    if (lb == 0) return la;
    uint32_t* row = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)lb + 1));
    for (uint32_t j = 0; j <= lb; ++j) row[j] = j;
    for (uint32_t i = 1; i <= la; ++i) {
        uint32_t diag = row[0];
        row[0] = i;
        for (uint32_t j = 1; j <= lb; ++j) {
            uint32_t up = row[j];
            uint32_t best = diag + (a[i - 1] != b[j - 1]);
            if (up + 1 < best) best = up + 1;
            if (row[j - 1] + 1 < best) best = row[j - 1] + 1;
            diag = up;
            row[j] = best;
        }
    }
    uint32_t d = row[lb];
    free(row);
    return d;
}

/* Same distance for long sequences: Ukkonen's diagonal band |i - j| <= k with the band doubled
 * until the result is <= k (then it is exact).  O(max(la, lb) * d) instead of O(la * lb); pinned
 * against orc_edit_distance by tests/test_oracle_pins.py.  Plain integer DP, no bit-vectors. */
static uint32_t banded_once(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, uint32_t k) {
    const uint32_t INF = 0x3FFFFFFFu;
    const uint32_t diff = la > lb ? la - lb : lb - la;
    if (diff > k) return INF;
    uint32_t* prev = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)lb + 2));
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)lb + 2));
    for (uint32_t j = 0; j <= lb + 1; ++j) { prev[j] = (j <= lb && j <= k) ? j : INF; cur[j] = INF; }
    for (uint32_t i = 1; i <= la; ++i) {
        const uint32_t lo = i > k ? i - k : 1;                      /* first column of the band, >= 1 */
        const uint32_t hi = (uint64_t)i + k < lb ? i + k : lb;      /* last column of the band        */
        cur[lo - 1] = (lo == 1 && i <= k) ? i : INF;                /* D[i][0] = i while inside the band */
        for (uint32_t j = lo; j <= hi; ++j) {
            uint32_t best = prev[j - 1] + (a[i - 1] != b[j - 1]);
            if (prev[j] + 1 < best) best = prev[j] + 1;
            if (cur[j - 1] + 1 < best) best = cur[j - 1] + 1;
            cur[j] = best > INF ? INF : best;
        }
        cur[hi + 1] = INF;                                          /* lb + 1 at most: allocated      */
        if (lo >= 2) cur[lo - 2] = INF;
        uint32_t* t = prev; prev = cur; cur = t;
    }
    const uint32_t d = prev[lb];
    free(prev);
    free(cur);
    return d;
}

uint32_t orc_edit_distance_banded(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    if (la == 0) return lb;
    if (lb == 0) return la;
    const uint32_t longest = la > lb ? la : lb;
    for (uint32_t k = 64;; k = k * 4) {
        if (k > longest) k = longest;
        const uint32_t d = banded_once(a, la, b, lb, k);
        if (d <= k) return d;
        if (k == longest) return d;
    }
}

/* ---- a7 / a3: complete linkage + flat cut, as scipy computes it ----------------------------------
 * The reference clusters every partition of 2..10 candidates with
 *     fcluster(linkage(distances, method="complete"), t, criterion="distance")
 * (SVIM_COMBINE.py:134-135,155-156) and every group of overlapping inversion breakpoints of a read
 * the same way (SVIM_inter.py:47-48).  Which member is cluster[0] — and supplies the coordinates of
 * the paired call — follows from scipy's cluster LABEL order, so the restatement has to reproduce
 * scipy's procedure, not just the set partition.  scipy (third-party, unpinned in the reference's
 * setup.py; 1.15.3 installed here) does, for method "complete" on a condensed distance vector:
 *   1. nearest-neighbour chain (Müllner): start the chain at the lowest live cluster; extend it with the
 *      nearest live cluster (strict <, lowest index wins, the previous chain element is preferred on
 *      ties); merge when two are mutual; the merged cluster keeps the larger index; Lance-Williams
 *      update for complete linkage d(z,i) = max(d(x,i), d(y,i));
 *   2. stable sort of the merges by distance;
 *   3. relabelling with a union-find (new cluster ids n, n+1, ... in sorted order; smaller root first);
 *   4. fcluster "distance": max distance in each subtree, then an explicit-stack traversal from the
 *      root, left child first; a node whose max distance <= t becomes one flat cluster, remaining
 *      leaves singletons; flat clusters are numbered in the order the traversal completes them.
 * Pinned against scipy itself by tests/test_oracle_pins.py (exhaustive rank patterns incl. ties for
 * n <= 6, random to n = 12).  labels[i] in 1..k.  n >= 1; scratch is heap-allocated. */
static size_t cond_index(uint32_t n, uint32_t i, uint32_t j) {
    if (i > j) { uint32_t t = i; i = j; j = t; }
    return (size_t)n * i - (size_t)i * (i + 1) / 2 + (j - i - 1);
}

void orc_linkage_cut(const double* cond, uint32_t n, double cutoff, uint32_t* labels) {
    if (n == 0) return;
    if (n == 1) { labels[0] = 1; return; }
    const size_t m = (size_t)n * (n - 1) / 2;
    double* D = (double*)malloc(sizeof(double) * m);
    memcpy(D, cond, sizeof(double) * m);
    int* size = (int*)malloc(sizeof(int) * n);
    int* chain = (int*)malloc(sizeof(int) * n);
    int* zx = (int*)malloc(sizeof(int) * (n - 1));
    int* zy = (int*)malloc(sizeof(int) * (n - 1));
    double* zd = (double*)malloc(sizeof(double) * (n - 1));
    for (uint32_t i = 0; i < n; ++i) size[i] = 1;
    int chain_len = 0;
    for (uint32_t k = 0; k + 1 < n; ++k) {
        int x = 0, y = 0;
        double cur = 0;
        if (chain_len == 0) {
            chain_len = 1;
            for (uint32_t i = 0; i < n; ++i)
                if (size[i] > 0) { chain[0] = (int)i; break; }
        }
        for (;;) {
            x = chain[chain_len - 1];
            if (chain_len > 1) {
                y = chain[chain_len - 2];
                cur = D[cond_index(n, (uint32_t)x, (uint32_t)y)];
            } else {
                cur = (double)INFINITY;
            }
            for (uint32_t i = 0; i < n; ++i) {
                if (size[i] == 0 || (int)i == x) continue;
                const double d = D[cond_index(n, (uint32_t)x, i)];
                if (d < cur) { cur = d; y = (int)i; }
            }
            if (chain_len > 1 && y == chain[chain_len - 2]) break;
            chain[chain_len++] = y;
        }
        chain_len -= 2;
        if (x > y) { int t = x; x = y; y = t; }
        const int nx = size[x], ny = size[y];
        zx[k] = x; zy[k] = y; zd[k] = cur;
        size[x] = 0;
        size[y] = nx + ny;
        for (uint32_t i = 0; i < n; ++i) {
            if (size[i] == 0 || (int)i == y) continue;
            const double a = D[cond_index(n, i, (uint32_t)x)], b = D[cond_index(n, i, (uint32_t)y)];
            D[cond_index(n, i, (uint32_t)y)] = a > b ? a : b;
        }
    }
    /* stable sort by distance (insertion sort) */
    for (uint32_t i = 1; i + 1 < n; ++i) {
        const int tx = zx[i], ty = zy[i];
        const double td = zd[i];
        uint32_t j = i;
        while (j > 0 && zd[j - 1] > td) { zx[j] = zx[j - 1]; zy[j] = zy[j - 1]; zd[j] = zd[j - 1]; --j; }
        zx[j] = tx; zy[j] = ty; zd[j] = td;
    }
    /* union-find relabelling */
    int* parent = (int*)malloc(sizeof(int) * (2 * (size_t)n - 1));
    for (uint32_t i = 0; i < 2 * n - 1; ++i) parent[i] = (int)i;
    int next = (int)n;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        int r[2] = {zx[i], zy[i]};
        for (int s = 0; s < 2; ++s) {
            int p = r[s], root = r[s];
            while (parent[root] != root) root = parent[root];
            while (parent[p] != root) { int q = parent[p]; parent[p] = root; p = q; }
            r[s] = root;
        }
        zx[i] = r[0] < r[1] ? r[0] : r[1];
        zy[i] = r[0] < r[1] ? r[1] : r[0];
        parent[r[0]] = next;
        parent[r[1]] = next;
        ++next;
    }
    /* max distance below every internal node (children were created earlier: lower rows) */
    double* md = (double*)malloc(sizeof(double) * (n - 1));
    for (uint32_t i = 0; i + 1 < n; ++i) {
        double v = zd[i];
        if (zx[i] >= (int)n && md[zx[i] - (int)n] > v) v = md[zx[i] - (int)n];
        if (zy[i] >= (int)n && md[zy[i] - (int)n] > v) v = md[zy[i] - (int)n];
        md[i] = v;
    }
    /* flat clusters: explicit-stack traversal, left child first */
    int* stack = (int*)malloc(sizeof(int) * n);
    unsigned char* visited = (unsigned char*)calloc(2 * (size_t)n - 1, 1);
    int kk = 0, n_cluster = 0, leader = -1;
    stack[0] = 2 * (int)n - 2;
    while (kk >= 0) {
        const int root = stack[kk] - (int)n;
        const int lc = zx[root], rc = zy[root];
        if (leader == -1 && md[root] <= cutoff) { leader = root; ++n_cluster; }
        if (lc >= (int)n && !visited[lc]) { visited[lc] = 1; stack[++kk] = lc; continue; }
        if (rc >= (int)n && !visited[rc]) { visited[rc] = 1; stack[++kk] = rc; continue; }
        if (lc < (int)n) { if (leader == -1) ++n_cluster; labels[lc] = (uint32_t)n_cluster; }
        if (rc < (int)n) { if (leader == -1) ++n_cluster; labels[rc] = (uint32_t)n_cluster; }
        if (leader == root) leader = -1;
        --kk;
    }
    free(D); free(size); free(chain); free(zx); free(zy); free(zd); free(parent); free(md); free(stack); free(visited);
}
