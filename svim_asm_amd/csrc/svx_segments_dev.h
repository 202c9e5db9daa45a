// svx_segments_dev.h — device routines of the split-segment decision tree (SVIM_inter.py:83-258), shared by
// k_segments (svx_segments.hip) and the fused split-segment chain of svx_collect_batch_dev (svx_cigar.hip).
#pragma once
#include "svx_internal.h"

namespace svx_seg_dev {

struct SegArgs {
    // fused chain only (else null): per-SEGMENT read lengths — a read's length is its first segment's — and where
    // the per-read lengths are published
    const int32_t* seg_rl;
    int32_t* read_len_out;
    const svx_seg* segs;
    svx_seg* sorted;  // scratch, n_segs
    const uint32_t* read_off;
    const int32_t* read_len;
    uint32_t n_reads;
    svx_seg_params o;
    svx_raw* out;
};

__device__ __forceinline__ svx_raw raw(int kind, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0,
                                       int a4 = 0, int a5 = 0) {
    svx_raw r;
    r.kind = kind; r.a0 = a0; r.a1 = a1; r.a2 = a2; r.a3 = a3; r.a4 = a4; r.a5 = a5; r.pad = 0;
    return r;
}

constexpr int kFwd = 0, kRev = 1;

// cur = segment earlier on the read, nxt = the following one (SVIM_inter.py:92-93)
__device__ __forceinline__ svx_raw classify(const svx_seg& cur, const svx_seg& nxt, int32_t read_len,
                            const svx_seg_params& o) {
    const int32_t gap_q = nxt.q_start - cur.q_end;  // distance_on_read (:95)
    const bool q_no_overlap = gap_q >= -o.query_overlap_tolerance;
    const bool q_no_gap = gap_q <= o.query_gap_tolerance;
    const bool cr = cur.is_reverse != 0, nr = nxt.is_reverse != 0;

    if (cur.ref_id != nxt.ref_id) {
        // different contigs (:224-258): breakend when the read positions abut
        if (!(q_no_overlap && q_no_gap)) return raw(SVX_RAW_NONE);
        const int p1 = cr ? cur.ref_start : cur.ref_end - 1;
        int p2;
        if (cr == nr) p2 = cr ? nxt.ref_end - 1 : nxt.ref_start;
        else p2 = cr ? nxt.ref_start : nxt.ref_end - 1;
        return raw(SVX_RAW_BND, cur.ref_id, p1, cr ? kRev : kFwd, nxt.ref_id, p2, nr ? kRev : kFwd);
    }

    const int chr = cur.ref_id;
    if (cr == nr) {
        // same strand (:101-168)
        const int32_t gap_r = cr ? cur.ref_start - nxt.ref_end : nxt.ref_start - cur.ref_end;
        if (!q_no_overlap) return raw(SVX_RAW_NONE);
        const int32_t dev = gap_q - gap_r;
        if (gap_r >= -o.reference_overlap_tolerance) {
            if (dev >= o.min_sv_size) {  // insertion (:113-121)
                if (gap_r > o.reference_gap_tolerance) return raw(SVX_RAW_NONE);
                if (!cr) return raw(SVX_RAW_INS, chr, cur.ref_end, cur.ref_end + dev, cur.q_end, dev);
                return raw(SVX_RAW_INS, chr, cur.ref_start, cur.ref_start + dev,
                           read_len - nxt.q_start, dev);
            }
            if (-o.max_sv_size <= dev && dev <= -o.min_sv_size) {  // deletion (:123-129)
                if (!q_no_gap) return raw(SVX_RAW_NONE);
                const int s = cr ? nxt.ref_end : cur.ref_end;
                return raw(SVX_RAW_DEL, chr, s, s - dev);
            }
            if (dev < -o.max_sv_size) {  // very large deletion or translocation (:131-139)
                if (!q_no_gap) return raw(SVX_RAW_NONE);
                if (!cr) return raw(SVX_RAW_BND, chr, cur.ref_end - 1, kFwd, chr, nxt.ref_start, kFwd);
                return raw(SVX_RAW_BND, chr, cur.ref_start, kRev, chr, nxt.ref_end - 1, kRev);
            }
            return raw(SVX_RAW_NONE);
        }
        // segments overlap on the reference (:141-168)
        if (!q_no_gap || dev < o.min_sv_size) return raw(SVX_RAW_NONE);
        if (!cr) {
            if (nxt.ref_end > cur.ref_start)
                return raw(SVX_RAW_TANDEM, chr, nxt.ref_start, nxt.ref_start + dev, 1, 1);
            if (gap_r >= -o.max_sv_size)
                return raw(SVX_RAW_TANDEM, chr, nxt.ref_start, nxt.ref_start + dev, 0, 1);
            return raw(SVX_RAW_BND, chr, cur.ref_end - 1, kFwd, chr, nxt.ref_start, kFwd);
        }
        if (nxt.ref_start < cur.ref_end)
            return raw(SVX_RAW_TANDEM, chr, cur.ref_start, cur.ref_start + dev, 1, 0);
        if (gap_r >= -o.max_sv_size)
            return raw(SVX_RAW_TANDEM, chr, cur.ref_start, cur.ref_start + dev, 0, 0);
        return raw(SVX_RAW_BND, chr, cur.ref_start, kRev, chr, nxt.ref_end - 1, kRev);
    }

    // opposite strands on one contig (:170-222)
    if (!(q_no_overlap && q_no_gap)) return raw(SVX_RAW_NONE);
    const bool case_a = nxt.ref_start - cur.ref_end >= -o.reference_overlap_tolerance;  // cases 1, 2
    const bool case_b = cur.ref_start - nxt.ref_end >= -o.reference_overlap_tolerance;  // cases 3, 4
    if (!case_a && !case_b) return raw(SVX_RAW_NONE);
    if (!cr) {  // forward → reverse (:172-193)
        const int32_t dev = gap_q - (nxt.ref_end - cur.ref_end);
        if (case_a) {
            if (o.min_sv_size <= -dev && -dev <= o.max_sv_size)
                return raw(SVX_RAW_INV, chr, cur.ref_end, cur.ref_end - dev, 0);
        } else {
            if (o.min_sv_size <= dev && dev <= o.max_sv_size)
                return raw(SVX_RAW_INV, chr, nxt.ref_end, nxt.ref_end + dev, 1);
        }
        return raw(SVX_RAW_BND, chr, cur.ref_end - 1, kFwd, chr, nxt.ref_end - 1, kRev);
    }
    // reverse → forward (:198-219)
    const int32_t dev = gap_q - (nxt.ref_start - cur.ref_start);
    if (case_a) {
        if (o.min_sv_size <= -dev && -dev <= o.max_sv_size)
            return raw(SVX_RAW_INV, chr, cur.ref_start, cur.ref_start - dev, 2);
    } else {
        if (o.min_sv_size <= dev && dev <= o.max_sv_size)
            return raw(SVX_RAW_INV, chr, nxt.ref_start, nxt.ref_start + dev, 3);
    }
    return raw(SVX_RAW_BND, chr, cur.ref_start, kRev, chr, nxt.ref_start, kFwd);
}

// serial path for reads with more than 8 segments: stable insertion sort in the HBM scratch slice
__device__ __forceinline__ void segments_serial(const SegArgs& p, uint32_t r) {
    const uint32_t b = p.read_off[r], e = p.read_off[r + 1];
    svx_seg* s = p.sorted + b;
    const uint32_t k = e - b;
    for (uint32_t i = 0; i < k; ++i) {
        // only the sort key stays in registers across the shifting loop; the record is read again for its store
        const int32_t x_start = p.segs[b + i].q_start, x_end = p.segs[b + i].q_end;
        uint32_t j = i;
        while (j > 0) {
            const svx_seg y = s[j - 1];
            if (y.q_start > x_start || (y.q_start == x_start && y.q_end > x_end)) {
                s[j] = y;
                --j;
            } else {
                break;
            }
        }
        s[j] = p.segs[b + i];
    }
    const int32_t rl = p.seg_rl ? p.seg_rl[b] : p.read_len[r];
    svx_seg cur = s[0];
    for (uint32_t i = 0; i + 1 < k; ++i) {
        const svx_seg nxt = s[i + 1];
        p.out[b + i] = classify(cur, nxt, rl, p.o);
        cur = nxt;
    }
    p.out[e - 1] = raw(SVX_RAW_NONE);
}

// Eight reads per wave: a group of 8 lanes owns one read, one lane per segment (reads carry a
// handful of segments).  Every lane ranks its segment by (q_start, q_end, original index) against
// the others of its group with width-8 shuffles (= the stable sort of SVIM_inter.py:83), segments
// move to their sorted lane with ds_permute, and lane i classifies the adjacent pair (i, i+1).
// The dependent-load chain per read is read_off → segments → store; reads with more than 8
// segments take the serial path (HBM scratch slice) on their group's first lane.
constexpr int kGroup = 8;

// One read per group of eight lanes (`gl` = lane inside the group, `gbase` = the group's first lane of the wave);
// all 64 lanes of the wave call this together (the ranking uses width-8 shuffles).
// (b, e = read_off[r], read_off[r + 1], zero for a group that is not live)
__device__ __forceinline__ void segments_group(const SegArgs& p, const uint32_t r, const bool live, const int gl, const int gbase,
                                               const uint32_t b, const uint32_t e) {
    const uint32_t k = e > b ? e - b : 0;
    const bool small = k <= (uint32_t)kGroup;
    if (!small && gl == 0) segments_serial(p, r);
    svx_seg s;
    s.q_start = s.q_end = s.ref_id = s.ref_start = s.ref_end = s.is_reverse = 0;
    int32_t rl = 0;
    if (small && (uint32_t)gl < k) {
        // 24-byte records, 8-byte aligned: three 8-byte loads instead of six dwords
        const uint2* q = reinterpret_cast<const uint2*>(p.segs + b + gl);
        const uint2 w0 = q[0], w1 = q[1], w2 = q[2];
        s.q_start = (int32_t)w0.x; s.q_end = (int32_t)w0.y; s.ref_id = (int32_t)w1.x;
        s.ref_start = (int32_t)w1.y; s.ref_end = (int32_t)w2.x; s.is_reverse = (int32_t)w2.y;
        rl = p.seg_rl ? p.seg_rl[b] : p.read_len[r];
    }
    if (p.read_len_out && live && gl == 0) p.read_len_out[r] = k ? (p.seg_rl ? p.seg_rl[b] : p.read_len[r]) : 0;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        const int qs = __shfl(s.q_start, j, kGroup), qe = __shfl(s.q_end, j, kGroup);
        const bool before = qs < s.q_start || (qs == s.q_start && (qe < s.q_end || (qe == s.q_end && j < gl)));
        rank += (before && (uint32_t)j < k) ? 1 : 0;
    }
    if (!small || (uint32_t)gl >= k) rank = gl;  // idle lanes keep distinct destinations
    const int dst = (gbase + rank) << 2;
    svx_seg t;
    t.q_start = __builtin_amdgcn_ds_permute(dst, s.q_start);
    t.q_end = __builtin_amdgcn_ds_permute(dst, s.q_end);
    t.ref_id = __builtin_amdgcn_ds_permute(dst, s.ref_id);
    t.ref_start = __builtin_amdgcn_ds_permute(dst, s.ref_start);
    t.ref_end = __builtin_amdgcn_ds_permute(dst, s.ref_end);
    t.is_reverse = __builtin_amdgcn_ds_permute(dst, s.is_reverse);
    svx_seg n;
    n.q_start = __shfl_down(t.q_start, 1, kGroup);
    n.q_end = __shfl_down(t.q_end, 1, kGroup);
    n.ref_id = __shfl_down(t.ref_id, 1, kGroup);
    n.ref_start = __shfl_down(t.ref_start, 1, kGroup);
    n.ref_end = __shfl_down(t.ref_end, 1, kGroup);
    n.is_reverse = __shfl_down(t.is_reverse, 1, kGroup);
    if (small && k > 0 && (uint32_t)gl < k) {
        const svx_raw v = (uint32_t)gl + 1 < k ? classify(t, n, rl, p.o) : raw(SVX_RAW_NONE);
        // 32-byte records, 16-byte aligned: two 16-byte stores
        uint4* o = reinterpret_cast<uint4*>(p.out + b + gl);
        o[0] = make_uint4((uint32_t)v.kind, (uint32_t)v.a0, (uint32_t)v.a1, (uint32_t)v.a2);
        o[1] = make_uint4((uint32_t)v.a3, (uint32_t)v.a4, (uint32_t)v.a5, 0u);
    }
}

__device__ __forceinline__ void segments_group(const SegArgs& p, const uint32_t r, const bool live, const int gl, const int gbase) {
    uint32_t b = 0, e = 0;
    if (live) { b = p.read_off[r]; e = p.read_off[r + 1]; }
    segments_group(p, r, live, gl, gbase, b, e);
}

}  // namespace svx_seg_dev
