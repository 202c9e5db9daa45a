// svx_segments.hip — split-segment classification on gfx950.
//
// Replaces the adjacent-pair decision tree of analyze_read_segments
// (reference SVIM_inter.py:62-258): per read, stable sort of its segments by
// (q_start, q_end) (:83), then one raw record per adjacent pair (:91-258).
//
// Layout: segs[] AoS of 6 x i32 (24 B, svx_seg) grouped per read by read_off[]; out[] one
// 32-B svx_raw per segment slot (pair i of read r at read_off[r] + i).  One lane per read:
// reads carry 1..k segments with k tiny, so the work is a few hundred
// integer compares per read; the sort runs in a per-read slice of an HBM scratch copy.
// 24 B in + 32 B out per segment; bound by launch latency, not bandwidth.
#include "svx_internal.h"

namespace {

struct SegArgs {
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
__device__ svx_raw classify(const svx_seg& cur, const svx_seg& nxt, int32_t read_len,
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

__global__ __launch_bounds__(256) void k_segments(SegArgs p) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < p.n_reads;
         r += gridDim.x * blockDim.x) {
        const uint32_t b = p.read_off[r], e = p.read_off[r + 1];
        if (e <= b) continue;
        svx_seg* s = p.sorted + b;
        const uint32_t k = e - b;
        // stable insertion sort by (q_start, q_end) into the scratch slice (:83)
        for (uint32_t i = 0; i < k; ++i) {
            const svx_seg x = p.segs[b + i];
            uint32_t j = i;
            while (j > 0) {
                const svx_seg y = s[j - 1];
                if (y.q_start > x.q_start || (y.q_start == x.q_start && y.q_end > x.q_end)) {
                    s[j] = y;
                    --j;
                } else {
                    break;
                }
            }
            s[j] = x;
        }
        const int32_t rl = p.read_len[r];
        svx_seg cur = s[0];
        for (uint32_t i = 0; i + 1 < k; ++i) {
            const svx_seg nxt = s[i + 1];
            p.out[b + i] = classify(cur, nxt, rl, p.o);
            cur = nxt;
        }
        p.out[e - 1] = raw(SVX_RAW_NONE);
    }
}

}  // namespace

extern "C" int svx_segments_classify_dev(svx_ctx* ctx, const svx_seg* d_segs, uint32_t n_segs,
                                         const uint32_t* d_read_off, uint32_t n_reads,
                                         const int32_t* d_read_len, const svx_seg_params* params,
                                         svx_raw* d_out) {
    if (!ctx || !params) return SVX_E_INVALID;
    if (n_reads == 0 || n_segs == 0) return SVX_OK;
    if (!d_segs || !d_read_off || !d_read_len || !d_out) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = svx_ws_reserve(ctx, svx_take_bytes(n_segs, sizeof(svx_seg)));
    if (rc != SVX_OK) return rc;
    SegArgs a;
    a.segs = d_segs;
    a.sorted = svx_ws_take<svx_seg>(ctx, n_segs);
    a.read_off = d_read_off;
    a.read_len = d_read_len;
    a.n_reads = n_reads;
    a.o = *params;
    a.out = d_out;
    uint32_t blocks = (n_reads + 255) / 256;
    uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    svx_timing_mark(ctx, 1);
    hipLaunchKernelGGL(k_segments, dim3(blocks < cap ? blocks : cap), dim3(256), 0, ctx->stream, a);
    svx_timing_mark(ctx, 2);
    SVX_HIP(ctx, hipGetLastError());
    return svx_timing_end(ctx);
}

extern "C" int svx_segments_classify(svx_ctx* ctx, const svx_seg* segs, const uint32_t* read_off,
                                     uint32_t n_reads, const int32_t* read_len,
                                     const svx_seg_params* params, svx_raw* out) {
    if (!ctx || !params) return SVX_E_INVALID;
    if (n_reads == 0) return SVX_OK;
    if (!read_off || !read_len) return SVX_E_INVALID;
    for (uint32_t r = 0; r < n_reads; ++r)
        if (read_off[r + 1] < read_off[r]) {
            SVX_SET_ERR(ctx, "read_off must be non-decreasing (index %u)", r);
            return SVX_E_INVALID;
        }
    const uint32_t n_segs = read_off[n_reads];
    if (n_segs == 0) return SVX_OK;
    if (!segs || !out || read_off[0] != 0) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n_segs, sizeof(svx_seg)) + svx_take_bytes((size_t)n_reads + 1, 4) +
                  svx_take_bytes(n_reads, 4) + svx_take_bytes(n_segs, sizeof(svx_raw));
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    svx_seg* d_s = svx_stage_take<svx_seg>(ctx, n_segs);
    uint32_t* d_off = svx_stage_take<uint32_t>(ctx, (size_t)n_reads + 1);
    int32_t* d_rl = svx_stage_take<int32_t>(ctx, n_reads);
    svx_raw* d_o = svx_stage_take<svx_raw>(ctx, n_segs);
    SVX_HIP(ctx, hipMemcpyAsync(d_s, segs, (size_t)n_segs * sizeof(svx_seg), hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_off, read_off, ((size_t)n_reads + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_rl, read_len, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = svx_segments_classify_dev(ctx, d_s, n_segs, d_off, n_reads, d_rl, params, d_o);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(out, d_o, (size_t)n_segs * sizeof(svx_raw), hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
