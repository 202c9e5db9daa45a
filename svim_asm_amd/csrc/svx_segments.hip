// svx_segments.hip — split-segment classification on gfx950.
//
// Replaces the adjacent-pair decision tree of analyze_read_segments
// (reference SVIM_inter.py:62-258): per read, stable sort of its segments by
// (q_start, q_end) (:83), then one raw record per adjacent pair (:91-258).
//
// Layout: segs[] AoS of 6 x i32 (24 B, svx_seg) grouped per read by read_off[]; out[] one
// 32-B svx_raw per segment slot (pair i of read r at read_off[r] + i).  Eight lanes per read, one
// lane per segment: reads carry 1..k segments with k tiny; ranking by width-8
// shuffles replaces the sort, reads with more than 8 segments take a serial path with an HBM
// scratch slice.  24 B in + 32 B out per segment; bound by launch latency, not bandwidth.
#include "svx_internal.h"
#include "svx_segments_dev.h"

using namespace svx_seg_dev;

namespace {

__global__ __launch_bounds__(256) void k_segments(SegArgs p) {
    const int lane = threadIdx.x & 63, gl = lane & (kGroup - 1), gbase = lane & ~(kGroup - 1);
    const uint32_t group = (blockIdx.x * 256 + threadIdx.x) / kGroup, n_groups = gridDim.x * 256 / kGroup;
    for (uint32_t r0 = 0; r0 < p.n_reads; r0 += n_groups) {
        const uint32_t r = r0 + group;
        segments_group(p, r, r < p.n_reads, gl, gbase);
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
    if ((reinterpret_cast<uintptr_t>(d_segs) & 7u) || (reinterpret_cast<uintptr_t>(d_out) & 15u)) {
        SVX_SET_ERR(ctx, "d_segs must be 8-byte aligned and d_out 16-byte aligned");
        return SVX_E_INVALID;
    }
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = svx_ws_reserve(ctx, svx_take_bytes(n_segs, sizeof(svx_seg)));
    if (rc != SVX_OK) return rc;
    SegArgs a;
    a.seg_rl = nullptr;
    a.read_len_out = nullptr;
    a.segs = d_segs;
    a.sorted = svx_ws_take<svx_seg>(ctx, n_segs);
    a.read_off = d_read_off;
    a.read_len = d_read_len;
    a.n_reads = n_reads;
    a.o = *params;
    a.out = d_out;
    uint32_t blocks = (n_reads + 31) / 32;  // eight lanes per read
    uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_segments, dim3(blocks < cap ? blocks : cap), dim3(256), 0, ctx->stream, a);
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
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
