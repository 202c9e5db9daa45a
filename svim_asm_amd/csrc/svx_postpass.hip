// svx_postpass.hip — the three per-read post-passes of analyze_read_segments on gfx950.
//
// Reference: SVIM_inter.py:260-338 with process_overlapping_inversions (:42-60),
// reciprocal_overlap_distance (:19-39) and is_similar (:12-16).  Input: the raw adjacency records of
// svx_segments_classify (one slot per segment, grouped per read); output: the derived candidates of
// each read, in the reference's order:
//   1. tandem duplications (:261-290): sweep over the TANDEM records in emission order, merging a record
//      into the running group when chromosome matches, |mean(starts) - start| < 20, |mean(ends) - end| < 20
//      (statistics.mean of ints is an exact rational: compared as |Σ - start·n| < 20·n) and its direction
//      equals that of the read's FIRST tandem record — the reference never updates current_direction
//      when it restarts a group; flushed as (chrom, int(mean(starts)), int(mean(ends)), copies, any(fully));
//   2. interspersed duplications (:293-320): every ordered pair (earlier, later) of BND records with
//      mirrored directions, the earlier one's source within 20 bp of the later one's destination, same
//      source chromosome and equal directions inside the earlier record;
//   3. inversions (:323-338): INV records sorted by (chromosome NAME, start, end) — the host passes the
//      rank of every contig name under Python str ordering —, swept into groups of overlapping
//      breakpoints (the breakpoint that closes a group is dropped, as in the reference), each group
//      clustered by complete linkage over the reciprocal-overlap distance (float64) cut at 0.3, clusters in
//      scipy's label order (svx_linkage_dev.h), each cluster → (chrom of its first member, max start,
//      min end, complete = more than one member).
// One lane per read: the passes are short sequential programs over a handful of records (a read has
// 2-5 segments); thousands of reads run side by side.  Scratch (sorted inversion list, distance vector,
// linkage state) is a per-read slice of the HBM workspace sized by the host from the read's slot count.
#include "svx_internal.h"
#include "svx_postpass_dev.h"

#include <algorithm>
#include <vector>

using namespace svx_post_dev;

namespace {

constexpr int kThreads = 64;

__global__ __launch_bounds__(kThreads) void k_segments_post(PostArgs p) {
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r < p.n_reads) post_one_read(p, r);
}

// scratch_off[r] = Σ_{r' < r} align16(post_scratch_bytes(slots of r')): one workgroup, each thread a contiguous chunk
__global__ __launch_bounds__(256) void k_post_scratch_offsets(const uint32_t* read_off, uint32_t n_reads, uint64_t* scratch_off) {
    __shared__ uint64_t s_part[256];
    const uint32_t chunk = (n_reads + 255) / 256;
    const uint32_t lo = threadIdx.x * chunk, hi = min(n_reads, lo + chunk);
    uint64_t sum = 0;
    for (uint32_t r = lo; r < hi; ++r) sum += (post_scratch_bytes(read_off[r + 1] - read_off[r]) + 15) / 16 * 16;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t) base += s_part[t];
    for (uint32_t r = lo; r < hi; ++r) {
        scratch_off[r] = base;
        base += (post_scratch_bytes(read_off[r + 1] - read_off[r]) + 15) / 16 * 16;
    }
}

uint64_t post_bound(uint32_t s) { return (uint64_t)s * ((uint64_t)s + 3) / 2; }

}  // namespace

extern "C" uint64_t svx_segments_postpass_bound(uint32_t n_slots) { return post_bound(n_slots); }

// one slice size for every read, if that wastes little: the largest read's slice, at most 64 MiB in total
static uint64_t post_uniform_stride(const uint32_t* read_off, uint32_t n_reads) {
    uint32_t s_max = 0;
    for (uint32_t r = 0; r < n_reads; ++r) s_max = std::max(s_max, read_off[r + 1] - read_off[r]);
    const uint64_t stride = svx_align_up(post_scratch_bytes(s_max), 16);
    return stride * n_reads <= (64ull << 20) ? stride : 0;
}

size_t svx_postpass_ws_need(const uint32_t* read_off, uint32_t n_reads) {
    const uint64_t stride = post_uniform_stride(read_off, n_reads);
    if (stride) return svx_take_bytes((size_t)n_reads * stride, 1);
    uint64_t n_scratch = 0;
    for (uint32_t r = 0; r < n_reads; ++r) n_scratch += svx_align_up(post_scratch_bytes(read_off[r + 1] - read_off[r]), 16);
    return svx_take_bytes((size_t)n_reads + 1, 8) + svx_take_bytes(n_scratch ? n_scratch : 1, 1);
}

// shared front end: validates the host copy of read_off / out_off and lays out the per-read scratch slices
static int post_plan(svx_ctx* ctx, const uint32_t* read_off, uint32_t n_reads, const uint64_t* out_off,
                     std::vector<uint64_t>* scratch_off, uint64_t* n_scratch) {
    if (read_off[0] != 0) return SVX_E_INVALID;
    scratch_off->resize(n_reads);
    *n_scratch = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (read_off[r + 1] < read_off[r] || out_off[r + 1] < out_off[r]) {
            SVX_SET_ERR(ctx, "read_off / out_off must be non-decreasing (index %u)", r);
            return SVX_E_INVALID;
        }
        const uint32_t s = read_off[r + 1] - read_off[r];
        if (out_off[r + 1] - out_off[r] < post_bound(s)) {
            SVX_SET_ERR(ctx, "read %u: %llu output slots for %u segment slots, svx_segments_postpass_bound() = %llu", r,
                        (unsigned long long)(out_off[r + 1] - out_off[r]), s, (unsigned long long)post_bound(s));
            return SVX_E_CAPACITY;
        }
        (*scratch_off)[r] = *n_scratch;
        *n_scratch += svx_align_up(post_scratch_bytes(s), 16);
    }
    return SVX_OK;
}

int svx_postpass_plan(svx_ctx* ctx, const uint32_t* read_off, uint32_t n_reads, const uint64_t* out_off, uint64_t* stride) {
    std::vector<uint64_t> scratch_off;
    uint64_t n_scratch = 0;
    int rc = post_plan(ctx, read_off, n_reads, out_off, &scratch_off, &n_scratch);
    if (rc != SVX_OK) return rc;
    *stride = post_uniform_stride(read_off, n_reads);
    return SVX_OK;
}

extern "C" int svx_segments_postpass_dev(svx_ctx* ctx, const svx_raw* d_raw, const uint32_t* read_off,
                                         const uint32_t* d_read_off, uint32_t n_reads, const int32_t* d_contig_rank,
                                         uint32_t n_contigs, const svx_seg_params* params, svx_post* d_out,
                                         const uint64_t* out_off, const uint64_t* d_out_off, uint32_t* d_out_cnt) {
    if (!ctx || !params) return SVX_E_INVALID;
    if (n_reads == 0) return SVX_OK;
    if (!read_off || !d_read_off || !out_off || !d_out_off || !d_out_cnt || (n_contigs && !d_contig_rank)) return SVX_E_INVALID;
    std::vector<uint64_t> scratch_off;
    uint64_t n_scratch = 0;
    int rc = post_plan(ctx, read_off, n_reads, out_off, &scratch_off, &n_scratch);
    if (rc != SVX_OK) return rc;
    if (read_off[n_reads] && (!d_raw || !d_out)) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    rc = svx_ws_reserve(ctx, svx_postpass_ws_need(read_off, n_reads));
    if (rc != SVX_OK) return rc;
    // scratch slices: one size for all reads when that is cheap (reads have a handful of segments: no offset table,
    // one launch less); otherwise laid out on the device from d_read_off.  Nothing of this call's host memory is
    // read after it returns: the call is asynchronous
    const uint64_t stride = post_uniform_stride(read_off, n_reads);
    uint64_t* d_soff = nullptr;
    char* d_scratch;
    if (stride) {
        d_scratch = svx_ws_take<char>(ctx, (size_t)n_reads * stride);
    } else {
        d_soff = svx_ws_take<uint64_t>(ctx, (size_t)n_reads + 1);
        d_scratch = svx_ws_take<char>(ctx, n_scratch ? n_scratch : 1);
        hipLaunchKernelGGL(k_post_scratch_offsets, dim3(1), dim3(256), 0, ctx->stream, d_read_off, n_reads, d_soff);
    }
    PostArgs a;
    a.scratch_stride = stride;
    a.raw = d_raw; a.read_off = d_read_off; a.n_reads = n_reads; a.contig_rank = d_contig_rank; a.n_contigs = n_contigs;
    a.min_sv = params->min_sv_size; a.max_sv = params->max_sv_size;
    a.out = d_out; a.out_off = d_out_off; a.out_cnt = d_out_cnt; a.scratch = d_scratch; a.scratch_off = d_soff;
    hipLaunchKernelGGL(k_segments_post, dim3((n_reads + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    return SVX_OK;
}

extern "C" int svx_segments_postpass(svx_ctx* ctx, const svx_raw* raw, const uint32_t* read_off, uint32_t n_reads,
                                     const int32_t* contig_rank, uint32_t n_contigs, const svx_seg_params* params,
                                     svx_post* out, const uint64_t* out_off, uint32_t* out_cnt) {
    if (!ctx || !params) return SVX_E_INVALID;
    if (n_reads == 0) return SVX_OK;
    if (!read_off || !out_off || !out_cnt || (n_contigs && !contig_rank)) return SVX_E_INVALID;
    if (read_off[0] != 0) return SVX_E_INVALID;
    std::vector<uint64_t> scratch_off(n_reads);
    uint64_t n_scratch = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (read_off[r + 1] < read_off[r] || out_off[r + 1] < out_off[r]) {
            SVX_SET_ERR(ctx, "read_off / out_off must be non-decreasing (index %u)", r);
            return SVX_E_INVALID;
        }
        const uint32_t s = read_off[r + 1] - read_off[r];
        if (out_off[r + 1] - out_off[r] < post_bound(s)) {
            SVX_SET_ERR(ctx, "read %u: %llu output slots for %u segment slots, svx_segments_postpass_bound() = %llu", r,
                        (unsigned long long)(out_off[r + 1] - out_off[r]), s, (unsigned long long)post_bound(s));
            return SVX_E_CAPACITY;
        }
        scratch_off[r] = n_scratch;
        n_scratch += svx_align_up(post_scratch_bytes(s), 16);
    }
    const uint32_t n_segs = read_off[n_reads];
    const uint64_t n_out = out_off[n_reads];
    if (n_segs && (!raw || !out)) return SVX_E_INVALID;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n_segs ? n_segs : 1, sizeof(svx_raw)) + svx_take_bytes((size_t)n_reads + 1, 4) +
                  svx_take_bytes(n_contigs ? n_contigs : 1, 4) + svx_take_bytes(n_out ? n_out : 1, sizeof(svx_post)) +
                  2 * svx_take_bytes((size_t)n_reads + 1, 8) + svx_take_bytes(n_reads, 4) +
                  svx_take_bytes(n_scratch ? n_scratch : 1, 1);
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    PostArgs a;
    svx_raw* d_raw = svx_stage_take<svx_raw>(ctx, n_segs ? n_segs : 1);
    uint32_t* d_off = svx_stage_take<uint32_t>(ctx, (size_t)n_reads + 1);
    int32_t* d_rank = svx_stage_take<int32_t>(ctx, n_contigs ? n_contigs : 1);
    svx_post* d_out = svx_stage_take<svx_post>(ctx, n_out ? n_out : 1);
    uint64_t* d_ooff = svx_stage_take<uint64_t>(ctx, (size_t)n_reads + 1);
    uint64_t* d_soff = svx_stage_take<uint64_t>(ctx, (size_t)n_reads + 1);
    uint32_t* d_cnt = svx_stage_take<uint32_t>(ctx, n_reads);
    char* d_scratch = svx_stage_take<char>(ctx, n_scratch ? n_scratch : 1);
    if (n_segs) SVX_HIP(ctx, hipMemcpyAsync(d_raw, raw, (size_t)n_segs * sizeof(svx_raw), hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_off, read_off, ((size_t)n_reads + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_contigs) SVX_HIP(ctx, hipMemcpyAsync(d_rank, contig_rank, (size_t)n_contigs * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_ooff, out_off, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_soff, scratch_off.data(), (size_t)n_reads * 8, hipMemcpyHostToDevice, ctx->stream));
    a.raw = d_raw; a.read_off = d_off; a.n_reads = n_reads; a.contig_rank = d_rank; a.n_contigs = n_contigs;
    a.min_sv = params->min_sv_size; a.max_sv = params->max_sv_size;
    a.out = d_out; a.out_off = d_ooff; a.out_cnt = d_cnt; a.scratch = d_scratch; a.scratch_off = d_soff;
    a.scratch_stride = 0;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_segments_post, dim3((n_reads + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_end(ctx);
    if (rc != SVX_OK) return rc;
    if (n_out) SVX_HIP(ctx, hipMemcpyAsync(out, d_out, (size_t)n_out * sizeof(svx_post), hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(out_cnt, d_cnt, (size_t)n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
