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
#include "svx_linkage_dev.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int kThreads = 64;

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

__global__ __launch_bounds__(kThreads) void k_segments_post(PostArgs p) {
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= p.n_reads) return;
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
