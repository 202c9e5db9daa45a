// svx_linkage.hip — batched complete linkage + flat cut on gfx950, one thread per partition.
//
// Replaces, for every partition of the PAIR step and every group of overlapping inversion
// breakpoints of a read,
//     fcluster(linkage(distances, method="complete"), t, criterion="distance")
// (reference SVIM_COMBINE.py:134-135,155-156 and SVIM_inter.py:47-48; scipy.cluster.hierarchy).
// The flat-cluster LABELS decide which member is cluster[0], i.e. whose coordinates the paired call
// carries (SVIM_COMBINE.py:184-363), so scipy's procedure is reproduced step by step:
//   nearest-neighbour chain with its tie rules (strict <, lowest index, previous chain element
//   preferred), merged cluster keeps the larger index, complete-linkage update max(d(x,i), d(y,i));
//   stable sort of the merges by distance; union-find relabelling (smaller root first); maximum
//   distance per subtree; explicit-stack traversal from the root, left child first, numbering flat
//   clusters as the traversal completes them.
// Partitions are tiny (2..10 members in PAIR) and there are thousands of them: the work is a short
// sequential program per partition, so the mapping is one lane per partition with its whole state
// (distance matrix, merge list, union-find, stack) in a private LDS slice — no global traffic besides
// the input vector and the labels.  Partitions with more members than the LDS slice holds (inversion
// groups of pathological reads) run the same code on a slice of the HBM workspace.
// Double precision throughout (scipy computes in float64; the cut compares with <=): only
// comparisons, max and copies — no arithmetic that could round differently.
#include "svx_internal.h"
#include "svx_linkage_dev.h"

#include <vector>

namespace {

constexpr int kThreads = 64;
constexpr uint32_t kLdsN = 10;  // partitions up to this size keep their state in LDS

constexpr size_t kSlice = (svx_link_bytes(kLdsN) + 15) / 16 * 16;

struct LinkArgs {
    const double* dist;          // condensed vectors, partition after partition
    const uint64_t* dist_off;    // [n_parts] first element of partition p
    const uint32_t* n_members;   // [n_parts]
    const uint64_t* label_off;   // [n_parts] first label of partition p
    const uint64_t* scratch_off; // [n_parts] byte offset into `scratch` (large partitions only)
    char* scratch;
    uint32_t n_parts;
    double cutoff;
    uint32_t* labels;
};

__global__ __launch_bounds__(kThreads) void k_linkage_cut(LinkArgs a) {
    __shared__ __attribute__((aligned(16))) char s_mem[kThreads * kSlice];
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= a.n_parts) return;
    const uint32_t n = a.n_members[p];
    char* mem = n <= kLdsN ? s_mem + (size_t)threadIdx.x * kSlice : a.scratch + a.scratch_off[p];
    svx_linkage_cut_one(n, a.dist + a.dist_off[p], a.cutoff, a.labels + a.label_off[p], mem);
}

// offsets of partition p in the distance / label / scratch arrays, from the member counts: one workgroup, each
// thread a contiguous chunk (the asynchronous entry point reads no host memory after it returns)
__global__ __launch_bounds__(256) void k_linkage_offsets(const uint32_t* n_members, uint32_t n_parts, uint64_t* dist_off,
                                                         uint64_t* label_off, uint64_t* scratch_off) {
    __shared__ uint64_t s_d[256], s_l[256], s_s[256];
    const uint32_t chunk = (n_parts + 255) / 256;
    const uint32_t lo = threadIdx.x * chunk, hi = min(n_parts, lo + chunk);
    uint64_t d = 0, l = 0, sc = 0;
    for (uint32_t p = lo; p < hi; ++p) {
        const uint64_t n = n_members[p];
        d += n * (n ? n - 1 : 0) / 2;
        l += n;
        if (n > kLdsN) sc += (svx_link_bytes((uint32_t)n) + 15) / 16 * 16;
    }
    s_d[threadIdx.x] = d; s_l[threadIdx.x] = l; s_s[threadIdx.x] = sc;
    __syncthreads();
    d = l = sc = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t) { d += s_d[t]; l += s_l[t]; sc += s_s[t]; }
    for (uint32_t p = lo; p < hi; ++p) {
        const uint64_t n = n_members[p];
        dist_off[p] = d; label_off[p] = l; scratch_off[p] = sc;
        d += n * (n ? n - 1 : 0) / 2;
        l += n;
        if (n > kLdsN) sc += (svx_link_bytes((uint32_t)n) + 15) / 16 * 16;
    }
}

}  // namespace

extern "C" int svx_linkage_cut_batch_dev(svx_ctx* ctx, const double* d_dist, const uint32_t* n_members,
                                         const uint32_t* d_n_members, uint32_t n_parts, double cutoff, uint32_t* d_labels) {
    if (!ctx) return SVX_E_INVALID;
    if (n_parts == 0) return SVX_OK;
    if (!n_members || !d_n_members || !d_labels) return SVX_E_INVALID;
    uint64_t n_dist = 0, n_lab = 0, n_scratch = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const uint64_t n = n_members[p];
        n_dist += n * (n ? n - 1 : 0) / 2;
        n_lab += n;
        if (n > kLdsN) n_scratch += svx_align_up(svx_link_bytes((uint32_t)n), 16);
    }
    if (n_dist && !d_dist) return SVX_E_INVALID;
    if (n_lab == 0) return SVX_OK;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = svx_ws_reserve(ctx, 3 * svx_take_bytes(n_parts, 8) + svx_take_bytes(n_scratch ? n_scratch : 1, 1));
    if (rc != SVX_OK) return rc;
    LinkArgs a;
    uint64_t* d_doff = svx_ws_take<uint64_t>(ctx, n_parts);
    uint64_t* d_loff = svx_ws_take<uint64_t>(ctx, n_parts);
    uint64_t* d_soff = svx_ws_take<uint64_t>(ctx, n_parts);
    char* d_scratch = svx_ws_take<char>(ctx, n_scratch ? n_scratch : 1);
    hipLaunchKernelGGL(k_linkage_offsets, dim3(1), dim3(256), 0, ctx->stream, d_n_members, n_parts, d_doff, d_loff, d_soff);
    a.dist = d_dist; a.dist_off = d_doff; a.n_members = d_n_members; a.label_off = d_loff; a.scratch_off = d_soff;
    a.scratch = d_scratch; a.n_parts = n_parts; a.cutoff = cutoff; a.labels = d_labels;
    hipLaunchKernelGGL(k_linkage_cut, dim3((n_parts + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    return SVX_OK;
}

extern "C" int svx_linkage_cut_batch(svx_ctx* ctx, const double* dist, const uint32_t* n_members, uint32_t n_parts,
                                     double cutoff, uint32_t* labels) {
    if (!ctx) return SVX_E_INVALID;
    if (n_parts == 0) return SVX_OK;
    if (!n_members || !labels) return SVX_E_INVALID;
    std::vector<uint64_t> dist_off(n_parts), label_off(n_parts), scratch_off(n_parts);
    uint64_t n_dist = 0, n_lab = 0, n_scratch = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const uint64_t n = n_members[p];
        dist_off[p] = n_dist;
        label_off[p] = n_lab;
        scratch_off[p] = n_scratch;
        n_dist += n * (n ? n - 1 : 0) / 2;
        n_lab += n;
        if (n > kLdsN) n_scratch += svx_align_up(svx_link_bytes((uint32_t)n), 16);
    }
    if (n_dist && !dist) return SVX_E_INVALID;
    if (n_lab == 0) return SVX_OK;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = svx_take_bytes(n_dist ? n_dist : 1, 8) + 3 * svx_take_bytes(n_parts, 8) + svx_take_bytes(n_parts, 4) +
                  svx_take_bytes(n_lab, 4) + svx_take_bytes(n_scratch ? n_scratch : 1, 1);
    int rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    LinkArgs a;
    double* d_dist = svx_stage_take<double>(ctx, n_dist ? n_dist : 1);
    uint64_t* d_doff = svx_stage_take<uint64_t>(ctx, n_parts);
    uint64_t* d_loff = svx_stage_take<uint64_t>(ctx, n_parts);
    uint64_t* d_soff = svx_stage_take<uint64_t>(ctx, n_parts);
    uint32_t* d_nm = svx_stage_take<uint32_t>(ctx, n_parts);
    uint32_t* d_lab = svx_stage_take<uint32_t>(ctx, n_lab);
    char* d_scratch = svx_stage_take<char>(ctx, n_scratch ? n_scratch : 1);
    if (n_dist) SVX_HIP(ctx, hipMemcpyAsync(d_dist, dist, n_dist * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_doff, dist_off.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_loff, label_off.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_soff, scratch_off.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_nm, n_members, (size_t)n_parts * 4, hipMemcpyHostToDevice, ctx->stream));
    a.dist = d_dist; a.dist_off = d_doff; a.n_members = d_nm; a.label_off = d_loff; a.scratch_off = d_soff;
    a.scratch = d_scratch; a.n_parts = n_parts; a.cutoff = cutoff; a.labels = d_lab;
    rc = svx_timing_begin(ctx);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_mark(ctx, 1);
    if (rc != SVX_OK) return rc;
    hipLaunchKernelGGL(k_linkage_cut, dim3((n_parts + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream, a);
    SVX_HIP(ctx, hipGetLastError());
    rc = svx_timing_mark(ctx, 2);
    if (rc != SVX_OK) return rc;
    rc = svx_timing_end(ctx);
    if (rc != SVX_OK) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(labels, d_lab, n_lab * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
