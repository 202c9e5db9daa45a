// svx_internal.h — context, workspace and error plumbing shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "svx.h"

struct svx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // growable HBM workspace (bump-allocated per call, reused across calls)
    char* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_used = 0;
    // second, independent region for host-pointer entry points (staged inputs/outputs)
    char* stage = nullptr;
    size_t stage_bytes = 0;
    size_t stage_used = 0;
    // page-locked host block of svx_collect_batch (control block up, packed results down)
    char* hpin = nullptr;
    size_t hpin_bytes = 0;
    // timing
    bool timing = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    // recorded right after the streaming kernel of every cigar call (svx_ctx_wait_dominant)
    hipEvent_t ev_block = nullptr;  // svx_wait_blocking
    hipEvent_t ev_dom = nullptr;
    bool ev_dom_recorded = false;
    bool want_dom = false;
    int n_cu = 256;
    uint32_t wfa_cap = 1024;  // svx_ctx_set_edit_wavefront_cap: edits the wavefront pass of the edit distance resolves
    bool barrier_pending = false;      // a kernel with a grid barrier went out since the last svx_barrier_check
    bool barrier_timed_out = false;    // the last failed svx_ctx_sync was a wait between workgroups that ran out
    bool pair_lds_set = false;         // k_pair_single's dynamic-LDS limit raised on this context's device
    uint32_t pair_launches = 0;        // launches of k_pair_single so far: which set of arrival counters is next
    bool pair_wait_free = false;       // svx_ctx_set_pair_wait_free: sort on the plan without waits inside a launch
    uint32_t pair_retries = 0;         // host-pointer calls that were re-run on that plan after a wait ran out
    uint32_t pair_single_max = 131072;  // svx_ctx_set_pair_single_launch_max: largest batch of the one-launch pair sort
    uint64_t small_batch_ops = 1ull << 23;  // svx_ctx_set_small_batch_ops: largest batch of the two-launch CIGAR path
    bool split_chain = false;               // svx_ctx_set_split_chain: the split-segment chain as three launches
    char err[512] = {0};
};

#define SVX_SET_ERR(ctx, ...)                                   \
    do {                                                        \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
    } while (0)

#define SVX_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            SVX_SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                        __FILE__, __LINE__);                                            \
            return (e__ == hipErrorOutOfMemory) ? SVX_E_NOMEM : SVX_E_HIP;              \
        }                                                                               \
    } while (0)

static inline size_t svx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Reserve `total` bytes of workspace for this call (may reallocate; synchronises the
// stream first when it has to, so earlier kernels never lose their scratch).
int svx_ws_reserve(svx_ctx* ctx, size_t total);
// Word of the workspace header where a kernel notes that a wait for other workgroups ran out; to be called
// with the stream synchronised: SVX_OK, or SVX_E_HIP after clearing the header for the calls to come.
#define SVX_WS_BARRIER_NOTE_WORD 80
int svx_barrier_check(svx_ctx* ctx);
int svx_stage_reserve(svx_ctx* ctx, size_t total);
// workspace bytes svx_cigar_extract*_dev / svx_segments_postpass_dev reserve for a batch (svx_collect_batch_dev sums them)
size_t svx_cigar_extract_ws_need(const svx_ctx* ctx, uint64_t n_ops);
size_t svx_postpass_ws_need(const uint32_t* read_off, uint32_t n_reads);

// The split-segment chain of one submission (the a3 fields of svx_collect_dev) with the per-read scratch slice
// size of its post-passes (svx_postpass_plan); svx_cigar_extract_chain_dev sends it out with the CIGAR path.
struct svx_a3_plan {
    const uint32_t* d_seg_src;
    const int32_t* d_seg_tid;
    const int32_t* d_seg_pos;
    const uint8_t* d_seg_rev;
    const int32_t* d_seg_qend;
    uint32_t n_segs;
    const uint32_t* d_read_off;
    uint32_t n_reads;
    const int32_t* d_contig_rank;
    uint32_t n_contigs;
    svx_seg_params params;
    svx_seg* d_segs;
    int32_t* d_read_len;
    svx_raw* d_raw;
    svx_post* d_post;
    const uint64_t* d_post_off;
    uint32_t* d_post_cnt;
    uint64_t post_stride;
    const uint32_t* d_deal;    // svx_collect_dev.d_chain_deal (nullable)
    uint32_t n_deal_blocks;
};
// Validates the host copies of read_off / out_off like svx_segments_postpass_dev and returns the uniform scratch
// slice size of the post-passes (0: the reads are too uneven for one size — take the separate launches).
int svx_postpass_plan(svx_ctx* ctx, const uint32_t* read_off, uint32_t n_reads, const uint64_t* out_off, uint64_t* stride);
int svx_cigar_extract_chain_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                                const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out, uint64_t cap,
                                uint64_t* d_n_out, const svx_a3_plan* a3);

template <typename T>
static inline T* svx_ws_take(svx_ctx* ctx, size_t count) {
    size_t off = svx_align_up(ctx->ws_used, 256);
    ctx->ws_used = off + count * sizeof(T);
    return reinterpret_cast<T*>(ctx->ws + off);
}
template <typename T>
static inline T* svx_stage_take(svx_ctx* ctx, size_t count) {
    size_t off = svx_align_up(ctx->stage_used, 256);
    ctx->stage_used = off + count * sizeof(T);
    return reinterpret_cast<T*>(ctx->stage + off);
}
static inline size_t svx_take_bytes(size_t count, size_t elem) {
    return svx_align_up(count * elem, 256) + 256;
}

// Wait for everything enqueued on the context's stream WITHOUT spinning: an event created with hipEventBlockingSync
// puts the thread to sleep until the device signals (hipStreamSynchronize burns a CPU for the whole wait: under a
// CPU quota, milliseconds of kernel time then cost the host's other threads their share).
// svx_inflate.hip, for the BAM reader's device leg (svx_bam.cpp): hipError_t as int
// (d_tok: room for the token lists of `tok_members` members, SVX_INFLATE_TOK_STRIDE slots of 8 bytes each — the members go
//  out that many at a time; d_n_tok: a word per member; both null: the one-launch kernel)
int svx_bgzf_inflate_on_stream(void* stream, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                               const uint64_t* d_out_off, uint32_t* d_status, uint32_t* d_n_tok, void* d_tok, uint32_t tok_members);
int svx_gather_ranges_on_stream(void* stream, const uint8_t* d_src, const uint64_t* d_src_off, const uint32_t* d_len,
                                const uint64_t* d_dst_off, uint32_t n, uint8_t* d_dst);
int svx_wait_blocking(svx_ctx* ctx);
int svx_timing_begin(svx_ctx* ctx);           // records ev[0]
int svx_timing_mark(svx_ctx* ctx, int which); // records ev[which] (1: dominant start, 2: dominant end)
int svx_timing_end(svx_ctx* ctx);             // records ev[3]
