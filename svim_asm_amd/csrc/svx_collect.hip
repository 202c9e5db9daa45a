// svx_collect.hip — svx_collect_batch: the device work of COLLECT for a whole sample in ONE submission
// (include/svx.h).  Replaces the arithmetic of analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83):
// per alignment analyze_alignment_indel (SVIM_intra.py:33-44) and, per chimeric read, analyze_read_segments
// (SVIM_inter.py:62-340), which the reference runs record by record in a Python loop.
//
// Stream order of one call:
//   H2D  every CIGAR pool as it lies in the reader's page-locked memory (no host-side concatenation), the CIGARs
//        of the SA-derived segments behind them, ONE packed control block (offsets, reference starts, segment
//        table, read offsets, contig ranks);
//   k_cigar_tiles + k_cigar_finish_small (or the five-launch streaming path)   a1 + a2, all records
//   k_segment_rows      CIGAR statistics + segment rows of the chimeric reads   (SVIM_inter.py:66-81)
//   k_segments          the adjacent-pair decision tree                         (:91-258)
//   k_segments_post     the three post-passes                                   (:260-338)
//   D2H  one packed block of counts (signature count, derived records per read), first synchronisation;
//   D2H  one packed block with exactly the signatures, raw records and derived-record regions, second one.
#include <algorithm>
#include <vector>

#include "svx_internal.h"

namespace {

struct Pack {  // layout of a packed block: sections aligned to 16 bytes
    size_t at = 0;
    size_t take(size_t bytes) {
        const size_t off = at;
        at = svx_align_up(at + bytes, 16);
        return off;
    }
};

int host_stage_reserve(svx_ctx* ctx, size_t bytes) {
    if (ctx->hpin_bytes >= bytes) return SVX_OK;
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    ctx->hpin = nullptr;
    ctx->hpin_bytes = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 1u << 20);
    void* p = nullptr;
    SVX_HIP(ctx, hipHostMalloc(&p, want, hipHostMallocDefault));
    ctx->hpin = static_cast<char*>(p);
    ctx->hpin_bytes = want;
    return SVX_OK;
}

}  // namespace

// The kernels of COLLECT on resident inputs, in stream order: a1 + a2, then the split-segment chain a3.
// (The two do not depend on each other; putting the chain on a second stream between a fork and a join event was
// measured and dropped: the cross-stream dependencies cost more than the overlap of five short launches gains —
// 45.6 vs 29.4 us per config-2 sample, profiles/README.md.)
extern "C" int svx_collect_batch_dev(svx_ctx* ctx, const svx_collect_dev* d) {
    if (!ctx || !d) return SVX_E_INVALID;
    if (d->n_aln == 0) {
        if (d->d_n_sig) SVX_HIP(ctx, hipMemsetAsync(d->d_n_sig, 0, 8, ctx->stream));
        return d->n_segs ? SVX_E_INVALID : SVX_OK;
    }
    if (!d->d_aln_off || !d->d_n_sig || (d->n_ops && !d->d_cigar)) return SVX_E_INVALID;
    const bool chain = d->n_reads != 0;
    if (chain && (!d->read_off || !d->d_read_off || !d->post_off || !d->d_post_off || !d->d_post_cnt || !d->d_segs ||
                  !d->d_read_len || !d->d_raw || !d->d_seg_src))
        return SVX_E_INVALID;
    if (!chain)
        return svx_cigar_extract_dev(ctx, d->d_cigar, d->n_ops, d->d_aln_off, d->n_aln, d->d_ref_start, d->min_len, d->d_sig,
                                     d->sig_cap, d->d_n_sig);
    if (!d->d_seg_tid || !d->d_seg_pos || !d->d_seg_rev || !d->d_seg_qend || (d->n_contigs && !d->d_contig_rank)) return SVX_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(d->d_segs) & 7u) || (reinterpret_cast<uintptr_t>(d->d_raw) & 15u)) {
        SVX_SET_ERR(ctx, "d_segs must be 8-byte aligned and d_raw 16-byte aligned");
        return SVX_E_INVALID;
    }
    if (d->read_off[d->n_reads] != d->n_segs) return SVX_E_INVALID;
    if (d->n_chain_blocks && (!d->d_chain_deal || d->n_chain_blocks > d->n_reads || (reinterpret_cast<uintptr_t>(d->d_chain_deal) & 7u)))
        return SVX_E_INVALID;
    uint64_t stride = 0;
    int rc = svx_postpass_plan(ctx, d->read_off, d->n_reads, d->post_off, &stride);
    if (rc != SVX_OK) return rc;
    if (d->post_off[d->n_reads] && !d->d_post) return SVX_E_INVALID;
    if (stride && !ctx->split_chain && d->n_ops != 0) {  // (records without a single op: nothing for the chain to ride in)
        // the chain goes out with the CIGAR path, inside two of its launches (svx_cigar.hip, a3_chain_block)
        svx_a3_plan q;
        q.d_seg_src = d->d_seg_src; q.d_seg_tid = d->d_seg_tid; q.d_seg_pos = d->d_seg_pos; q.d_seg_rev = d->d_seg_rev;
        q.d_seg_qend = d->d_seg_qend; q.n_segs = d->n_segs; q.d_read_off = d->d_read_off; q.n_reads = d->n_reads;
        q.d_contig_rank = d->d_contig_rank; q.n_contigs = d->n_contigs; q.params = d->params; q.d_segs = d->d_segs;
        q.d_read_len = d->d_read_len; q.d_raw = d->d_raw; q.d_post = d->d_post; q.d_post_off = d->d_post_off;
        q.d_post_cnt = d->d_post_cnt; q.post_stride = stride;
        q.d_deal = d->n_chain_blocks ? d->d_chain_deal : nullptr; q.n_deal_blocks = q.d_deal ? d->n_chain_blocks : 0;
        return svx_cigar_extract_chain_dev(ctx, d->d_cigar, d->n_ops, d->d_aln_off, d->n_aln, d->d_ref_start, d->min_len,
                                           d->d_sig, d->sig_cap, d->d_n_sig, &q);
    }
    // reads too uneven for one scratch slice size (or svx_ctx_set_split_chain): the single-purpose launches
    rc = svx_cigar_extract_dev(ctx, d->d_cigar, d->n_ops, d->d_aln_off, d->n_aln, d->d_ref_start, d->min_len, d->d_sig,
                               d->sig_cap, d->d_n_sig);
    if (rc != SVX_OK) return rc;
    rc = svx_segments_rows_dev(ctx, d->d_cigar, d->d_aln_off, d->d_seg_src, d->d_seg_tid, d->d_seg_pos, d->d_seg_rev,
                               d->d_seg_qend, d->n_segs, d->d_read_off, d->n_reads, d->d_segs, d->d_read_len);
    if (rc != SVX_OK) return rc;
    rc = svx_segments_classify_dev(ctx, d->d_segs, d->n_segs, d->d_read_off, d->n_reads, d->d_read_len, &d->params, d->d_raw);
    if (rc != SVX_OK) return rc;
    return svx_segments_postpass_dev(ctx, d->d_raw, d->read_off, d->d_read_off, d->n_reads, d->d_contig_rank, d->n_contigs,
                                     &d->params, d->d_post, d->post_off, d->d_post_off, d->d_post_cnt);
}

// Reads per workgroup of the chain's rows + tree stage for a submission of n_reads (0: no table — svx_cigar.hip deals
// one or two reads per workgroup itself): as many as keep the grid at about 2 000 workgroups — eight per CU, all
// resident at once —, at most 32.
static uint32_t chain_reads_per_block(uint32_t n_reads) {
    if (n_reads < 384u) return 0;
    const uint32_t per = n_reads / 1024u;
    return per < 2u ? 2u : (per > 32u ? 32u : per);
}

extern "C" int svx_chain_deal(const uint32_t* read_off, uint32_t n_reads, const uint32_t* seg_src, const uint64_t* aln_off,
                              uint32_t* deal) {
    if (!deal || (n_reads && (!read_off || !seg_src || !aln_off))) return SVX_E_INVALID;
    const uint32_t per = chain_reads_per_block(n_reads);
    if (per == 0) return 0;
    const uint32_t n_blocks = (n_reads + per - 1) / per;
    // cost of a read in ops: the chunks of its segments' CIGARs (svx_cigar.hip: kChunkOps = 128; alignments of at most
    // 8 ops are finished by one lane), and what the read costs whatever its size (its row stores, its share of the tree)
    constexpr uint64_t kChunk = 128, kPerRead = 96, kPerTiny = 8;
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (read_off[r + 1] < read_off[r]) return SVX_E_INVALID;
        total += kPerRead;
        for (uint32_t j = read_off[r]; j < read_off[r + 1]; ++j) {
            const uint64_t n = aln_off[(size_t)seg_src[j] + 1] - aln_off[seg_src[j]];
            total += n <= 8 ? kPerTiny : (n + kChunk - 1) / kChunk * kChunk;
        }
    }
    // read r goes to workgroup floor(prefix(r) * n_blocks / total): consecutive ranges; a read that costs more than a
    // workgroup's share leaves the workgroups it skips empty
    uint64_t prefix = 0;
    uint32_t next = 0;  // first workgroup without a start yet
    for (uint32_t r = 0; r < n_reads; ++r) {
        uint32_t b = (uint32_t)((unsigned __int128)prefix * n_blocks / total);
        if (b >= n_blocks) b = n_blocks - 1;
        for (; next <= b; ++next) { deal[2 * next] = r; deal[2 * next + 1] = read_off[r]; }
        prefix += kPerRead;
        for (uint32_t j = read_off[r]; j < read_off[r + 1]; ++j) {
            const uint64_t n = aln_off[(size_t)seg_src[j] + 1] - aln_off[seg_src[j]];
            prefix += n <= 8 ? kPerTiny : (n + kChunk - 1) / kChunk * kChunk;
        }
    }
    for (; next <= n_blocks; ++next) { deal[2 * next] = n_reads; deal[2 * next + 1] = read_off[n_reads]; }
    return (int)n_blocks;
}

extern "C" int svx_collect_batch(svx_ctx* ctx, const svx_collect_in* in, svx_collect_out* out) {
    if (!ctx || !in || !out) return SVX_E_INVALID;
    out->n_sig = 0;
    const uint32_t n_aln = in->n_aln, n_extra = in->n_extra, n_segs = in->n_segs, n_reads = in->n_reads;
    if (n_aln && (!in->aln_off || !in->ref_start)) return SVX_E_INVALID;
    if (in->n_parts && (!in->cigar_parts || !in->part_ops)) return SVX_E_INVALID;
    if (n_extra && !in->extra_off) return SVX_E_INVALID;
    if (n_segs && (!in->seg_src || !in->seg_tid || !in->seg_pos || !in->seg_rev || !in->seg_qend)) return SVX_E_INVALID;
    if (n_reads && (!in->read_off || !out->post_off || !out->post_cnt)) return SVX_E_INVALID;
    if (n_segs && !out->raw) return SVX_E_INVALID;
    if (in->n_contigs && !in->contig_rank) return SVX_E_INVALID;
    // ---- validate what the kernels will index with (host copies; nothing is trusted on the device)
    uint64_t n_ops = 0;
    for (uint32_t k = 0; k < in->n_parts; ++k) {
        if (in->part_ops[k] && !in->cigar_parts[k] && !(in->part_dev && in->part_dev[k])) return SVX_E_INVALID;
        n_ops += in->part_ops[k];
    }
    if (n_aln) {
        if (in->aln_off[0] != 0 || in->aln_off[n_aln] != n_ops) {
            SVX_SET_ERR(ctx, "aln_off must start at 0 and end at the total op count of the pools");
            return SVX_E_INVALID;
        }
        for (uint32_t i = 0; i < n_aln; ++i)
            if (in->aln_off[i + 1] < in->aln_off[i]) {
                SVX_SET_ERR(ctx, "aln_off must be non-decreasing (index %u)", i);
                return SVX_E_INVALID;
            }
    } else if (n_ops) {
        return SVX_E_INVALID;
    }
    uint64_t n_xops = 0;
    if (n_extra) {
        if (in->extra_off[0] != 0) return SVX_E_INVALID;
        for (uint32_t i = 0; i < n_extra; ++i)
            if (in->extra_off[i + 1] < in->extra_off[i]) return SVX_E_INVALID;
        n_xops = in->extra_off[n_extra];
        if (n_xops && !in->extra_cigar) return SVX_E_INVALID;
    }
    if (n_ops + n_xops >= (1ull << 32)) {
        SVX_SET_ERR(ctx, "%llu CIGAR ops exceed the 2^32-1 per-call limit; split the batch", (unsigned long long)(n_ops + n_xops));
        return SVX_E_TOO_LARGE;
    }
    for (uint32_t j = 0; j < n_segs; ++j)
        if (in->seg_src[j] >= n_aln + n_extra) {
            SVX_SET_ERR(ctx, "segment %u names alignment %u of %u", j, in->seg_src[j], n_aln + n_extra);
            return SVX_E_INVALID;
        }
    uint64_t n_post = 0;
    if (n_reads) {
        if (in->read_off[0] != 0 || in->read_off[n_reads] != n_segs) return SVX_E_INVALID;
        for (uint32_t r = 0; r < n_reads; ++r) {
            if (in->read_off[r + 1] < in->read_off[r] || out->post_off[r + 1] < out->post_off[r]) return SVX_E_INVALID;
            if (out->post_off[r + 1] - out->post_off[r] < svx_segments_postpass_bound(in->read_off[r + 1] - in->read_off[r])) {
                SVX_SET_ERR(ctx, "read %u: too few output slots (svx_segments_postpass_bound)", r);
                return SVX_E_CAPACITY;
            }
        }
        n_post = out->post_off[n_reads];
        if (n_post && !out->post) return SVX_E_INVALID;
    } else if (n_segs) {
        return SVX_E_INVALID;
    }
    const uint64_t cap = out->sig_cap;
    if (cap && (!out->sig.aln || !out->sig.ref_pos || !out->sig.read_pos || !out->sig.len || !out->sig.type)) return SVX_E_INVALID;
    if (n_aln == 0) {
        if (n_reads) memset(out->post_cnt, 0, (size_t)n_reads * 4);
        return n_segs ? SVX_E_INVALID : SVX_OK;
    }
    SVX_HIP(ctx, hipSetDevice(ctx->device));

    // ---- control block (host, page-locked): built here, uploaded with one copy
    const uint32_t n_all = n_aln + n_extra;
    Pack cb;
    const size_t o_off = cb.take(((size_t)n_all + 1) * 8), o_poff = cb.take(((size_t)n_reads + 1) * 8);
    const size_t o_rs = cb.take((size_t)n_aln * 4), o_src = cb.take((size_t)n_segs * 4), o_tid = cb.take((size_t)n_segs * 4);
    const size_t o_pos = cb.take((size_t)n_segs * 4), o_qe = cb.take((size_t)n_segs * 4);
    const size_t o_roff = cb.take(((size_t)n_reads + 1) * 4), o_rank = cb.take((size_t)in->n_contigs * 4);
    const size_t o_rev = cb.take(n_segs);
    const size_t o_deal = cb.take(chain_reads_per_block(n_reads) ? ((size_t)n_reads + 2) * 8 : 0);
    const size_t cb_bytes = cb.at;
    // ---- results block on the device; its head (counts) and its body are read back separately
    Pack rb;
    const size_t r_n = rb.take(8), r_cnt = rb.take((size_t)n_reads * 4);
    const size_t head_bytes = rb.at;
    const size_t r_aln = rb.take(cap * 4), r_ref = rb.take(cap * 4), r_read = rb.take(cap * 4), r_len = rb.take(cap * 4);
    const size_t r_type = rb.take(cap), r_raw = rb.take((size_t)n_segs * sizeof(svx_raw)), r_post = rb.take(n_post * sizeof(svx_post));
    const size_t rb_bytes = rb.at;
    int rc = host_stage_reserve(ctx, std::max(cb_bytes, rb_bytes));
    if (rc != SVX_OK) return rc;
    const size_t need = svx_take_bytes(n_ops + n_xops + 4, 4) + svx_take_bytes(cb_bytes, 1) + svx_take_bytes(rb_bytes, 1) +
                        svx_take_bytes(n_segs ? n_segs : 1, sizeof(svx_seg)) + svx_take_bytes(n_reads ? n_reads : 1, 4);
    rc = svx_stage_reserve(ctx, need);
    if (rc != SVX_OK) return rc;
    uint32_t* d_cigar = svx_stage_take<uint32_t>(ctx, n_ops + n_xops + 4);
    char* d_cb = svx_stage_take<char>(ctx, cb_bytes);
    char* d_rb = svx_stage_take<char>(ctx, rb_bytes);
    svx_seg* d_segs = svx_stage_take<svx_seg>(ctx, n_segs ? n_segs : 1);
    int32_t* d_read_len = svx_stage_take<int32_t>(ctx, n_reads ? n_reads : 1);

    char* h = ctx->hpin;
    uint64_t* h_off = reinterpret_cast<uint64_t*>(h + o_off);
    memcpy(h_off, in->aln_off, ((size_t)n_aln + 1) * 8);
    for (uint32_t i = 0; i < n_extra; ++i) h_off[n_aln + 1 + i] = n_ops + in->extra_off[i + 1];
    if (n_reads) memcpy(h + o_poff, out->post_off, ((size_t)n_reads + 1) * 8);
    memcpy(h + o_rs, in->ref_start, (size_t)n_aln * 4);
    if (n_segs) {
        memcpy(h + o_src, in->seg_src, (size_t)n_segs * 4);
        memcpy(h + o_tid, in->seg_tid, (size_t)n_segs * 4);
        memcpy(h + o_pos, in->seg_pos, (size_t)n_segs * 4);
        memcpy(h + o_qe, in->seg_qend, (size_t)n_segs * 4);
        memcpy(h + o_rev, in->seg_rev, n_segs);
    }
    if (n_reads) memcpy(h + o_roff, in->read_off, ((size_t)n_reads + 1) * 4);
    if (in->n_contigs) memcpy(h + o_rank, in->contig_rank, (size_t)in->n_contigs * 4);
    int n_deal = 0;
    if (chain_reads_per_block(n_reads)) {
        n_deal = svx_chain_deal(in->read_off, n_reads, in->seg_src, h_off, reinterpret_cast<uint32_t*>(h + o_deal));
        if (n_deal < 0) return n_deal;
    }

    // ---- uploads: the pools where they lie (page-locked reader memory: true DMA), extra CIGARs, control block
    uint64_t at = 0;
    for (uint32_t k = 0; k < in->n_parts; ++k) {
        if (in->part_ops[k] && in->part_dev && in->part_dev[k]) {  // the reader's copy in HBM (svx_bam_device_pool)
            if (in->part_ready && in->part_ready[k])
                SVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, static_cast<hipEvent_t>(in->part_ready[k]), 0));
            SVX_HIP(ctx, hipMemcpyAsync(d_cigar + at, in->part_dev[k], (size_t)in->part_ops[k] * 4, hipMemcpyDeviceToDevice, ctx->stream));
        } else if (in->part_ops[k]) {
            SVX_HIP(ctx, hipMemcpyAsync(d_cigar + at, in->cigar_parts[k], (size_t)in->part_ops[k] * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        at += in->part_ops[k];
    }
    if (n_xops) SVX_HIP(ctx, hipMemcpyAsync(d_cigar + n_ops, in->extra_cigar, (size_t)n_xops * 4, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(d_cb, h, cb_bytes, hipMemcpyHostToDevice, ctx->stream));

    // ---- kernels (svx_collect_batch_dev)
    svx_collect_dev dv;
    memset(&dv, 0, sizeof(dv));
    dv.d_cigar = d_cigar; dv.n_ops = n_ops;
    dv.d_aln_off = reinterpret_cast<const uint64_t*>(d_cb + o_off); dv.n_aln = n_aln; dv.n_extra = n_extra;
    dv.d_ref_start = reinterpret_cast<const int32_t*>(d_cb + o_rs); dv.min_len = in->min_len;
    dv.d_seg_src = reinterpret_cast<const uint32_t*>(d_cb + o_src); dv.d_seg_tid = reinterpret_cast<const int32_t*>(d_cb + o_tid);
    dv.d_seg_pos = reinterpret_cast<const int32_t*>(d_cb + o_pos); dv.d_seg_rev = reinterpret_cast<const uint8_t*>(d_cb + o_rev);
    dv.d_seg_qend = reinterpret_cast<const int32_t*>(d_cb + o_qe); dv.n_segs = n_segs;
    dv.read_off = in->read_off; dv.d_read_off = reinterpret_cast<const uint32_t*>(d_cb + o_roff); dv.n_reads = n_reads;
    dv.d_contig_rank = reinterpret_cast<const int32_t*>(d_cb + o_rank); dv.n_contigs = in->n_contigs; dv.params = in->params;
    dv.d_sig.aln = reinterpret_cast<uint32_t*>(d_rb + r_aln); dv.d_sig.ref_pos = reinterpret_cast<uint32_t*>(d_rb + r_ref);
    dv.d_sig.read_pos = reinterpret_cast<uint32_t*>(d_rb + r_read); dv.d_sig.len = reinterpret_cast<uint32_t*>(d_rb + r_len);
    dv.d_sig.type = reinterpret_cast<uint8_t*>(d_rb + r_type); dv.sig_cap = cap; dv.d_n_sig = reinterpret_cast<uint64_t*>(d_rb + r_n);
    dv.d_segs = d_segs; dv.d_read_len = d_read_len; dv.d_raw = reinterpret_cast<svx_raw*>(d_rb + r_raw);
    dv.d_post = reinterpret_cast<svx_post*>(d_rb + r_post); dv.post_off = out->post_off;
    dv.d_post_off = reinterpret_cast<const uint64_t*>(d_cb + o_poff); dv.d_post_cnt = reinterpret_cast<uint32_t*>(d_rb + r_cnt);
    if (n_deal > 0) { dv.d_chain_deal = reinterpret_cast<const uint32_t*>(d_cb + o_deal); dv.n_chain_blocks = (uint32_t)n_deal; }
    rc = svx_collect_batch_dev(ctx, &dv);
    if (rc != SVX_OK) return rc;

    // ---- read-back 1: the counts
    SVX_HIP(ctx, hipMemcpyAsync(h, d_rb, head_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = svx_barrier_check(ctx);
    if (rc != SVX_OK) return rc;
    const uint64_t n_sig = *reinterpret_cast<const uint64_t*>(h + r_n);
    out->n_sig = n_sig;
    if (n_reads) memcpy(out->post_cnt, h + r_cnt, (size_t)n_reads * 4);
    if (n_sig > cap) {
        SVX_SET_ERR(ctx, "output capacity %llu < %llu signatures", (unsigned long long)cap, (unsigned long long)n_sig);
        return SVX_E_CAPACITY;
    }
    // ---- read-back 2: exactly what was produced, as one copy of [first signature column .. end of the records]
    // (the columns sit at capacity strides; the gaps travel along when the capacity is generous, so the caller
    // sizes it from the op count)
    if (n_sig || n_segs || n_post) {
        SVX_HIP(ctx, hipMemcpyAsync(h + r_aln, d_rb + r_aln, rb_bytes - r_aln, hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(out->sig.aln, h + r_aln, n_sig * 4);
        memcpy(out->sig.ref_pos, h + r_ref, n_sig * 4);
        memcpy(out->sig.read_pos, h + r_read, n_sig * 4);
        memcpy(out->sig.len, h + r_len, n_sig * 4);
        memcpy(out->sig.type, h + r_type, n_sig);
        if (n_segs) memcpy(out->raw, h + r_raw, (size_t)n_segs * sizeof(svx_raw));
        if (n_post) memcpy(out->post, h + r_post, n_post * sizeof(svx_post));
    }
    return SVX_OK;
}
