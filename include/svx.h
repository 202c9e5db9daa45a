/*
 * svx.h — C-ABI of libsvx.so: the MI355X (gfx950) hot path of SVIM-asm.
 *
 * The reference (eldariont/svim-asm v1.0.3) is pure Python and has no FFI layer;
 * the boundary a maintainer would bind is the set of Python function seams listed
 * in SURVEY.md §8(b).  Each entry point below names the reference function
 * (file:line under /root/reference/src/svim_asm) whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes stub that binds them from the reference side.
 *
 * Conventions
 *   - every function returns int: SVX_OK (0) or a negative svx_status; nothing
 *     throws, aborts or keeps hidden global state;
 *   - a context (svx_ctx) = one device + one HIP stream + a growable HBM
 *     workspace.  One thread uses a context at a time; contexts are independent
 *     (8 contexts for the 8 GPUs of a node);
 *   - functions without a suffix take HOST pointers (the library stages them to
 *     HBM, runs the kernels and copies the results back, then synchronises);
 *     functions ending in _dev take DEVICE pointers, enqueue on the context's
 *     stream and do not synchronise (the caller owns the buffers, e.g. torch
 *     tensors, and synchronises the stream it handed in);
 *   - all outputs are written in the reference's deterministic order (prefix-sum
 *     compaction and stable sorts; no atomically-ordered appends).
 *   - there is NO CPU fallback inside this library: without a usable HIP device
 *     svx_ctx_create fails with SVX_E_NODEVICE.
 */
#ifndef SVX_H_
#define SVX_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svx_ctx svx_ctx;

typedef enum svx_status {
    SVX_OK = 0,
    SVX_E_INVALID = -1,   /* bad argument (null pointer, non-monotone offsets, ...) */
    SVX_E_CAPACITY = -2,  /* output buffer too small; *n_out holds the needed count */
    SVX_E_HIP = -3,       /* a HIP runtime call failed; see svx_last_error          */
    SVX_E_NOMEM = -4,     /* HBM / host allocation failed                           */
    SVX_E_TOO_LARGE = -5, /* batch exceeds a documented limit (e.g. >= 2^32 ops)    */
    SVX_E_NODEVICE = -6   /* no HIP device / device index out of range              */
} svx_status;

/* ---------------------------------------------------------------- context -- */

/* Create a context on HIP device `device` with its own non-blocking stream. */
int svx_ctx_create(int device, svx_ctx** out);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current stream:
 * torch.cuda.current_stream().cuda_stream).  Pass NULL for the default stream. */
int svx_ctx_create_on_stream(int device, void* hip_stream, svx_ctx** out);
void svx_ctx_destroy(svx_ctx* ctx);
/* Block until everything enqueued on the context's stream has finished. */
int svx_ctx_sync(svx_ctx* ctx);
/* Last error text for this context (never NULL; "" when no error). */
const char* svx_last_error(const svx_ctx* ctx);
const char* svx_version(void);
/* Number of visible HIP devices (0 when none / no driver). Never fails. */
int svx_device_count(void);
/* PCI address of a visible device as "domain:bus:device.function" (hipDeviceGetPCIBusId): what a host process needs to
 * find the device's NUMA node (/sys/bus/pci/devices/<address>/numa_node) and keep its threads there — one worker
 * process per GPU of a node (svim-asm-cohort --device k).  `len` >= 16. */
int svx_device_pci_bus_id(int device, char* out, int len);

/* svx_cigar_extract* run batches of at most `max_ops` CIGAR ops in TWO kernel launches (tiles of 1024
 * ops; every workgroup of the second kernel scans all tile descriptors itself): the operating point of
 * the svim-asm CLI — one BAM of an assembly (~1.5 M ops) or both haplotype BAMs of a diploid sample
 * (~6 M ops) per call —, where five dependent launches cost more than the work.  Larger batches take the
 * streaming path (tiles of 4096 ops, five launches).  Default and upper limit 2^23 ops; 0 disables the
 * small-batch path; larger values are clamped to 2^23.  Results are identical on both. */
int svx_ctx_set_small_batch_ops(svx_ctx* ctx, uint64_t max_ops);
/* svx_collect_batch* send the split-segment chain of a submission (segment rows -> decision tree -> post-passes,
 * SVIM_inter.py:62-340) out INSIDE the launches of the CIGAR path: rows and decision tree among the workgroups of the
 * tile launch (two-launch path) or of the finish launch (streaming path), the post-passes among the workgroups of
 * the path's last launch.  on != 0 restores the single-purpose launches (svx_segments_rows_dev,
 * svx_segments_classify_dev, svx_segments_postpass_dev behind svx_cigar_extract_dev) — same results; kept for
 * comparison and for submissions whose reads are too uneven for one scratch slice size, which take it by
 * themselves.  Default off. */
int svx_ctx_set_split_chain(svx_ctx* ctx, int on);
/* svx_pair_partition* sort batches of at most `max_candidates` keys in ONE kernel launch (buckets by the
 * leading key bits, every window of buckets sorted inside one workgroup's LDS — merged when the window
 * arrives as a few sorted runs, as PAIR's two haplotype lists do, counted otherwise —, two arrival barriers
 * inside the launch): the PAIR step of one diploid sample is 60-90 k candidates.  Larger batches take the
 * radix path (P + 2 launches).  Default and upper limit 131072; 0 disables the one-launch path.  Results
 * are identical on both.  The launch holds at most min(64, CUs / 4) workgroups, all resident: up to four
 * contexts may run it on one device at the same time.  Should its workgroups (or those of the radix path's
 * partition sweep) ever wait for each other longer than 20 s, they give up: svx_pair_partition re-runs the call on
 * the wait-free plan (below) and succeeds; for the asynchronous entry points the next svx_ctx_sync returns
 * SVX_E_HIP, the outputs of that call are invalid, the context stays usable. */
int svx_ctx_set_pair_single_launch_max(svx_ctx* ctx, uint32_t max_candidates);
/* For callers of the asynchronous svx_pair_partition_dev*: 1 when the SVX_E_HIP that the latest svx_ctx_sync (or a
 * later call on the context) returned was a wait between workgroups that ran out — the outputs of that call are
 * invalid, the context is usable, and the same call enqueued again after svx_ctx_set_pair_wait_free(ctx, 1) cannot
 * fail that way (what svx_pair_partition does by itself for host-pointer callers).  Reading clears the flag. */
int svx_ctx_barrier_timed_out(svx_ctx* ctx);
/* The plan that never waits between workgroups inside a launch: radix passes (P + 1 launches) and the partition
 * sweep as two launches.  Slower than the plans above and independent of who else is resident on the device.
 * svx_pair_partition (host pointers, synchronous) falls back to it by itself when a wait of the faster plans
 * runs out — the call then succeeds, svx_ctx_pair_retries counts such calls —; callers of the asynchronous *_dev
 * entry points, which cannot be re-run behind their back, may select it up front.  Results are identical. */
int svx_ctx_set_pair_wait_free(svx_ctx* ctx, int enabled);
int svx_ctx_pair_retries(const svx_ctx* ctx);

/* Device buffers for callers that have no other owner of HBM (a ctypes binding without torch, the
 * tests): plain hipMalloc / hipFree / hipMemcpyAsync on the context's device and stream.
 * svx_dev_upload and svx_dev_download are ordered on the context's stream with the kernels of the
 * *_dev entry points; svx_dev_download synchronises the stream before it returns. */
int svx_dev_malloc(svx_ctx* ctx, size_t bytes, void** d_out);
int svx_dev_free(svx_ctx* ctx, void* d_ptr);
int svx_dev_upload(svx_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int svx_dev_download(svx_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* Kernel timing with HIP events on the context's stream.  When enabled, every
 * *_dev entry point brackets its kernels with events; after svx_ctx_sync the
 * elapsed GPU time of the last call is returned in milliseconds. */
int svx_ctx_set_timing(svx_ctx* ctx, int enabled);
int svx_ctx_last_kernel_ms(svx_ctx* ctx, float* ms_total, float* ms_dominant);
/* Measurement aid: the read-only streaming rate this device sustains right now — `reps` passes of a kernel that
 * reads d_buf[0 .. bytes) once per pass with the loads of the CIGAR stream (16 bytes per lane, four in flight,
 * nontemporal) and writes nothing; *ms_per_pass = HIP-event time / reps.  bytes / that time is the second
 * denominator a read-mostly kernel's achieved GB/s is quoted against (a device-to-device copy is not a
 * ceiling for such a kernel).  Synchronises the stream.  d_buf 16-byte aligned. */
int svx_hbm_read_probe_dev(svx_ctx* ctx, const void* d_buf, size_t bytes, uint32_t reps, float* ms_per_pass);

/* Pipelining independent batches over two contexts: make everything enqueued on `ctx` from now on
 * wait until the streaming (dominant) kernel of `other`'s most recent svx_cigar_extract*_dev call
 * has finished — the scan/finish tail of that call may still overlap.  The first call only arms
 * `other` (it starts recording an event after its streaming kernel); no-op until one was recorded.
 * Both contexts must live on the same device. */
int svx_ctx_wait_dominant(svx_ctx* ctx, svx_ctx* other);

/* ------------------------------------------------------------ a1 + a2 ------ */
/*
 * CIGAR walk → indel signatures.
 * Replaces analyze_cigar_indel (SVIM_intra.py:8-30) over a whole batch of
 * alignments, fused with the `ref_start + pos_ref` of analyze_alignment_indel
 * (SVIM_intra.py:36-43).  The batch boundary is the per-contig loop of
 * analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83).
 *
 *   cigar      BAM-native packed ops, `len << 4 | op`, all alignments back to back
 *              (pysam cigartuples (op,len) == (w & 15, w >> 4)); n_ops = aln_off[n_aln]
 *   aln_off    n_aln + 1 non-decreasing offsets into `cigar` (empty alignments allowed)
 *   ref_start  per-alignment reference_start added to pos_ref, or NULL for the
 *              raw analyze_cigar_indel result (pos_ref relative to the alignment)
 *   min_len    options.min_sv_size; an I/D op emits when len >= min_len
 *
 * Output (SoA, 17 B per signature, in (alignment, op) order):
 *   aln        index of the alignment the op belongs to
 *   ref_pos    ref_start[aln] + running reference offset BEFORE the op
 *   read_pos   running query offset BEFORE the op (counts leading soft clips)
 *   len        op length
 *   type       SVX_SIG_INS (op 1) or SVX_SIG_DEL (op 2)
 *
 * Ops 0(M),7(=),8(X) advance both cursors, 1(I),4(S) the query, 2(D) the
 * reference; 3(N),5(H),6(P),9(B) and codes 10-15 advance nothing
 * (SVIM_intra.py:14-29 has no branch for them).
 *
 * If more than `cap` signatures exist, nothing beyond `cap` is written, *n_out
 * receives the exact count and SVX_E_CAPACITY is returned.
 * Limits: n_ops < 2^32 per call; per-alignment cursor sums < 2^32.
 */
#define SVX_SIG_INS 0
#define SVX_SIG_DEL 1

typedef struct svx_sig_soa {
    uint32_t* aln;
    uint32_t* ref_pos;
    uint32_t* read_pos;
    uint32_t* len;
    uint8_t* type;
} svx_sig_soa;

int svx_cigar_extract(svx_ctx* ctx, const uint32_t* cigar, const uint64_t* aln_off,
                      uint32_t n_aln, const int32_t* ref_start, uint32_t min_len,
                      svx_sig_soa out, uint64_t cap, uint64_t* n_out);

/* SoA input variant named by the north star: op codes and lengths in two arrays
 * (pysam cigartuples flattened as u8 op[], u32 len[]). Same output contract.
 * The PACKED layout above is the fast one: it is what a BAM record holds (no conversion on ingest) and
 * it moves 4 B per op instead of 5.  Both kernels stream at the same HBM rate (≈ 6 TB/s on MI355X:
 * 331 µs for the 1.98 GB of a 395 M-op SoA batch, 265 µs for the 1.58 GB of the same batch packed), so
 * in ops/s — and in the 4 B/op the roofline credits — SoA runs at 4/5 of packed (0.62 vs 0.78 of
 * peak).  Callers that hold packed words should pass them as they are. */
int svx_cigar_extract_soa(svx_ctx* ctx, const uint8_t* op, const uint32_t* len,
                          const uint64_t* aln_off, uint32_t n_aln, const int32_t* ref_start,
                          uint32_t min_len, svx_sig_soa out, uint64_t cap, uint64_t* n_out);

/* Device-pointer variants.  PRECONDITION (not checked: the offsets live in HBM and nothing is read
 * back): d_aln_off[0] == 0, non-decreasing, d_aln_off[n_aln] == n_ops — the host-pointer entry points
 * validate exactly this and return SVX_E_INVALID; with offsets that violate it the behaviour is
 * undefined (alignment indices derived from them index d_ref_start).  A caller that cannot vouch
 * for its offsets should validate them before the upload, as svx_cigar_extract does.
 * n_ops must equal aln_off[n_aln] (the caller knows it;
 * the device copy is not read back).  d_cigar / d_len / d_op must be 16-byte aligned;
 * d_op is fetched as dwords, i.e. it must be readable up to the next multiple of 4 bytes
 * after n_ops (any hipMalloc'ed buffer is); the extra bytes are ignored.  d_n_out: one uint64 in device memory.
 * Exactly min(count, cap) signatures are written. Asynchronous on the ctx stream. */
int svx_cigar_extract_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops,
                          const uint64_t* d_aln_off, uint32_t n_aln, const int32_t* d_ref_start,
                          uint32_t min_len, svx_sig_soa d_out, uint64_t cap, uint64_t* d_n_out);
int svx_cigar_extract_soa_dev(svx_ctx* ctx, const uint8_t* d_op, const uint32_t* d_len,
                              uint64_t n_ops, const uint64_t* d_aln_off, uint32_t n_aln,
                              const int32_t* d_ref_start, uint32_t min_len, svx_sig_soa d_out,
                              uint64_t cap, uint64_t* d_n_out);

/* Per-alignment CIGAR statistics in the same pass family (pysam/htslib
 * definitions used by SVIM_inter.py:68-79 and SVIM_COLLECT.py:11):
 *   ref_len    Σ len over {M,D,N,=,X}   (reference_end - reference_start, htslib bam_endpos)
 *   q_start    Σ leading S ops (skipping H)        = query_alignment_start
 *   q_end      q_start + Σ len over {M,I,=,X}      = query_alignment_end
 *   read_len   Σ len over {M,I,S,=,X,H}            = infer_read_length()
 *   n_hard     Σ len over H                        = get_cigar_stats()[0][5]
 * Each output array has n_aln entries (any may be NULL to skip). Host pointers. */
typedef struct svx_aln_stats {
    uint32_t* ref_len;
    uint32_t* q_start;
    uint32_t* q_end;
    uint32_t* read_len;
    uint32_t* n_hard;
} svx_aln_stats;

int svx_cigar_stats(svx_ctx* ctx, const uint32_t* cigar, const uint64_t* aln_off, uint32_t n_aln,
                    svx_aln_stats out);
int svx_cigar_stats_dev(svx_ctx* ctx, const uint32_t* d_cigar, uint64_t n_ops,
                        const uint64_t* d_aln_off, uint32_t n_aln, svx_aln_stats d_out);

/* ---------------------------------------------------------------- a3 ------- */
/*
 * Split-segment classification: the adjacent-pair decision tree of
 * analyze_read_segments (SVIM_inter.py:62-258) for a batch of reads.
 *
 * For read r the caller passes its 1 + k segments in the reference's order
 * ([primary] + supplementaries, SVIM_inter.py:64) at segs[read_off[r] ..
 * read_off[r+1]).  q_start/q_end are the values of SVIM_inter.py:68-73 (already
 * flipped for reverse-strand records).  The kernel sorts each read's segments
 * stably by (q_start, q_end) (SVIM_inter.py:83) and classifies each adjacent
 * pair; out[read_off[r] + i] describes sorted pair (i, i+1); the last slot of
 * each read is SVX_RAW_NONE.  read_len[r] = primary.infer_read_length()
 * (SVIM_inter.py:120).
 *
 * Raw records (a0..a5):
 *   SVX_RAW_INS    ref_id, start, end, seq_start, seq_len  (slice of primary.query_sequence)
 *   SVX_RAW_DEL    ref_id, start, end
 *   SVX_RAW_BND    ref_id1, pos1, dir1, ref_id2, pos2, dir2   (dir: 0 'fwd', 1 'rev');
 *                  every BND is also a `translocations` tuple (SVIM_inter.py:136...)
 *   SVX_RAW_TANDEM ref_id, start, end, fully_covered, direction_fwd   (:150-164)
 *   SVX_RAW_INV    ref_id, start, end, side  (0 left_fwd, 1 left_rev, 2 right_fwd, 3 right_rev)
 * The post-passes (tandem merge :261-290, interspersed duplications :293-320,
 * inversion clustering :323-338) consume these records on the host.
 */
#define SVX_RAW_NONE 0
#define SVX_RAW_INS 1
#define SVX_RAW_DEL 2
#define SVX_RAW_BND 3
#define SVX_RAW_TANDEM 4
#define SVX_RAW_INV 5

typedef struct svx_seg {
    int32_t q_start, q_end, ref_id, ref_start, ref_end, is_reverse;
} svx_seg;

typedef struct svx_seg_params {
    int32_t min_sv_size;
    int32_t max_sv_size;
    int32_t query_gap_tolerance;
    int32_t query_overlap_tolerance;
    int32_t reference_gap_tolerance;
    int32_t reference_overlap_tolerance;
} svx_seg_params;

typedef struct svx_raw {
    int32_t kind;
    int32_t a0, a1, a2, a3, a4, a5;
    int32_t pad;
} svx_raw;

int svx_segments_classify(svx_ctx* ctx, const svx_seg* segs, const uint32_t* read_off,
                          uint32_t n_reads, const int32_t* read_len, const svx_seg_params* params,
                          svx_raw* out);
/* _dev: d_segs 8-byte aligned, d_out 16-byte aligned (any hipMalloc'ed buffer is). */
int svx_segments_classify_dev(svx_ctx* ctx, const svx_seg* d_segs, uint32_t n_segs,
                              const uint32_t* d_read_off, uint32_t n_reads,
                              const int32_t* d_read_len, const svx_seg_params* params,
                              svx_raw* d_out);

/*
 * The three per-read post-passes of analyze_read_segments (SVIM_inter.py:260-338) over the raw records
 * of svx_segments_classify: tandem-duplication merge (:261-290), interspersed duplications from pairs of
 * breakends (:293-320), inversion sweep + complete-linkage clustering (:323-338, :42-60).  The INS / DEL /
 * BND records of `raw` ARE candidates already (one each, in slot order); this call adds the derived ones.
 *
 *   raw, read_off   as written by / passed to svx_segments_classify (slot i of read r = raw[read_off[r] + i])
 *   contig_rank     per reference id: rank of the contig NAME under Python str ordering — the inversion
 *                   sweep sorts by (name, start, end) (:323); ids >= n_contigs rank as themselves
 *   params          min_sv_size / max_sv_size bound the interspersed-duplication length (:309,:313)
 *   out, out_off    derived candidates of read r at out[out_off[r] .. out_off[r] + out_cnt[r]), in the
 *                   reference's order: tandem duplications, interspersed duplications, inversions;
 *                   out_off[r+1] - out_off[r] must be at least svx_segments_postpass_bound(slots of r)
 *                   (= s (s + 3) / 2: s tandems, s (s - 1) / 2 breakend pairs, s inversions), else
 *                   SVX_E_CAPACITY
 * Records (a0..a5):
 *   SVX_POST_TANDEM   ref_id, start, end, copies, fully_covered        (CandidateDuplicationTandem)
 *   SVX_POST_DUP_INT  src ref_id, src start, src end, dst ref_id, dst start, dst end
 *   SVX_POST_INV      ref_id, start, end, complete                       (CandidateInversion)
 */
#define SVX_POST_TANDEM 1
#define SVX_POST_DUP_INT 2
#define SVX_POST_INV 3

typedef struct svx_post {
    int32_t kind;
    int32_t a0, a1, a2, a3, a4, a5;
    int32_t pad;
} svx_post;

uint64_t svx_segments_postpass_bound(uint32_t n_slots);
int svx_segments_postpass(svx_ctx* ctx, const svx_raw* raw, const uint32_t* read_off, uint32_t n_reads,
                          const int32_t* contig_rank, uint32_t n_contigs, const svx_seg_params* params,
                          svx_post* out, const uint64_t* out_off, uint32_t* out_cnt);

/* Asynchronous form on device pointers.  The HOST copies `read_off` / `out_off` size the per-read scratch and
 * are validated exactly like above (nothing of them is read after the call returns); d_read_off / d_out_off are
 * the same arrays in HBM.  Enqueues on the context's stream, does not synchronise. */
int svx_segments_postpass_dev(svx_ctx* ctx, const svx_raw* d_raw, const uint32_t* read_off, const uint32_t* d_read_off,
                              uint32_t n_reads, const int32_t* d_contig_rank, uint32_t n_contigs,
                              const svx_seg_params* params, svx_post* d_out, const uint64_t* out_off,
                              const uint64_t* d_out_off, uint32_t* d_out_cnt);

/*
 * Segment rows of chimeric reads straight from CIGARs in HBM: the dictionaries analyze_read_segments builds per
 * alignment (SVIM_inter.py:66-81) — q_start / q_end from query_alignment_start / _end (flipped with
 * infer_read_length() for reverse records), reference_end = reference_start + Σ{M,D,N,=,X} (1 when that is 0,
 * htslib bam_endpos).  Segment j is alignment d_seg_src[j] of (d_cigar, d_aln_off); d_seg_qend[j] >= 0 overrides
 * query_alignment_end (pysam takes it from the stored sequence when the record has one: l_seq minus the trailing
 * soft clips), -1 derives it from the CIGAR.  d_read_len[r] = infer_read_length() of read r's first segment
 * (its primary, SVIM_inter.py:120).  One wave per segment; asynchronous.
 */
int svx_segments_rows_dev(svx_ctx* ctx, const uint32_t* d_cigar, const uint64_t* d_aln_off, const uint32_t* d_seg_src,
                          const int32_t* d_seg_tid, const int32_t* d_seg_pos, const uint8_t* d_seg_rev,
                          const int32_t* d_seg_qend, uint32_t n_segs, const uint32_t* d_read_off, uint32_t n_reads,
                          svx_seg* d_segs, int32_t* d_read_len);

/* ------------------------------------------------------- COLLECT, whole ---- */
/*
 * The device work of analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83) for every record of a sample —
 * or of both haplotype BAMs of a diploid sample — as ONE submission: svx_cigar_extract_dev over all records
 * (a1 + a2), svx_segments_rows_dev + svx_segments_classify_dev + svx_segments_postpass_dev over the chimeric reads
 * (a3), enqueued on the context's stream behind the uploads; two read-backs (the counts, then exactly the results).
 *
 *   cigar_parts / part_ops   n_parts host arrays of BAM-native CIGAR words, logically back to back (the pools of
 *                 svx_bam_get_columns: page-locked, uploaded where they lie — no concatenation on the host)
 *   aln_off       n_aln + 1 offsets into that concatenation, EVERY record of the pools (records the caller's
 *                 filters drop are walked too — their signatures are masked by the caller; cheaper than compacting
 *                 the pools); ref_start[n_aln]; min_len = options.min_sv_size
 *   extra_cigar / extra_off   CIGARs of the n_extra SA-derived segments (SVIM_COLLECT.py:33-55)
 *   seg_*[n_segs] one row per segment of every chimeric read, [primary] + supplementaries per read
 *                 (SVIM_inter.py:64): seg_src < n_aln names a record of the pools, otherwise extra alignment
 *                 seg_src - n_aln; seg_tid / seg_pos / seg_rev its reference id, start and strand; seg_qend as in
 *                 svx_segments_rows_dev
 *   read_off      n_reads + 1 offsets into the segment rows
 *   contig_rank, params   as for svx_segments_postpass
 *   part_dev / part_ready   optional (NULL: every part is uploaded from cigar_parts): part_dev[k] != NULL says part k's
 *                 part_ops[k] words are in HBM already, on the context's device (svx_bam_device_pool: the reader
 *                 uploads its pool while it assembles it) — cigar_parts[k] is not read; part_ready[k], when
 *                 part_ready != NULL and the entry is not NULL, is a hipEvent_t the part's producer recorded behind its
 *                 last write: the context's stream waits for it, then moves the words into place device-to-device.
 *                 The submission no longer carries 4 bytes per op over the host link between the walk and the kernels.
 * Output (host arrays):
 *   sig, sig_cap, n_sig    as svx_cigar_extract (SVX_E_CAPACITY with n_sig set when sig_cap is too small)
 *   raw[n_segs]            as svx_segments_classify
 *   post, post_off, post_cnt   as svx_segments_postpass (post_off: n_reads + 1 region offsets, chosen by the caller)
 */
typedef struct svx_collect_in {
    const uint32_t* const* cigar_parts;
    const uint64_t* part_ops;
    uint32_t n_parts;
    const uint64_t* aln_off;
    const int32_t* ref_start;
    uint32_t n_aln;
    uint32_t min_len;
    const uint32_t* extra_cigar;
    const uint64_t* extra_off;
    uint32_t n_extra;
    const uint32_t* seg_src;
    const int32_t* seg_tid;
    const int32_t* seg_pos;
    const uint8_t* seg_rev;
    const int32_t* seg_qend;
    uint32_t n_segs;
    const uint32_t* read_off;
    uint32_t n_reads;
    const int32_t* contig_rank;
    uint32_t n_contigs;
    svx_seg_params params;
    const uint32_t* const* part_dev;
    void* const* part_ready;
} svx_collect_in;

typedef struct svx_collect_out {
    svx_sig_soa sig;
    uint64_t sig_cap;
    uint64_t n_sig;
    svx_raw* raw;
    svx_post* post;
    const uint64_t* post_off;
    uint32_t* post_cnt;
} svx_collect_out;

int svx_collect_batch(svx_ctx* ctx, const svx_collect_in* in, svx_collect_out* out);

/*
 * The kernels of svx_collect_batch on inputs that are already in HBM, asynchronously: what that call enqueues
 * between its uploads and its read-backs — svx_cigar_extract_dev (a1 + a2), then svx_segments_rows_dev,
 * svx_segments_classify_dev and svx_segments_postpass_dev (a3) — on the context's stream, no synchronisation.
 *   d_cigar      the pools' words and, from word n_ops on, the CIGARs of the n_extra SA-derived segments
 *   d_aln_off    n_aln + n_extra + 1 offsets into it (PRECONDITIONS as for svx_cigar_extract_dev)
 *   read_off, post_off   HOST copies (sizes of the per-read scratch and output regions; validated by
 *                svx_collect_batch, here a precondition), d_read_off / d_post_off the same in HBM
 *   d_segs, d_read_len   n_segs rows / n_reads lengths of scratch the chain writes and reads
 *   d_chain_deal, n_chain_blocks   optional: the table svx_chain_deal (below) wrote, uploaded by the caller — which
 *                workgroup computes which reads' rows (8-byte aligned, entries within the submission's reads and
 *                segments: a precondition; the kernel clamps what it indexes with).  Results never depend on it.
 * Outputs as svx_collect_batch, in HBM: d_sig (exactly min(count, sig_cap) signatures), d_n_sig (one uint64),
 * d_raw[n_segs], d_post / d_post_cnt.
 */
typedef struct svx_collect_dev {
    const uint32_t* d_cigar;
    uint64_t n_ops;
    const uint64_t* d_aln_off;
    uint32_t n_aln;
    uint32_t n_extra;
    const int32_t* d_ref_start;
    uint32_t min_len;
    const uint32_t* d_seg_src;
    const int32_t* d_seg_tid;
    const int32_t* d_seg_pos;
    const uint8_t* d_seg_rev;
    const int32_t* d_seg_qend;
    uint32_t n_segs;
    const uint32_t* read_off;
    const uint32_t* d_read_off;
    uint32_t n_reads;
    const int32_t* d_contig_rank;
    uint32_t n_contigs;
    svx_seg_params params;
    svx_sig_soa d_sig;
    uint64_t sig_cap;
    uint64_t* d_n_sig;
    svx_seg* d_segs;
    int32_t* d_read_len;
    svx_raw* d_raw;
    svx_post* d_post;
    const uint64_t* post_off;
    const uint64_t* d_post_off;
    uint32_t* d_post_cnt;
    const uint32_t* d_chain_deal;   /* optional (NULL / 0): the table svx_chain_deal wrote, in HBM */
    uint32_t n_chain_blocks;        /* its return value */
} svx_collect_dev;

int svx_collect_batch_dev(svx_ctx* ctx, const svx_collect_dev* d);

/*
 * Deals the chimeric reads of a submission to the workgroups of the split-segment chain by CIGAR OP COUNT.  The
 * reference takes a read's rows from pysam properties that walk the record's CIGAR (reference_end,
 * query_alignment_start / _end, infer_read_length: SVIM_inter.py:66-81); here a workgroup computes the rows of
 * consecutive reads, and a primary can be anything between a few and 10^5 ops, so equal READ counts leave the launch
 * waiting for the workgroup with the longest primaries.  The caller has every offset in hand when it builds the
 * control block: this helper (host arithmetic, O(n_reads + n_segs)) cuts the reads into consecutive ranges of about
 * equal cost — per segment its CIGAR rounded up to the chain's chunk size, plus a constant per read.
 *   read_off[n_reads + 1], seg_src[n_segs]   as in svx_collect_in
 *   aln_off      offsets of EVERY alignment seg_src can name (svx_collect_dev.d_aln_off's host copy)
 *   deal         out: 2 * (n_reads + 2) words are always enough — {first read, first segment} per workgroup and a
 *                closing {n_reads, n_segs} entry
 * Returns the number of workgroups (entries - 1) — upload the table and pass both as d_chain_deal / n_chain_blocks —,
 * 0 when the submission has too few reads for a table to matter (pass NULL / 0: reads are dealt out in equal
 * counts), or SVX_E_INVALID.  Results never depend on the table; svx_collect_batch builds one itself.
 */
int svx_chain_deal(const uint32_t* read_off, uint32_t n_reads, const uint32_t* seg_src, const uint64_t* aln_off,
                   uint32_t* deal);

/* ------------------------------------------------------------ f-1 on the device ------------------- */
/*
 * BGZF members inflated and checked on the device: what htslib's bgzf_read_block does under every record the
 * reference reads (pysam bam.fetch, SVIM_COLLECT.py:65-68) — inflate the member's raw DEFLATE payload, compare
 * the CRC32 and the length with the member's trailer.  All members of a call at once, a wave per member.
 *   d_in                 compressed payloads (the bytes between a member's header and its 8-byte trailer), anywhere
 *                        in one buffer: member m is d_in[d_in_off[m] .. + d_in_len[m]); the buffer must be readable 3 bytes
 *                        past the end of any member (input words are fetched whole)
 *   d_isize, d_crc       the trailer's ISIZE and CRC32 of every member; a member whose ISIZE exceeds 65536 (no BGZF
 *                        member can) is not decoded: status 2, nothing written
 *   d_out, d_out_off     member m's bytes are written to d_out[d_out_off[m] .. + d_isize[m]); the stretches must be
 *                        separated by at least 8 bytes (and d_out end 8 bytes behind the last one): match copies move
 *                        whole 8-byte words and may touch up to 7 bytes behind a member's end
 *   d_status             per member: 0 ok; 1 malformed stream; 2 length != ISIZE; 3 CRC32 mismatch; 4 input ended early.
 *                        The two paddings above are the only bytes outside a member's own input and output stretch that
 *                        are ever read (3 behind the input) or written (7 behind the output).
 * Asynchronous on the context's stream.
 * Three forms with the same bytes and statuses (svx_inflate.hip).  Shipped: three launches — a WAVE per member parses the bit
 * stream (64 stretches of it decoded at once and resynchronised), storing the literals and writing every match as a token
 * (k_inflate_wparse); a lane per member does the same for whatever the wave parse has left alone — anything but plain
 * fixed / dynamic blocks, every malformed stream — and is the judge of those (k_inflate_parse); a wave per member applies the
 * tokens and takes the CRC-32 (k_inflate_resolve).  The token lists come out of the context's workspace, 175 KB a member,
 * for up to 20 480 members at a time (more members: one set of launches behind the other).  The other two: the lane-per-
 * member parse for every member; and ONE launch, a lane per member that decodes and copies (k_bgzf_inflate, rounds 4-5).
 * svx_bgzf_inflate_set_two_pass chooses for the whole process — this entry and the BAM reader's device leg
 * (svx_bam_set_device_inflate): 0 the one-launch kernel, 2 the lane-per-member parse, any other value the shipped form; it
 * returns the previous choice in the same terms (0 / 2 / 1).  SVX_INFLATE_KERNEL=1|2|3 in the environment starts the process
 * on the one-launch / lane-parse / shipped form.
 */
int svx_bgzf_inflate_dev(svx_ctx* ctx, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                         const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                         const uint64_t* d_out_off, uint32_t* d_status);
int svx_bgzf_inflate_set_two_pass(int on);
/* Members whose token lists svx_bgzf_inflate_dev keeps at once (default 20 480, 175 KB each; 0: back to the default): a call
 * with more members goes out in slices of that many.  Returns the previous value.  A memory / latency trade-off — a slice
 * costs at least one member's decode latency — and what the tests use to run many slices over few members. */
uint32_t svx_bgzf_inflate_set_arena(uint32_t members);

/* ------------------------------------------------------------ a5 + a6 ------ */
/*
 * Pair sort + partition: form_partitions (SVIM_COMBINE.py:15-32).
 *
 *   keys[i] = group << 32 | pos, where `group` encodes everything of
 *   Candidate.get_key() (SVCandidate.py:17-19,147-148,292-293,386-387) that
 *   must be EQUAL inside a partition — (type, rank of the contig name under
 *   Python str ordering) — and pos is the non-negative key position.
 *
 * Output: perm[j] = input index of the j-th candidate in stable sorted order
 * (ties keep input order, i.e. hap-1 list then hap-2 list, :17,:182);
 * part_id[j] = partition number of sorted position j; a new partition starts
 * when group differs or |pos - previous pos| > max_dist (strict, :24-26).
 * *n_parts = number of partitions (0 when n == 0).
 */
int svx_pair_partition(svx_ctx* ctx, const uint64_t* keys, uint32_t n, uint32_t max_dist,
                       uint32_t* perm, uint32_t* part_id, uint32_t* n_parts);
int svx_pair_partition_dev(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n, uint32_t max_dist,
                           uint32_t* d_perm, uint32_t* d_part_id, uint32_t* d_n_parts);
/* Same with the caller's knowledge of the key layout: key_bits has a 1 wherever some key may have
 * one (the OR of all keys, or any superset such as `type bits | contig bits | position bits`).
 * Only those bits are sorted on (one launch up to svx_ctx_set_pair_single_launch_max candidates; beyond
 * that P = ceil(live bits / 9) radix passes, P + 2 launches in total), and nothing is read back: fully
 * asynchronous.  svx_pair_partition_dev derives the mask itself with one
 * extra reduction and an 8-byte read-back, which synchronises the stream once. */
int svx_pair_partition_dev_bits(svx_ctx* ctx, const uint64_t* d_keys, uint32_t n, uint32_t max_dist,
                                uint64_t key_bits, uint32_t* d_perm, uint32_t* d_part_id,
                                uint32_t* d_n_parts);

/* ---------------------------------------------------------------- a7 ------- */
/*
 * Batched global (Needleman-Wunsch, unit cost) edit distance, the arithmetic of
 * edlib.align(a, b)["editDistance"] as called by compute_distance
 * (SVIM_COMBINE.py:50,64,76,88,100).  Sequences are bytes in one pool;
 * pair p compares seq[a_off[p] .. a_off[p]+a_len[p]) with
 * seq[b_off[p] .. b_off[p]+b_len[p)).  dist[p] receives the exact distance when
 * it is <= k_max and any value > k_max otherwise (complete linkage cut at
 * max_edit_distance only needs "> t").  k_max = 0xFFFFFFFF
 * requests exact distances.
 */
/* The distance is computed in two stages: a wavefront (diagonal-transition, O(n + d^2)) pass resolves
 * every pair whose distance is at most min(k_max, max_edits, max(64, (|a| + |b|) / 32)) — the haplotypes of one
 * variant are long and nearly identical —, the bit-vector kernel (O(n m / 64), banded) the rest.  Both are
 * exact; max_edits (default 1024, at most 4096, 0 = bit-vector kernel only) only moves work between them. */
int svx_ctx_set_edit_wavefront_cap(svx_ctx* ctx, uint32_t max_edits);

int svx_edit_distance_batch(svx_ctx* ctx, const uint8_t* seq, uint64_t seq_bytes,
                            const uint64_t* a_off, const uint32_t* a_len, const uint64_t* b_off,
                            const uint32_t* b_len, uint32_t n_pairs, uint32_t k_max,
                            uint32_t* dist);

/*
 * The same distances with the haplotype strings assembled on the device.  compute_distance
 * (SVIM_COMBINE.py:43-100) aligns  reference[region_start : c.start] + MIDDLE + reference[c.end : region_end]
 * of two candidates; MIDDLE is empty (DEL), the reverse complement of reference[c.start : c.end] (INV),
 * that interval copies + 1 times (DUP_TAN), the inserted sequence (INS) or the source interval (DUP_INT);
 * reference slices are upper-cased (:45-99 `.upper()`), inserted sequences are taken as they are.  Each
 * haplotype is THREE pieces of one byte pool (reference windows — one per partition is enough — and the
 * inserted sequences): bytes pool[off .. off + len), written `repeat` times (0: piece absent), with
 * SVX_PIECE_UPPER (ASCII str.upper()) and / or SVX_PIECE_REVCOMP (read backwards; A<->T, C<->G after
 * the upper-casing, every other byte unchanged — SVIM_COMBINE.py:63).  pieces holds 6 per pair: the
 * three of haplotype a, then the three of b.  Result contract as svx_edit_distance_batch.
 */
#define SVX_PIECE_UPPER 1
#define SVX_PIECE_REVCOMP 2

typedef struct svx_hap_piece {
    uint64_t off;
    uint32_t len;
    uint16_t repeat;
    uint16_t flags;
} svx_hap_piece;

int svx_haplotype_distance_batch(svx_ctx* ctx, const uint8_t* pool, uint64_t pool_bytes,
                                 const svx_hap_piece* pieces, uint32_t n_pairs, uint32_t k_max, uint32_t* dist);

/*
 * The recipes of a PAIR step's jobs from the candidate columns — host arithmetic (threads), the counterpart of the
 * six `reference.fetch` calls and the string concatenations of compute_distance (SVIM_COMBINE.py:43-100) per pair:
 * job j compares rows job_a[j] and job_b[j] (different haplotypes) of partition job_part[j].  Per partition: its SV
 * type (0 DEL, 1 INV, 2 INS, 3 DUP_TAN, 4 DUP_INT), the length of its contig, and ONE reference window in the pool —
 * [win_lo, ...) starting at pool byte win_base — that covers [min start - 100, max end + 100) of all its members.  The
 * pool continues behind the windows at `extra_at` with (in this order) the DUP_INT source intervals the caller fetched
 * (their places per job and haplotype in mid_off / mid_len, n_jobs x 2; NULL without DUP_INT jobs) and the stretches of
 * the inserted-sequence pool the INS jobs' alleles lie in: returned in seq_range as {lo, hi} of the first haplotype
 * table's part and {lo, hi} of the second's (offsets >= seq_split; seq_split < 0: one table) — the caller appends
 * seqs[lo:hi] of both.  pieces receives 6 per job (3 of haplotype a, 3 of b).  worst_copies: the largest `copies` of a
 * tandem job (SVX_E_TOO_LARGE when a piece would repeat more than 65 535 times).
 */
typedef struct svx_recipe_in {
    uint32_t n_rows;
    const uint8_t* type;
    const int64_t* ss;
    const int64_t* se;
    const int64_t* ds;
    const int64_t* q_off;
    const int64_t* q_len;
    const int64_t* copies;
    uint64_t n_jobs;
    const int64_t* job_a;
    const int64_t* job_b;
    const int64_t* job_part;
    uint32_t n_parts;
    const int64_t* part_type;
    const int64_t* part_len;
    const int64_t* win_base;
    const int64_t* win_lo;
    uint64_t extra_at;
    int64_t seq_split;
    const int64_t* mid_off;
    const int64_t* mid_len;
} svx_recipe_in;

int svx_pair_recipes(const svx_recipe_in* in, svx_hap_piece* pieces, int64_t* seq_range, int64_t* worst_copies);
/* The same with the byte pool already in HBM (d_pool: e.g. one upload of the reference windows serving the
 * thresholded and the exact batch of a PAIR step).  pieces and dist stay host arrays and the call synchronises:
 * the stages of the distance computation are planned from read-backs (which pairs the wavefront pass resolved). */
int svx_haplotype_distance_batch_dev(svx_ctx* ctx, const uint8_t* d_pool, uint64_t pool_bytes,
                                     const svx_hap_piece* pieces, uint32_t n_pairs, uint32_t k_max, uint32_t* dist);
/* One call for a PAIR step's whole batch: k_max[p] per pair — the threshold for the pairs of two-member partitions
 * ("<= max_edit_distance?" is all pair_haplotypes needs of them, SVIM_COMBINE.py:120-139), 0xFFFFFFFF (exact) for
 * the pairs of larger partitions, whose dendrogram above the cut decides scipy's label order.  One upload of the
 * pool, one haplotype assembly, one wavefront pass and one bit-vector pass for all of them.  The waits inside the
 * call sleep on a blocking event instead of spinning on the stream. */
int svx_haplotype_distance_batch_mixed(svx_ctx* ctx, const uint8_t* pool, uint64_t pool_bytes,
                                       const svx_hap_piece* pieces, uint32_t n_pairs, const uint32_t* k_max,
                                       uint32_t* dist);

/*
 * Batched complete linkage + flat cut: for every partition p (n_members[p] candidates, condensed
 * distance vector of n(n-1)/2 doubles, partition after partition in `dist`)
 *     fcluster(linkage(y, method="complete"), cutoff, criterion="distance")
 * as the reference calls scipy at SVIM_COMBINE.py:134-135 (haplotype edit distances, cut at
 * max_edit_distance), :155-156 (breakend span-position distance, cut at 0.3) and SVIM_inter.py:47-48
 * (inversion breakpoints of one read, cut at 0.3).  labels (one per member, partition after
 * partition) are scipy's 1-based flat-cluster labels IN SCIPY'S ORDER — the reference takes the
 * coordinates of a paired call from cluster[0] (SVIM_COMBINE.py:184-363), so the order is part of the
 * result.  Distances are compared exactly as doubles (scipy's float64).  Partitions of one member get
 * label 1.  One lane per partition; any partition size is accepted.
 */
int svx_linkage_cut_batch(svx_ctx* ctx, const double* dist, const uint32_t* n_members, uint32_t n_parts,
                          double cutoff, uint32_t* labels);
/* Asynchronous form: distances, member counts and labels in HBM; the HOST copy of n_members only sizes the scratch
 * of partitions too large for LDS (not read after the call returns).  No synchronisation. */
int svx_linkage_cut_batch_dev(svx_ctx* ctx, const double* d_dist, const uint32_t* n_members,
                              const uint32_t* d_n_members, uint32_t n_parts, double cutoff, uint32_t* d_labels);

#ifdef __cplusplus
}
#endif
#endif /* SVX_H_ */
