/*
 * svx_bam.h — C-ABI of the native BAM ingest of libsvx.so.
 *
 * The reference reads alignments through pysam/htslib: `pysam.AlignmentFile(path)`
 * (svim-asm:63-90), `bam.fetch(contig=...)` (SVIM_COLLECT.py:65), and per record
 * `cigartuples` (SVIM_intra.py:37), `get_cigar_stats()` / `get_tag("SA")`
 * (SVIM_COLLECT.py:11,14), `query_sequence[a:b]` (SVIM_intra.py:42, SVIM_inter.py:117,120).
 * These entry points replace that surface for the hot path with a columnar reader:
 * one call indexes the records of the requested contigs and returns, for all of them at
 * once, the fixed fields, names, aux bytes (SA tag located) and ONE flattened array of
 * BAM-native CIGAR words (`len << 4 | op`, page-locked when a HIP device is present so
 * that the H2D copy of svx_cigar_extract is a true asynchronous DMA).
 *
 * What is inflated: genome-genome alignment records carry SEQ/QUAL fields of 10^5..10^8
 * bases; only the BGZF blocks that hold record headers, names, CIGARs and aux tags are
 * inflated (block sizes come from the BSIZE/ISIZE fields, so SEQ/QUAL blocks are hopped
 * over), plus the blocks of the base ranges later requested with svx_bam_seq_slices.
 * Inflate = the build's own DEFLATE decoder (svx_inflate_raw below; SVX_BAM_ZLIB=1: zlib); a member's CRC32 is
 * checked whenever the member is inflated to its end (svx_bam_set_verify: always).
 * With a `.bai` or a `.csi` next to the BAM — `<file>.bai`, `<stem>.bai`, `<file>.csi`, `<stem>.csi`, in htslib's
 * order (the reference requires an index, svim-asm:67-72; pysam takes either kind) — every bin
 * chunk boundary and linear-index entry is a record boundary: the file is cut there and
 * the pieces are walked by `n_threads` host threads; the per-contig chunk ranges restrict
 * the walk to the contigs a rank owns (contig sharding).  Without a usable
 * index the records are walked sequentially from the header.
 *
 * Long CIGARs: when a record's stored CIGAR starts with a soft clip as long as the read and
 * a `CG:B,I` aux array is present, the array is the real CIGAR (SAM spec §4.2.2; same test
 * as htslib's bam_tag2cigar, which pysam applies on read): it replaces the placeholder
 * and the CG tag is dropped from the record's aux bytes.
 *
 * All functions return SVX_OK (0) or a negative svx_status (svx.h).  A handle is used by
 * one thread at a time.  Pointers handed out stay valid until the next svx_bam_load on the
 * handle or svx_bam_close.
 */
#ifndef SVX_BAM_H_
#define SVX_BAM_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svx_bam svx_bam;

/* Open `path` (memory-mapped), parse the BAM header and the reference dictionary, and read
 * `<path>.bai` / `<stem>.bai` when present.  n_threads <= 0: one per hardware thread, at most 64.
 * On failure *out is NULL and, when err != NULL, a message is copied into err[err_cap]. */
int svx_bam_open(const char* path, int n_threads, svx_bam** out, char* err, size_t err_cap);
void svx_bam_close(svx_bam* bam);
const char* svx_bam_last_error(const svx_bam* bam);

/* Header text (@HD/@SQ... lines, not NUL-terminated: use *l_text) and reference count. */
int svx_bam_header(const svx_bam* bam, const char** text, uint64_t* l_text, int32_t* n_ref);
int svx_bam_reference(const svx_bam* bam, int32_t tid, const char** name, int32_t* length);

/* 0: no index file; 1: .bai parsed and consistent with the file (parallel + per-contig walks);
 * 2: an index file exists but cannot be used (stub, inconsistent .bai / .csi): sequential walk.  (A .csi of any
 *    min_shift / depth is parsed and serves like a .bai: state 1.)
 * `check_index()` of the reference (svim-asm:67) only needs != 0. */
int svx_bam_index_state(const svx_bam* bam);
/* Compressed bytes the index attributes to each contig (0 for contigs without records): the
 * weights of the contig → rank plan.  span[n_ref].  SVX_E_INVALID when index_state != 1. */
int svx_bam_contig_spans(const svx_bam* bam, uint64_t* span);

/* What is inflated of a member that is needed.  Default (on != 0): every member a walk or a sequence slice touches
 * is inflated completely and its CRC32 checked, as htslib does under the reference (bgzf_read_block behind
 * bam.fetch, SVIM_COLLECT.py:65-68) — a damaged member is an error, never silently different variants.
 * on == 0 (opt-in; also SVX_BAM_VERIFY=0 in the environment, `svim-asm --no_bgzf_crc`): a member is inflated only
 * as far as the last byte asked for (a record walk stops right behind a record's CIGAR instead of going on into
 * its SEQ bytes, a sequence slice at its last base: half the CPU time of the ingest) and its CRC32 is checked only
 * when it happens to be inflated to its end.  Either way a malformed DEFLATE stream, a stream longer than its
 * ISIZE or a stream that ends early is an error. */
int svx_bam_set_verify(svx_bam* bam, int on);

/* Page-lock the CIGAR pool of later svx_bam_load calls in the context of HIP device `device`
 * (the one the svx_ctx that will consume it lives on); device < 0 (default): pageable memory. */
int svx_bam_set_pinned_device(svx_bam* bam, int device);

/* The CIGAR pool's copy in HBM.  With a pinned device set, svx_bam_load also uploads the pool to that device, part by
 * part on a stream of its own WHILE it assembles the pool from the walkers' chunks (the last part leaves within
 * microseconds of the walk's end), so that svx_collect_batch can take it where it lies (svx_collect_in.part_dev /
 * part_ready in svx.h) instead of uploading 4 bytes per op between the walk and the kernels.
 *   *d_cigar  device address of the n_ops words of svx_bam_columns.cigar, NULL when there is no copy (no pinned device,
 *             an empty pool, SVX_BAM_DEVICE_POOL=0 in the environment, or no memory: the caller uploads as before)
 *   *ready    a hipEvent_t recorded behind the last part: a consumer on another stream waits for it
 *             (hipStreamWaitEvent) before it reads *d_cigar
 * Both stay valid until the next svx_bam_load on the handle or svx_bam_close.
 * svx_bam_device_pool_wait blocks until the copy is complete; *waited_us = how long that took (0 without a copy). */
int svx_bam_device_pool(svx_bam* bam, const uint32_t** d_cigar, uint64_t* n_ops, void** ready);
int svx_bam_device_pool_wait(svx_bam* bam, double* waited_us);

/* A share of svx_bam_seq_slices on the device.  percent > 0 (with a pinned device set, verification on — the default —
 * and a call of at least 2048 slices): the BGZF members under the first `percent` % of a call's slices are inflated and
 * checked (CRC32, ISIZE: the same judgement as the host decoder's, svx_bgzf_inflate_dev's kernel) on the pinned device
 * while the handle's threads decode the members of the other slices; the slices' packed bases come back, nothing else.
 * DEFLATE on the device is a wave per member (svx_inflate.hip): 4 ms for a thousand members, 13 for a full-size call's
 * 14 000 — the whole call where CPU-seconds are what a run is short of; results never depend on the share.  0 (default):
 * host only.
 * svx_bam_device_members: members the device has inflated for this handle so far. */
int svx_bam_set_device_inflate(svx_bam* bam, int percent);
/* The share goes to the device only when it holds at least `members` BGZF members (default 500: a launch costs one member's
 * latency on the device, 3-4 ms, and the leg its staging — below that the threads are through sooner.  BASELINE config 5's
 * 1 200-member calls: the same wall-clock on the device, 0.6 instead of 1.0 CPU-seconds: profiles/r06_wave_min_members.txt). */
int svx_bam_set_device_inflate_min(svx_bam* bam, uint32_t members);
/* The record walks' share of the default check, deferred to the device.  With `on` (and a device share, a pinned device and
 * the check itself on) svx_bam_load's walks inflate a member only as far as the bytes they need — as they do with the check
 * off: a third of the CPU time — and note every member they took bytes from; the next svx_bam_seq_slices call hands those
 * members to its device leg, which inflates them whole and checks CRC32 and ISIZE beside its own members (no leg in that
 * call: the handle's threads do it before the call returns).  A caller that sets this MUST call svx_bam_verify_pending
 * before it trusts the records when no svx_bam_seq_slices call follows the load: it checks what is still pending on the
 * threads (SVX_E_INVALID for a damaged member, like the load itself would have said).  svx_bam_pending_members: how many
 * members wait for their check.  Off by default: every load returns with its members checked. */
int svx_bam_set_defer_verify(svx_bam* bam, int on);
int svx_bam_verify_pending(svx_bam* bam);
uint64_t svx_bam_pending_members(const svx_bam* bam);
/* A device has two inflate lanes by default (stream + page-locked ring each; svx_bam_set_inflate_lanes): a call that finds them all taken — a process with more
 * than two readers decoding at once, svim-asm-cohort's workers — gives its whole call to the threads (default, 0) or
 * sleeps up to `milliseconds` for the first lane to come free: the better choice where the process's wall-clock is its
 * CPU-seconds over a CPU quota and the device would otherwise idle. */
int svx_bam_set_device_inflate_wait(svx_bam* bam, uint32_t milliseconds);
/* Inflate lanes per device of this process, 1..16 (default 2: a diploid sample's two readers), effective for devices whose
 * lanes have not been brought up yet (the first load with a device share does that): a process that keeps many readers
 * decoding at once asks for more before its first load (svim-asm-cohort: one per reader its workers keep in flight).  Each
 * lane is a stream and 16 MiB of page-locked ring. */
int svx_bam_set_inflate_lanes(int lanes);
uint64_t svx_bam_device_members(const svx_bam* bam);

/* Index the records of contigs tids[0..n_tids) (NULL: every record of the file, unplaced ones
 * included), in file order. */
int svx_bam_load(svx_bam* bam, const int32_t* tids, int32_t n_tids);

typedef struct svx_bam_columns {
    uint64_t n_records;
    const int32_t* tid;        /* refID                                            */
    const int32_t* pos;        /* 0-based leftmost coordinate (reference_start)     */
    const int32_t* l_seq;      /* stored sequence length                           */
    const int32_t* ref_len;    /* Σ len over {M,D,N,=,X} of the (real) CIGAR        */
    const uint16_t* flag;
    const uint8_t* mapq;
    const uint64_t* cigar_off; /* n_records + 1 offsets into `cigar`               */
    const uint32_t* cigar;     /* BAM-native words of all records, back to back    */
    const uint64_t* name_off;  /* n_records + 1 offsets into `names` (no NULs)      */
    const char* names;
    const uint64_t* aux_off;   /* n_records + 1 offsets into `aux` (CG tag removed) */
    const uint8_t* aux;
    const int64_t* sa_off;     /* offset into `aux` of the SA:Z string, -1 if none  */
    const uint32_t* sa_len;    /* its length without the NUL                       */
    const uint64_t* voffset;   /* BGZF virtual offset of the record (index builder) */
    uint64_t blocks_inflated;  /* BGZF blocks inflated so far on this handle        */
    uint64_t blocks_spanned;   /* BGZF blocks inside the walked file ranges         */
    int cigar_pinned;          /* 1 when `cigar` is page-locked host memory         */
    int n_threads;
} svx_bam_columns;

int svx_bam_get_columns(const svx_bam* bam, svx_bam_columns* out);

/* Decode bases [begin[i], end[i]) of record rec[i] (index into the loaded columns) to ASCII
 * (=ACMGRSVTWYHKDBN, BAM orientation) at out + out_off[i]; n slices, walked by the handle's
 * threads.  Ranges are clipped to [0, l_seq]; out_off[i+1] - out_off[i] must hold the
 * clipped length.  Slices sorted by (rec, begin) reuse inflated blocks.  A slice is 0.3 % of the 64 KiB member it
 * sits in: members are inflated only up to the last byte a slice needs unless svx_bam_set_verify is on. */
int svx_bam_seq_slices(svx_bam* bam, const uint32_t* rec, const uint32_t* begin, const uint32_t* end,
                       uint32_t n, const uint64_t* out_off, uint8_t* out);

/*
 * The DEFLATE decoder behind all of the above (svim_asm_amd/csrc/svx_inflate.h), callable on its own: raw RFC 1951
 * stream in[in_len] → out[cap].  The decoder is first run up to each of the n_stops output positions in turn (the way
 * svx_bam_seq_slices extends the inflated prefix of a member; a run may overshoot its stop by up to one match), then
 * to the end of the stream.  *n_out = bytes produced.  SVX_E_INVALID: malformed stream, a stream that yields more
 * than cap bytes, or one that ends before a stop.  Replaces zlib's inflate() under htslib's bgzf_read
 * (pysam, SVIM_COLLECT.py:68); zlib is the oracle of tests/test_inflate.py.
 */
int svx_inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, const uint64_t* stops,
                    uint32_t n_stops, uint64_t* n_out);

/* Two independent streams decoded side by side in one thread (the rounds of one fill the issue slots the other's
 * dependency chain leaves empty; svx_bam_seq_slices inflates its members two at a time this way).  stop_x = the output
 * position to reach (a run may overshoot it by up to one match), or UINT64_MAX: to the end of the stream, which must then
 * yield at most cap_x bytes.  Per stream: *n_out_x bytes produced, *rc_x SVX_OK or SVX_E_INVALID, exactly as
 * svx_inflate_raw on that stream alone would report.  The return value only reflects the arguments. */
int svx_inflate_raw_pair(const uint8_t* in_a, size_t in_len_a, uint8_t* out_a, size_t cap_a, uint64_t stop_a,
                         uint64_t* n_out_a, int* rc_a, const uint8_t* in_b, size_t in_len_b, uint8_t* out_b,
                         size_t cap_b, uint64_t stop_b, uint64_t* n_out_b, int* rc_b);

#ifdef __cplusplus
}
#endif
#endif /* SVX_BAM_H_ */
