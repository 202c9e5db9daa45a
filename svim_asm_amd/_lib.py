"""ctypes binding of libsvx.so (C-ABI declared in include/svx.h).

The product path has no CPU fallback: if the HIP library is missing or no device is
usable, loading / context creation raises.  numpy arrays go through the host-pointer
entry points; torch tensors already in HBM go through the *_dev entry points.
"""
import ctypes as C
import os
import threading

import numpy as np

from svim_asm_amd import _warm

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVX_LIB") or os.path.join(HERE, "libsvx.so")  # SVX_LIB: A/B builds

SVX_OK = 0
SVX_E_INVALID = -1
SVX_E_CAPACITY = -2
SVX_E_HIP = -3
SVX_E_NOMEM = -4
SVX_E_TOO_LARGE = -5
SVX_E_NODEVICE = -6
_STATUS = {0: "SVX_OK", -1: "SVX_E_INVALID", -2: "SVX_E_CAPACITY", -3: "SVX_E_HIP", -4: "SVX_E_NOMEM",
           -5: "SVX_E_TOO_LARGE", -6: "SVX_E_NODEVICE"}

SIG_INS, SIG_DEL = 0, 1
RAW_NONE, RAW_INS, RAW_DEL, RAW_BND, RAW_TANDEM, RAW_INV = range(6)
POST_TANDEM, POST_DUP_INT, POST_INV = 1, 2, 3


class SvxError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("%s: %s" % (_STATUS.get(status, status), msg))


class SigSoa(C.Structure):
    _fields_ = [("aln", C.c_void_p), ("ref_pos", C.c_void_p), ("read_pos", C.c_void_p),
                ("len", C.c_void_p), ("type", C.c_void_p)]


class AlnStats(C.Structure):
    _fields_ = [("ref_len", C.c_void_p), ("q_start", C.c_void_p), ("q_end", C.c_void_p),
                ("read_len", C.c_void_p), ("n_hard", C.c_void_p)]


class SegParams(C.Structure):
    _fields_ = [("min_sv_size", C.c_int32), ("max_sv_size", C.c_int32),
                ("query_gap_tolerance", C.c_int32), ("query_overlap_tolerance", C.c_int32),
                ("reference_gap_tolerance", C.c_int32), ("reference_overlap_tolerance", C.c_int32)]


class VcfIn(C.Structure):
    """svx_vcf_in (include/svx_text.h)."""
    _fields_ = [("n_rows", C.c_uint32), ("sc", C.c_void_p), ("ss", C.c_void_p), ("se", C.c_void_p), ("dc", C.c_void_p),
                ("ds", C.c_void_p), ("de", C.c_void_p), ("flag", C.c_void_p), ("copies", C.c_void_p), ("gt", C.c_void_p),
                ("q_off", C.c_void_p), ("q_len", C.c_void_p), ("r_off", C.c_void_p), ("r_flat", C.c_void_p),
                ("seqs", C.c_void_p), ("seqs_bytes", C.c_uint64), ("names", C.c_void_p), ("name_off", C.c_void_p),
                ("n_names", C.c_uint64), ("contigs", C.c_void_p),
                ("contig_off", C.c_void_p), ("contig_rank", C.c_void_p), ("n_contigs", C.c_uint32),
                ("genotypes", C.c_void_p), ("genotype_off", C.c_void_p), ("n_genotypes", C.c_uint32),
                ("n_entries", C.c_uint32), ("kind", C.c_void_p), ("row", C.c_void_p), ("bases", C.c_void_p),
                ("bases_bytes", C.c_uint64), ("b_off", C.c_void_p), ("b_len", C.c_void_p), ("b2_off", C.c_void_p), ("b2_len", C.c_void_p),
                ("sequence_alleles", C.c_int), ("read_names", C.c_int)]


VCF_DEL, VCF_INV, VCF_INS, VCF_DUPTAN_INS, VCF_DUPTAN_DUP, VCF_DUPINT_INS, VCF_DUPINT_DUP, VCF_BND, VCF_BND_REV = range(9)

class CollectIn(C.Structure):
    """svx_collect_in (include/svx.h)."""
    _fields_ = [("cigar_parts", C.POINTER(C.c_void_p)), ("part_ops", C.c_void_p), ("n_parts", C.c_uint32),
                ("aln_off", C.c_void_p), ("ref_start", C.c_void_p), ("n_aln", C.c_uint32), ("min_len", C.c_uint32),
                ("extra_cigar", C.c_void_p), ("extra_off", C.c_void_p), ("n_extra", C.c_uint32),
                ("seg_src", C.c_void_p), ("seg_tid", C.c_void_p), ("seg_pos", C.c_void_p), ("seg_rev", C.c_void_p),
                ("seg_qend", C.c_void_p), ("n_segs", C.c_uint32), ("read_off", C.c_void_p), ("n_reads", C.c_uint32),
                ("contig_rank", C.c_void_p), ("n_contigs", C.c_uint32), ("params", SegParams),
                ("part_dev", C.POINTER(C.c_void_p)), ("part_ready", C.POINTER(C.c_void_p))]


class CollectDev(C.Structure):
    """svx_collect_dev (include/svx.h)."""
    _fields_ = [("d_cigar", C.c_void_p), ("n_ops", C.c_uint64), ("d_aln_off", C.c_void_p), ("n_aln", C.c_uint32),
                ("n_extra", C.c_uint32), ("d_ref_start", C.c_void_p), ("min_len", C.c_uint32), ("d_seg_src", C.c_void_p),
                ("d_seg_tid", C.c_void_p), ("d_seg_pos", C.c_void_p), ("d_seg_rev", C.c_void_p), ("d_seg_qend", C.c_void_p),
                ("n_segs", C.c_uint32), ("read_off", C.c_void_p), ("d_read_off", C.c_void_p), ("n_reads", C.c_uint32),
                ("d_contig_rank", C.c_void_p), ("n_contigs", C.c_uint32), ("params", SegParams), ("d_sig", SigSoa),
                ("sig_cap", C.c_uint64), ("d_n_sig", C.c_void_p), ("d_segs", C.c_void_p), ("d_read_len", C.c_void_p),
                ("d_raw", C.c_void_p), ("d_post", C.c_void_p), ("post_off", C.c_void_p), ("d_post_off", C.c_void_p),
                ("d_post_cnt", C.c_void_p), ("d_chain_deal", C.c_void_p), ("n_chain_blocks", C.c_uint32)]


class RecipeIn(C.Structure):
    """svx_recipe_in (include/svx.h)."""
    _fields_ = [("n_rows", C.c_uint32), ("type", C.c_void_p), ("ss", C.c_void_p), ("se", C.c_void_p), ("ds", C.c_void_p),
                ("q_off", C.c_void_p), ("q_len", C.c_void_p), ("copies", C.c_void_p), ("n_jobs", C.c_uint64),
                ("job_a", C.c_void_p), ("job_b", C.c_void_p), ("job_part", C.c_void_p), ("n_parts", C.c_uint32),
                ("part_type", C.c_void_p), ("part_len", C.c_void_p), ("win_base", C.c_void_p), ("win_lo", C.c_void_p),
                ("extra_at", C.c_uint64), ("seq_split", C.c_int64), ("mid_off", C.c_void_p), ("mid_len", C.c_void_p)]


class CollectOut(C.Structure):
    """svx_collect_out (include/svx.h)."""
    _fields_ = [("sig", SigSoa), ("sig_cap", C.c_uint64), ("n_sig", C.c_uint64), ("raw", C.c_void_p),
                ("post", C.c_void_p), ("post_off", C.c_void_p), ("post_cnt", C.c_void_p)]


SEG_DTYPE = np.dtype([("q_start", "<i4"), ("q_end", "<i4"), ("ref_id", "<i4"), ("ref_start", "<i4"),
                      ("ref_end", "<i4"), ("is_reverse", "<i4")])
HAP_PIECE_DTYPE = np.dtype([("off", "<u8"), ("len", "<u4"), ("repeat", "<u2"), ("flags", "<u2")])  # svx_hap_piece
PIECE_UPPER, PIECE_REVCOMP = 1, 2
RAW_DTYPE = np.dtype([("kind", "<i4"), ("a0", "<i4"), ("a1", "<i4"), ("a2", "<i4"), ("a3", "<i4"),
                      ("a4", "<i4"), ("a5", "<i4"), ("pad", "<i4")])

# symbol -> (restype, argtypes); must list every symbol include/svx.h declares
_P = C.c_void_p
SYMBOLS = {
    "svx_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "svx_ctx_create_on_stream": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "svx_ctx_destroy": (None, [_P]),
    "svx_ctx_sync": (C.c_int, [_P]),
    "svx_last_error": (C.c_char_p, [_P]),
    "svx_version": (C.c_char_p, []),
    "svx_device_count": (C.c_int, []),
    "svx_ctx_set_small_batch_ops": (C.c_int, [_P, C.c_uint64]),
    "svx_bgzf_inflate_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P]),
    "svx_hbm_read_probe_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.POINTER(C.c_float)]),
    "svx_ctx_set_split_chain": (C.c_int, [_P, C.c_int]),
    "svx_bgzf_inflate_set_two_pass": (C.c_int, [C.c_int]),
    "svx_bgzf_inflate_set_arena": (C.c_uint32, [C.c_uint32]),
    "svx_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "svx_ctx_set_pair_single_launch_max": (C.c_int, [_P, C.c_uint32]),
    "svx_ctx_barrier_timed_out": (C.c_int, [_P]),
    "svx_ctx_set_edit_wavefront_cap": (C.c_int, [_P, C.c_uint32]),
    "svx_ctx_set_pair_wait_free": (C.c_int, [_P, C.c_int]),
    "svx_ctx_pair_retries": (C.c_int, [_P]),
    "svx_dev_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "svx_dev_free": (C.c_int, [_P, _P]),
    "svx_dev_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "svx_dev_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "svx_ctx_set_timing": (C.c_int, [_P, C.c_int]),
    "svx_ctx_last_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "svx_ctx_wait_dominant": (C.c_int, [_P, _P]),
    "svx_cigar_extract": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, SigSoa, C.c_uint64,
                                    C.POINTER(C.c_uint64)]),
    "svx_cigar_extract_soa": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, SigSoa,
                                        C.c_uint64, C.POINTER(C.c_uint64)]),
    "svx_cigar_extract_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P, C.c_uint32, SigSoa,
                                        C.c_uint64, _P]),
    "svx_cigar_extract_soa_dev": (C.c_int, [_P, _P, _P, C.c_uint64, _P, C.c_uint32, _P, C.c_uint32,
                                            SigSoa, C.c_uint64, _P]),
    "svx_cigar_stats": (C.c_int, [_P, _P, _P, C.c_uint32, AlnStats]),
    "svx_cigar_stats_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, AlnStats]),
    "svx_segments_classify": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.POINTER(SegParams), _P]),
    "svx_segments_classify_dev": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P,
                                            C.POINTER(SegParams), _P]),
    "svx_segments_postpass_bound": (C.c_uint64, [C.c_uint32]),
    "svx_segments_postpass": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(SegParams), _P, _P, _P]),
    "svx_segments_postpass_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(SegParams), _P, _P, _P, _P]),
    "svx_segments_rows_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P]),
    "svx_collect_batch": (C.c_int, [_P, C.POINTER(CollectIn), C.POINTER(CollectOut)]),
    "svx_collect_batch_dev": (C.c_int, [_P, C.POINTER(CollectDev)]),
    "svx_chain_deal": (C.c_int, [_P, C.c_uint32, _P, _P, _P]),
    "svx_linkage_cut_batch_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_double, _P]),
    "svx_haplotype_distance_batch_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, _P]),
    "svx_pair_partition": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, C.POINTER(C.c_uint32)]),
    "svx_pair_partition_dev": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "svx_pair_partition_dev_bits": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint64, _P, _P, _P]),
    "svx_edit_distance_batch": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P, C.c_uint32, C.c_uint32,
                                          _P]),
    "svx_haplotype_distance_batch": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, _P]),
    "svx_haplotype_distance_batch_mixed": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P, _P]),
    "svx_pair_recipes": (C.c_int, [C.POINTER(RecipeIn), _P, _P, _P]),
    "svx_linkage_cut_batch": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_double, _P]),
    # native BAM ingest (include/svx_bam.h)
    "svx_bam_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "svx_bam_close": (None, [_P]),
    "svx_bam_last_error": (C.c_char_p, [_P]),
    "svx_bam_header": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "svx_bam_reference": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
    "svx_bam_index_state": (C.c_int, [_P]),
    "svx_bam_contig_spans": (C.c_int, [_P, _P]),
    "svx_bam_set_pinned_device": (C.c_int, [_P, C.c_int]),
    "svx_bam_set_verify": (C.c_int, [_P, C.c_int]),
    "svx_bam_set_device_inflate": (C.c_int, [_P, C.c_int]),
    "svx_bam_set_device_inflate_min": (C.c_int, [_P, C.c_uint32]),
    "svx_bam_set_device_inflate_wait": (C.c_int, [_P, C.c_uint32]),
    "svx_bam_set_defer_verify": (C.c_int, [_P, C.c_int]),
    "svx_bam_set_inflate_lanes": (C.c_int, [C.c_int]),
    "svx_bam_verify_pending": (C.c_int, [_P]),
    "svx_bam_pending_members": (C.c_uint64, [_P]),
    "svx_bam_device_members": (C.c_uint64, [_P]),
    "svx_bam_device_pool": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(_P)]),
    "svx_bam_device_pool_wait": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "svx_bam_load": (C.c_int, [_P, _P, C.c_int32]),
    "svx_bam_get_columns": (C.c_int, [_P, _P]),
    "svx_bam_seq_slices": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, _P]),
    "svx_inflate_raw": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, C.POINTER(C.c_uint64)]),
    "svx_inflate_raw_pair": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                       _P, C.c_size_t, _P, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    # native text side (include/svx_text.h)
    "svx_fasta_open": (C.c_int, [C.c_char_p, C.c_int32, _P, _P, _P, _P, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "svx_fasta_close": (None, [_P]),
    "svx_fasta_fetch_batch": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_int, _P, _P, C.c_int]),
    "svx_vcf_format": (C.c_int, [C.POINTER(VcfIn), C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "svx_vcf_free": (None, [_P]),
    "svx_vcf_write": (C.c_int, [C.POINTER(VcfIn), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}

_lib = None


def load():
    """Load libsvx.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvxError(SVX_E_NODEVICE, "libsvx.so not built (%s); run `python -m svim_asm_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _as(a, dtype):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=dtype)


class Context:
    """One device + one stream + scratch HBM (svx_ctx).  Not thread-safe; one per GPU."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _P()
        warm = _warm.take(device) if stream is None else None
        if warm is not None:
            h, rc = warm, SVX_OK  # created while the process was still importing (svim_asm_amd/_warm.py)
        elif stream is None:
            rc = self.lib.svx_ctx_create(int(device), C.byref(h))
        else:
            rc = self.lib.svx_ctx_create_on_stream(int(device), _P(stream), C.byref(h))
        if rc != SVX_OK:
            raise SvxError(rc, "svx_ctx_create(device=%d) failed — a HIP device is required; "
                           "there is no CPU fallback" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.svx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SVX_OK:
            raise SvxError(rc, (self.lib.svx_last_error(self.h) or b"").decode())

    def sync(self):
        self._check(self.lib.svx_ctx_sync(self.h))

    def set_small_batch_ops(self, max_ops):
        """Largest batch (CIGAR ops) of the small-batch (two-launch) path; 0 forces the streaming path."""
        self._check(self.lib.svx_ctx_set_small_batch_ops(self.h, int(max_ops)))

    def set_split_chain(self, on=True):
        """The split-segment chain of collect_batch as three single-purpose launches instead of one fused kernel."""
        self._check(self.lib.svx_ctx_set_split_chain(self.h, 1 if on else 0))

    def set_pair_single_launch_max(self, max_candidates):
        """Largest batch (candidates) of the one-launch pair sort; 0 forces the radix path."""
        self._check(self.lib.svx_ctx_set_pair_single_launch_max(self.h, int(max_candidates)))

    def set_pair_wait_free(self, on=True):
        """Sort on the plan without waits between workgroups inside a launch (radix passes + two-launch sweep)."""
        self._check(self.lib.svx_ctx_set_pair_wait_free(self.h, 1 if on else 0))

    def barrier_timed_out(self):
        """True once after a sync failed because a wait between workgroups ran out (svx_ctx_barrier_timed_out)."""
        return bool(self.lib.svx_ctx_barrier_timed_out(self.h))

    def pair_retries(self):
        """Host-pointer pair_partition calls re-run on the wait-free plan after a wait of the fast plans ran out."""
        return int(self.lib.svx_ctx_pair_retries(self.h))

    def set_edit_wavefront_cap(self, max_edits):
        """Edits the wavefront pass of the edit distance resolves (0: bit-vector kernel only)."""
        self._check(self.lib.svx_ctx_set_edit_wavefront_cap(self.h, int(max_edits)))

    # ---------------------------------------------------------------- device buffers
    def dev_array(self, host=None, nbytes=None):
        """HBM buffer owned by this context (svx_dev_malloc); `host`: numpy array uploaded into it."""
        return DeviceArray(self, host=host, nbytes=nbytes)

    def wait_dominant(self, other):
        """Order this context's next launches after `other`'s latest streaming kernel (pipelining)."""
        self._check(self.lib.svx_ctx_wait_dominant(self.h, other.h))

    def set_timing(self, on=True):
        self._check(self.lib.svx_ctx_set_timing(self.h, 1 if on else 0))

    def bgzf_inflate(self, payloads, isize, crc, keep_output=True):
        """Inflate + CRC-check BGZF member payloads on the device (svx_bgzf_inflate_dev: the kernel of the reader's device leg).  payloads: list of
        bytes-like raw DEFLATE streams; isize / crc: the members' trailers.  Returns (status u32[n], outputs or None,
        kernel milliseconds)."""
        n = len(payloads)
        in_len = np.array([len(p) for p in payloads], dtype=np.uint32)
        in_off = np.zeros(n, np.uint64)
        if n > 1:
            np.cumsum(((in_len[:-1].astype(np.uint64) + 3) // 4) * 4, out=in_off[1:])
        total_in = int(in_off[-1] + in_len[-1]) if n else 0
        blob = np.zeros(total_in + 8, np.uint8)
        for p, o, l in zip(payloads, in_off.tolist(), in_len.tolist()):
            blob[o:o + l] = np.frombuffer(p, dtype=np.uint8)
        isize = np.ascontiguousarray(isize, dtype=np.uint32)
        crc = np.ascontiguousarray(crc, dtype=np.uint32)
        out_off = np.zeros(n, np.uint64)
        if n > 1:
            np.cumsum(((isize[:-1].astype(np.uint64) + 8 + 15) // 16) * 16, out=out_off[1:])  # >= 8 bytes between members
        total_out = int(out_off[-1] + isize[-1]) if n else 0
        d = [self.dev_array(x) for x in (blob, in_off, in_len, isize, crc, out_off)]
        d_out, d_st = self.dev_array(nbytes=total_out + 16), self.dev_array(np.full(n, 0xFFFFFFFF, np.uint32))
        self.set_timing(True)
        self._check(self.lib.svx_bgzf_inflate_dev(self.h, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, n, d_out.ptr,
                                                  d[5].ptr, d_st.ptr))
        self.sync()
        ms = self.last_kernel_ms()[1]
        self.set_timing(False)
        status = d_st.download(np.uint32)
        outs = None
        if keep_output:
            flat = d_out.download(np.uint8)
            outs = [flat[int(o):int(o) + int(l)].tobytes() for o, l in zip(out_off.tolist(), isize.tolist())]
        for x in d + [d_out, d_st]:
            x.free()
        return status, outs, ms

    def hbm_read_probe(self, d_ptr, nbytes, reps=5):
        """GB/s of a read-only nontemporal stream over a resident buffer (svx_hbm_read_probe_dev)."""
        ms = C.c_float()
        self._check(self.lib.svx_hbm_read_probe_dev(self.h, d_ptr, int(nbytes), int(reps), C.byref(ms)))
        return nbytes / (ms.value * 1e-3) / 1e9

    def last_kernel_ms(self):
        t, d = C.c_float(), C.c_float()
        self._check(self.lib.svx_ctx_last_kernel_ms(self.h, C.byref(t), C.byref(d)))
        return t.value, d.value

    # ---------------------------------------------------------------- a1 + a2
    def cigar_extract(self, cigar, aln_off, ref_start=None, min_len=40, cap=None, op=None):
        """Batch analyze_cigar_indel.  `cigar`: packed u32 words (or lengths when `op` is given).
        Returns dict(aln, ref_pos, read_pos, len, type) in (alignment, op) order."""
        cigar = _as(cigar, np.uint32)
        aln_off = _as(aln_off, np.uint64)
        ref_start = _as(ref_start, np.int32)
        op = _as(op, np.uint8)
        n_aln = len(aln_off) - 1 if len(aln_off) else 0
        n_ops = int(aln_off[-1]) if n_aln else 0
        if len(cigar) < n_ops or (op is not None and len(op) < n_ops):
            raise SvxError(SVX_E_INVALID, "cigar shorter than aln_off[-1]")
        if ref_start is not None and len(ref_start) != n_aln:
            raise SvxError(SVX_E_INVALID, "ref_start length != n_aln")
        if cap is None:
            cap = max(1024, n_ops // 16)
        while True:
            out = {"aln": np.empty(cap, np.uint32), "ref_pos": np.empty(cap, np.uint32),
                   "read_pos": np.empty(cap, np.uint32), "len": np.empty(cap, np.uint32),
                   "type": np.empty(cap, np.uint8)}
            soa = SigSoa(_ptr(out["aln"]), _ptr(out["ref_pos"]), _ptr(out["read_pos"]),
                         _ptr(out["len"]), _ptr(out["type"]))
            n = C.c_uint64(0)
            if op is None:
                rc = self.lib.svx_cigar_extract(self.h, _ptr(cigar), _ptr(aln_off), n_aln,
                                                _ptr(ref_start), int(min_len), soa, cap, C.byref(n))
            else:
                rc = self.lib.svx_cigar_extract_soa(self.h, _ptr(op), _ptr(cigar), _ptr(aln_off), n_aln,
                                                    _ptr(ref_start), int(min_len), soa, cap, C.byref(n))
            if rc == SVX_E_CAPACITY:
                cap = int(n.value)
                continue
            self._check(rc)
            k = int(n.value)
            return {key: v[:k] for key, v in out.items()}

    def cigar_extract_dev(self, d_cigar, n_ops, d_aln_off, n_aln, d_ref_start, min_len, d_out, cap,
                          d_n_out, d_op=None):
        """Device-pointer form; arguments are integer device addresses (tensor.data_ptr())."""
        soa = SigSoa(*d_out)
        if d_op is None:
            rc = self.lib.svx_cigar_extract_dev(self.h, d_cigar, n_ops, d_aln_off, n_aln, d_ref_start,
                                                int(min_len), soa, cap, d_n_out)
        else:
            rc = self.lib.svx_cigar_extract_soa_dev(self.h, d_op, d_cigar, n_ops, d_aln_off, n_aln,
                                                    d_ref_start, int(min_len), soa, cap, d_n_out)
        self._check(rc)

    def cigar_stats(self, cigar, aln_off):
        cigar = _as(cigar, np.uint32)
        aln_off = _as(aln_off, np.uint64)
        n_aln = len(aln_off) - 1 if len(aln_off) else 0
        out = {k: np.zeros(n_aln, np.uint32) for k in ("ref_len", "q_start", "q_end", "read_len", "n_hard")}
        st = AlnStats(*[_ptr(out[k]) for k in ("ref_len", "q_start", "q_end", "read_len", "n_hard")])
        self._check(self.lib.svx_cigar_stats(self.h, _ptr(cigar), _ptr(aln_off), n_aln, st))
        return out

    # ---------------------------------------------------------------- a3
    def segments_classify(self, segs, read_off, read_len, params):
        segs = np.ascontiguousarray(segs, dtype=SEG_DTYPE)
        read_off = _as(read_off, np.uint32)
        read_len = _as(read_len, np.int32)
        n_reads = len(read_off) - 1 if len(read_off) else 0
        out = np.zeros(len(segs), dtype=RAW_DTYPE)
        if n_reads == 0 or len(segs) == 0:
            return out
        p = params if isinstance(params, SegParams) else SegParams(*[int(x) for x in params])
        self._check(self.lib.svx_segments_classify(self.h, _ptr(segs), _ptr(read_off), n_reads,
                                                   _ptr(read_len), C.byref(p), _ptr(out)))
        return out

    def segments_postpass(self, raw, read_off, contig_rank, params):
        """Derived candidates (tandem / interspersed duplications, inversions) of every read from its raw
        adjacency records.  Returns (post records, first record per read [n_reads + 1])."""
        raw = np.ascontiguousarray(raw, dtype=RAW_DTYPE)
        read_off = _as(read_off, np.uint32)
        contig_rank = _as(contig_rank, np.int32)
        n_reads = len(read_off) - 1 if len(read_off) else 0
        if n_reads == 0:
            return np.zeros(0, dtype=RAW_DTYPE), np.zeros(1, np.int64)
        slots = np.diff(read_off.astype(np.int64))
        out_off = np.concatenate(([0], np.cumsum(slots * (slots + 3) // 2))).astype(np.uint64)
        out = np.zeros(int(out_off[-1]), dtype=RAW_DTYPE)  # svx_post has svx_raw's layout: kind, a0..a5, pad
        cnt = np.zeros(n_reads, np.uint32)
        p = params if isinstance(params, SegParams) else SegParams(*[int(x) for x in params])
        self._check(self.lib.svx_segments_postpass(self.h, _ptr(raw), _ptr(read_off), n_reads, _ptr(contig_rank),
                                                   len(contig_rank), C.byref(p), _ptr(out), _ptr(out_off), _ptr(cnt)))
        # compact: records of read r at packed[first[r] : first[r + 1]]
        first = np.concatenate(([0], np.cumsum(cnt.astype(np.int64))))
        if int(first[-1]):
            take = np.repeat(out_off[:-1].astype(np.int64) - first[:-1], cnt) + np.arange(int(first[-1]))
            packed = out[take]
        else:
            packed = out[:0]
        return packed, first

    # ---------------------------------------------------------------- COLLECT of a sample in one go
    def collect_batch(self, cigar_parts, aln_off, ref_start, min_len, extra_cigar, extra_off, seg_src, seg_tid,
                      seg_pos, seg_rev, seg_qend, read_off, contig_rank, params, part_dev=None):
        """a1 + a2 + a3 of one sample (or of both haplotypes of one): see collect_batch_composed for the
        contract.  One submission on the context's stream (svx_collect_batch).  `part_dev`: per part None or
        (device address, hipEvent_t or None) of a copy of the part that is in HBM already
        (BamFile.device_pool(): svx_collect_in.part_dev / part_ready) — that part is not uploaded again."""
        parts = [np.ascontiguousarray(p, dtype=np.uint32) for p in cigar_parts]
        aln_off = _as(aln_off, np.uint64)
        ref_start = _as(ref_start, np.int32)
        n_aln = len(aln_off) - 1 if len(aln_off) else 0
        extra_cigar, extra_off = _as(extra_cigar, np.uint32), _as(extra_off, np.uint64)
        n_extra = len(extra_off) - 1 if len(extra_off) else 0
        seg_src, seg_tid, seg_pos = _as(seg_src, np.uint32), _as(seg_tid, np.int32), _as(seg_pos, np.int32)
        seg_rev, seg_qend = _as(seg_rev, np.uint8), _as(seg_qend, np.int32)
        read_off = _as(read_off, np.uint32)
        contig_rank = _as(contig_rank, np.int32)
        n_segs = len(seg_src)
        n_reads = len(read_off) - 1 if len(read_off) and n_segs else 0
        prm = params if isinstance(params, SegParams) else SegParams(*[int(x) for x in params])
        n_ops = int(aln_off[-1]) if n_aln else 0
        part_ops = np.array([len(p) for p in parts], dtype=np.uint64)
        part_ptrs = (C.c_void_p * max(1, len(parts)))(*[p.ctypes.data if p.size else None for p in parts])
        slots = np.diff(read_off[:n_reads + 1].astype(np.int64)) if n_reads else np.zeros(0, np.int64)
        post_off = np.zeros(n_reads + 1, np.uint64)
        if n_reads:
            np.cumsum(slots * (slots + 3) // 2, out=post_off[1:])
        raw = np.zeros(n_segs, dtype=RAW_DTYPE)
        post = np.zeros(int(post_off[-1]), dtype=RAW_DTYPE)  # svx_post has svx_raw's layout
        cnt = np.zeros(n_reads, np.uint32)
        cap = max(1024, n_ops // 16)
        arg = CollectIn(cigar_parts=part_ptrs, part_ops=_ptr(part_ops), n_parts=len(parts), aln_off=_ptr(aln_off),
                        ref_start=_ptr(ref_start), n_aln=n_aln, min_len=int(min_len), extra_cigar=_ptr(extra_cigar),
                        extra_off=_ptr(extra_off) if n_extra else None, n_extra=n_extra, seg_src=_ptr(seg_src),
                        seg_tid=_ptr(seg_tid), seg_pos=_ptr(seg_pos), seg_rev=_ptr(seg_rev), seg_qend=_ptr(seg_qend),
                        n_segs=n_segs if n_reads else 0, read_off=_ptr(read_off), n_reads=n_reads,
                        contig_rank=_ptr(contig_rank), n_contigs=len(contig_rank), params=prm)
        if part_dev is not None and any(d is not None and d[0] for d in part_dev):
            if len(part_dev) != len(parts):
                raise ValueError("part_dev: one entry per CIGAR part")
            dev = [d if d is not None and d[0] else (None, None) for d in part_dev]
            arg.part_dev = (C.c_void_p * len(parts))(*[d[0] for d in dev])
            arg.part_ready = (C.c_void_p * len(parts))(*[d[1] for d in dev])
        while True:
            sig = {"aln": np.empty(cap, np.uint32), "ref_pos": np.empty(cap, np.uint32), "read_pos": np.empty(cap, np.uint32),
                   "len": np.empty(cap, np.uint32), "type": np.empty(cap, np.uint8)}
            res = CollectOut(sig=SigSoa(*[_ptr(sig[k]) for k in ("aln", "ref_pos", "read_pos", "len", "type")]),
                             sig_cap=cap, n_sig=0, raw=_ptr(raw), post=_ptr(post), post_off=_ptr(post_off),
                             post_cnt=_ptr(cnt))
            rc = self.lib.svx_collect_batch(self.h, C.byref(arg), C.byref(res))
            if rc == SVX_E_CAPACITY and int(res.n_sig) > cap:
                cap = int(res.n_sig)
                continue
            self._check(rc)
            break
        k = int(res.n_sig)
        sig = {key: v[:k] for key, v in sig.items()}
        first = np.zeros(n_reads + 1, np.int64)
        if n_reads:
            np.cumsum(cnt.astype(np.int64), out=first[1:])
        if int(first[-1]):
            take = np.repeat(post_off[:-1].astype(np.int64) - first[:-1], cnt) + np.arange(int(first[-1]))
            packed = post[take]
        else:
            packed = post[:0]
        return sig, raw, packed, first

    def collect_batch_composed(self, cigar_parts, aln_off, ref_start, min_len, extra_cigar, extra_off, seg_src,
                               seg_tid, seg_pos, seg_rev, seg_qend, read_off, contig_rank, params, part_dev=None):
        """(`part_dev` is accepted for collect_batch's signature and not used: the single-purpose calls upload.)
        The arithmetic of analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83) for a whole batch,
        composed from the single-purpose entry points (one call each):
          cigar_parts   BAM-native CIGAR pools, logically back to back; aln_off[n_aln + 1] / ref_start[n_aln]
                        describe EVERY record of the pools (records the filters drop are masked by the caller)
          extra_cigar / extra_off   CIGARs of the SA-derived segments (SVIM_COLLECT.py:33-55)
          seg_*         one row per segment of every chimeric read, [primary] + supplementaries per read
                        (SVIM_inter.py:64): seg_src < n_aln names a record of the pools, otherwise extra
                        alignment seg_src - n_aln; seg_qend >= 0 overrides query_alignment_end (pysam takes it
                        from the stored sequence when there is one)
          read_off      first segment of each read (n_reads + 1)
        Returns (sig, raw, post, post_first): indel signatures of all records, the adjacency records of
        svx_segments_classify per segment slot, the packed derived records and their first index per read."""
        cigar = cigar_parts[0] if len(cigar_parts) == 1 else (
            np.concatenate(cigar_parts) if len(cigar_parts) else np.zeros(0, np.uint32))
        sig = self.cigar_extract(cigar, aln_off, ref_start, min_len)
        n_aln = len(aln_off) - 1
        n_reads = len(read_off) - 1 if len(read_off) else 0
        if n_reads <= 0 or len(seg_src) == 0:
            return sig, np.zeros(0, dtype=RAW_DTYPE), np.zeros(0, dtype=RAW_DTYPE), np.zeros(max(n_reads, 0) + 1, np.int64)
        st_a = self.cigar_stats(cigar, aln_off)
        st_x = self.cigar_stats(extra_cigar, extra_off)
        src = np.asarray(seg_src, dtype=np.int64)
        st = {k: np.concatenate((st_a[k], st_x[k])).astype(np.int64)[src] for k in ("ref_len", "q_start", "q_end", "read_len")}
        segs, read_len = segment_rows(st, seg_tid, seg_pos, seg_rev, seg_qend, read_off)
        prm = params if isinstance(params, SegParams) else SegParams(*[int(x) for x in params])
        raw = self.segments_classify(segs, read_off, read_len, prm)
        post, first = self.segments_postpass(raw, read_off, contig_rank, prm)
        return sig, raw, post, first

    # ---------------------------------------------------------------- a5 + a6
    def pair_partition(self, keys, max_dist):
        keys = _as(keys, np.uint64)
        n = len(keys)
        perm = np.empty(n, np.uint32)
        part = np.empty(n, np.uint32)
        n_parts = C.c_uint32(0)
        self._check(self.lib.svx_pair_partition(self.h, _ptr(keys), n, int(max_dist), _ptr(perm),
                                                _ptr(part), C.byref(n_parts)))
        return perm, part, int(n_parts.value)

    # ---------------------------------------------------------------- a7
    def edit_distance_batch(self, seq, a_off, a_len, b_off, b_len, k_max=0xFFFFFFFF):
        seq = _as(seq, np.uint8)
        a_off = _as(a_off, np.uint64)
        b_off = _as(b_off, np.uint64)
        a_len = _as(a_len, np.uint32)
        b_len = _as(b_len, np.uint32)
        n = len(a_off)
        dist = np.zeros(n, np.uint32)
        if n:
            self._check(self.lib.svx_edit_distance_batch(self.h, _ptr(seq), len(seq), _ptr(a_off),
                                                         _ptr(a_len), _ptr(b_off), _ptr(b_len), n,
                                                         int(k_max) & 0xFFFFFFFF, _ptr(dist)))
        return dist

    def resident(self, host_bytes):
        """`host_bytes` (uint8 array) uploaded once into HBM: pass the result as `pool` to several
        haplotype_distance_batch calls (svx_haplotype_distance_batch_dev) instead of staging it per call."""
        return DeviceArray(self, host=_as(host_bytes, np.uint8))

    def haplotype_distance_batch(self, pool, pieces, k_max=0xFFFFFFFF):
        """Edit distances of haplotype pairs assembled on the device from pieces of `pool` (bytes: a uint8 array,
        or the DeviceArray `resident()` returned): `pieces` is a HAP_PIECE_DTYPE array with 6 entries per pair
        (3 of haplotype a, 3 of b)."""
        pieces = np.ascontiguousarray(pieces, dtype=HAP_PIECE_DTYPE)
        if len(pieces) % 6:
            raise SvxError(SVX_E_INVALID, "6 pieces per pair")
        n = len(pieces) // 6
        dist = np.zeros(n, np.uint32)
        if not n:
            return dist
        if isinstance(pool, DeviceArray):
            self._check(self.lib.svx_haplotype_distance_batch_dev(self.h, pool.ptr, pool.nbytes, _ptr(pieces), n,
                                                                  int(k_max) & 0xFFFFFFFF, _ptr(dist)))
        else:
            pool = _as(pool, np.uint8)
            self._check(self.lib.svx_haplotype_distance_batch(self.h, _ptr(pool), len(pool), _ptr(pieces), n,
                                                              int(k_max) & 0xFFFFFFFF, _ptr(dist)))
        return dist

    def haplotype_distance_batch_mixed(self, pool, pieces, k_max):
        """haplotype_distance_batch with one threshold per pair (0xFFFFFFFF: exact) — a PAIR step's thresholded and
        exact pairs in ONE call (svx_haplotype_distance_batch_mixed)."""
        pieces = np.ascontiguousarray(pieces, dtype=HAP_PIECE_DTYPE)
        k_max = np.ascontiguousarray(k_max, dtype=np.uint32)
        if len(pieces) != 6 * len(k_max):
            raise SvxError(SVX_E_INVALID, "6 pieces and one threshold per pair")
        dist = np.zeros(len(k_max), np.uint32)
        if len(k_max):
            pool = _as(pool, np.uint8)
            self._check(self.lib.svx_haplotype_distance_batch_mixed(self.h, _ptr(pool), len(pool), _ptr(pieces), len(k_max),
                                                                    _ptr(k_max), _ptr(dist)))
        return dist

    def linkage_cut_batch(self, dist, n_members, cutoff):
        """scipy fcluster(linkage(y, "complete"), cutoff, "distance") for many partitions at once.
        dist: condensed vectors back to back (float64); n_members: partition sizes.  Returns the
        1-based labels, partition after partition (scipy's label order)."""
        dist = _as(dist, np.float64)
        n_members = _as(n_members, np.uint32)
        if int((n_members.astype(np.int64) * (n_members.astype(np.int64) - 1) // 2).sum()) != len(dist):
            raise SvxError(SVX_E_INVALID, "dist length does not match the partition sizes")
        labels = np.zeros(int(n_members.sum()), np.uint32)
        if len(n_members):
            self._check(self.lib.svx_linkage_cut_batch(self.h, _ptr(dist), _ptr(n_members), len(n_members),
                                                       float(cutoff), _ptr(labels)))
        return labels


def segment_rows(st, seg_tid, seg_pos, seg_rev, seg_qend, read_off):
    """Segment rows of SVIM_inter.py:66-81 from per-segment CIGAR statistics (int64 columns ref_len, q_start,
    q_end, read_len): (SEG_DTYPE array, read_len of every read's primary)."""
    n = len(seg_tid)
    seg_pos = np.asarray(seg_pos, dtype=np.int64)
    rev = np.asarray(seg_rev).astype(bool)
    qend = np.asarray(seg_qend, dtype=np.int64)
    q_end = np.where(qend >= 0, qend, st["q_end"])
    segs = np.zeros(n, dtype=SEG_DTYPE)
    segs["q_start"] = np.where(rev, st["read_len"] - q_end, st["q_start"]).astype(np.int32)
    segs["q_end"] = np.where(rev, st["read_len"] - st["q_start"], q_end).astype(np.int32)
    segs["ref_id"] = seg_tid
    segs["ref_start"] = seg_pos
    segs["ref_end"] = (seg_pos + np.where(st["ref_len"] > 0, st["ref_len"], 1)).astype(np.int32)  # htslib bam_endpos
    segs["is_reverse"] = rev
    first = np.asarray(read_off, dtype=np.int64)[:-1]
    return segs, st["read_len"][first].astype(np.int32)


class DeviceArray:
    """A hipMalloc'ed buffer for the *_dev entry points (no torch needed): .ptr is the device address."""

    def __init__(self, ctx, host=None, nbytes=None):
        self.ctx = ctx
        if host is not None:
            host = np.ascontiguousarray(host)
            nbytes = host.nbytes
        self.nbytes = int(nbytes)
        p = _P()
        ctx._check(ctx.lib.svx_dev_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        if host is not None and self.nbytes:
            ctx._check(ctx.lib.svx_dev_upload(ctx.h, self.ptr, host.ctypes.data, self.nbytes))
            ctx.sync()

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(n, dtype)
        self.ctx._check(self.ctx.lib.svx_dev_download(self.ctx.h, out.ctypes.data, self.ptr, n * dtype.itemsize))
        return out

    def free(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.svx_dev_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default_ctx = {}
_default_ctx_lock = threading.Lock()


def new_context(device=0):
    """A context of its own (stream, workspace) on `device` — for a thread that works beside the process-wide one
    (svim-asm-cohort's workers)."""
    return Context(device)


def default_context(device=0):
    """Process-wide context per device (created on first use; raises without a GPU)."""
    ctx = _default_ctx.get(device)
    if ctx is None:
        with _default_ctx_lock:
            ctx = _default_ctx.get(device)
            if ctx is None:
                ctx = Context(device)
                _default_ctx[device] = ctx
    return ctx

