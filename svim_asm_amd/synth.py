"""Deterministic synthetic inputs for the hot path.

`synth_cigar_batch` builds the flattened, BAM-native CIGAR layout the kernels consume
(`len << 4 | op` words, u64 offsets, per-alignment tid / reference_start) for a human-scale
genome-genome alignment: contigs with GRCh38 primary lengths, ~5 000 alignments tiling them,
CIGAR grammar  S? (M (I|D))* M S?  with geometric M runs and a small SV-sized indel tail.
No file I/O here; the BAM writer for end-to-end runs lives in svim_asm_amd/bamio.py.
"""
import numpy as np

# GRCh38 primary assembly, chr1..22, X, Y
GRCH38_LENGTHS = np.array([
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
    138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
    83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415], dtype=np.int64)
GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X, OP_B = range(10)


def _indel_lengths(rng, n, sv_frac, sv_min=40, sv_max=10000):
    """98.5 % uniform 1..39, sv_frac log-uniform sv_min..sv_max."""
    small = rng.integers(1, sv_min, size=n, dtype=np.int64)
    is_sv = rng.random(n) < sv_frac
    big = np.exp(rng.uniform(np.log(sv_min), np.log(sv_max), size=n)).astype(np.int64)
    big = np.clip(big, sv_min, sv_max)
    return np.where(is_sv, big, small)


def synth_cigar_batch(seed=2, genome_len=None, contig_lengths=GRCH38_LENGTHS, median_aln=300_000,
                      sigma_aln=1.2, mean_m=4000, sv_frac=0.015, softclip_frac=0.3,
                      ops_target=None):
    """One haplotype-vs-reference alignment set as flat arrays.

    Returns dict: cigar u32[n_ops], aln_off u64[n_aln+1], ref_start i32[n_aln], tid i32[n_aln],
    contig_lengths i64[C].  mean_m=4000 → ≈1.5 M ops for 3.1 Gbp (config 2); mean_m=400 → ≈15 M
    (config 5).  `ops_target` truncates/extends the genome so that about that many ops come out.
    """
    rng = np.random.default_rng(seed)
    contig_lengths = np.asarray(contig_lengths, dtype=np.int64)
    total = int(contig_lengths.sum()) if genome_len is None else int(genome_len)
    if ops_target is not None:
        total = int(ops_target * (mean_m + 20) / 2)
    # M runs and the indel following each run
    n_pairs = int(total / (mean_m + 20) * 1.02) + 16
    m_len = rng.geometric(1.0 / mean_m, size=n_pairs).astype(np.int64)
    is_del = rng.random(n_pairs) < 0.5
    x_len = _indel_lengths(rng, n_pairs, sv_frac)
    ref_adv = m_len + np.where(is_del, x_len, 0)
    cum = np.cumsum(ref_adv)
    keep = cum <= total
    n_pairs = int(keep.sum())
    m_len, is_del, x_len, cum = m_len[:n_pairs], is_del[:n_pairs], x_len[:n_pairs], cum[:n_pairs]
    # alignment spans: log-normal lengths tiling the genome coordinate
    n_guess = int(total / (median_aln * np.exp(sigma_aln ** 2 / 2)) * 1.5) + 8
    spans = np.exp(rng.normal(np.log(median_aln), sigma_aln, size=n_guess)).astype(np.int64) + 1000
    ends = np.cumsum(spans)
    ends = ends[ends < total]
    ends = np.append(ends, total)
    # pair p belongs to the alignment whose span contains the END of its reference advance
    aln_of_pair = np.searchsorted(ends, cum, side="left")
    n_aln = int(aln_of_pair.max()) + 1 if n_pairs else 0
    k = np.bincount(aln_of_pair, minlength=n_aln).astype(np.int64)  # pairs per alignment
    first_pair = np.concatenate(([0], np.cumsum(k)[:-1]))
    sl = (rng.random(n_aln) < softclip_frac).astype(np.int64)
    st = (rng.random(n_aln) < softclip_frac).astype(np.int64)
    n_ops_aln = sl + 2 * k + 1 + st
    aln_off = np.concatenate(([0], np.cumsum(n_ops_aln))).astype(np.uint64)
    n_ops = int(aln_off[-1])
    cigar = np.zeros(n_ops, dtype=np.uint32)
    off = aln_off[:-1].astype(np.int64)
    # leading / trailing soft clips
    s_len = rng.integers(1, 5000, size=n_aln, dtype=np.int64)
    cigar[off[sl == 1]] = (s_len[sl == 1] << 4 | OP_S).astype(np.uint32)
    t_len = rng.integers(1, 5000, size=n_aln, dtype=np.int64)
    last = off + n_ops_aln - 1
    cigar[last[st == 1]] = (t_len[st == 1] << 4 | OP_S).astype(np.uint32)
    # M / indel pairs
    pos_m = off[aln_of_pair] + sl[aln_of_pair] + 2 * (np.arange(n_pairs) - first_pair[aln_of_pair])
    cigar[pos_m] = (m_len << 4 | OP_M).astype(np.uint32)
    cigar[pos_m + 1] = (x_len << 4 | np.where(is_del, OP_D, OP_I)).astype(np.uint32)
    # closing M run
    f_len = rng.geometric(1.0 / mean_m, size=n_aln).astype(np.int64)
    cigar[off + sl + 2 * k] = (f_len << 4 | OP_M).astype(np.uint32)
    # reference placement: genome coordinate of the first pair → (tid, reference_start)
    starts_g = np.concatenate(([0], ends[:-1]))[:n_aln]
    c_end = np.cumsum(contig_lengths)
    # when a custom genome_len / ops_target exceeds the contig table, wrap around
    starts_w = starts_g % int(c_end[-1])
    tid = np.searchsorted(c_end, starts_w, side="right").astype(np.int32)
    c_start = np.concatenate(([0], c_end[:-1]))
    ref_start = (starts_w - c_start[tid]).astype(np.int32)
    return {"cigar": cigar, "aln_off": aln_off, "ref_start": ref_start, "tid": tid,
            "contig_lengths": contig_lengths}


def concat_batches(batches):
    """Concatenate several haplotype batches into one launch-sized batch (cohort mode)."""
    cig = np.concatenate([b["cigar"] for b in batches])
    offs = [np.zeros(1, np.uint64)]
    base = np.uint64(0)
    for b in batches:
        offs.append(b["aln_off"][1:] + base)
        base = base + b["aln_off"][-1]
    return {"cigar": cig, "aln_off": np.concatenate(offs),
            "ref_start": np.concatenate([b["ref_start"] for b in batches]),
            "tid": np.concatenate([b["tid"] for b in batches]),
            "contig_lengths": batches[0]["contig_lengths"]}


def random_cigar_case(rng, n_aln, max_ops, min_len=40, dense=False, all_ops=True):
    """Small adversarial batches for parity tests: every op code 0..15, empty alignments,
    lengths straddling min_len, optional all-indel (dense) CIGARs."""
    n_ops_aln = rng.integers(0, max_ops + 1, size=n_aln)
    if n_aln > 3:
        n_ops_aln[rng.integers(0, n_aln, size=max(1, n_aln // 10))] = 0  # empty alignments
    aln_off = np.concatenate(([0], np.cumsum(n_ops_aln))).astype(np.uint64)
    n = int(aln_off[-1])
    if dense:
        ops = rng.integers(1, 3, size=n)
        lens = rng.integers(min_len, min_len + 50, size=n)
    else:
        ops = rng.integers(0, 16 if all_ops else 9, size=n)
        lens = np.where(rng.random(n) < 0.3, rng.integers(max(1, min_len - 3), min_len + 4, size=n),
                        rng.integers(0, 5000, size=n))
    cigar = (lens.astype(np.uint32) << 4) | ops.astype(np.uint32)
    ref_start = rng.integers(0, 1 << 28, size=n_aln).astype(np.int32)
    return cigar.astype(np.uint32), aln_off, ref_start
