"""Pure-Python lazy BGZF reader (the differential reference of the native one): only header/CIGAR/tag blocks are inflated at open; sequence slices and the
parallel prefetch return exactly the bases an eager decode gives.  CPU only."""
import os

import numpy as np

from svim_asm_amd import bamio, synth_bam


def test_lazy_reader_equals_eager_decode(tmp_path):
    contigs = (("a", 400000), ("b", 250000))
    fa, bams = synth_bam.write_dataset(str(tmp_path), seed=9, contigs=contigs, n_shared=10, n_private=2,
                                       median_aln=150000, dense_cluster=False, with_splits=False)
    f = bamio.AlignmentFile(bams[0], reader="python")
    assert len(f) > 0
    n_blocks = len(f._z._start)
    assert f._z.blocks_inflated < n_blocks / 2, "most blocks lie inside SEQ/QUAL and must stay compressed"
    eager = bamio.bgzf_decompress(bams[0])
    assert f._z.size == len(eager)
    rng = np.random.default_rng(0)
    reqs = []
    for i in range(len(f)):
        r = f.record(i)
        for _ in range(5):
            a = int(rng.integers(0, max(1, r._l_seq - 1)))
            b = min(r._l_seq, a + int(rng.integers(1, 3000)))
            reqs.append((i, a, b))
    f.prefetch_sequence(reqs)
    for i, a, b in reqs:
        r = f.record(i)
        off = int(f._cols["seq_off"][i])
        packed = np.frombuffer(eager, dtype=np.uint8, count=(r._l_seq + 1) // 2, offset=off)
        full = bamio._SEQ_PAIR_LUT[packed].reshape(-1)[:r._l_seq].tobytes().decode()
        assert r.seq_slice(a, b) == full[a:b]
    # columnar batch == per-record words
    cig, off, pos, tid = f.batch()
    for i in range(len(f)):
        assert np.array_equal(cig[int(off[i]):int(off[i + 1])], f.record(i).cigar_words)
    sub = [1, 0, len(f) - 1]
    c2, o2, p2, t2 = f.batch(sub)
    assert np.array_equal(c2[int(o2[1]):int(o2[2])], f.record(0).cigar_words) and list(p2) == [int(f._cols["pos"][i]) for i in sub]
