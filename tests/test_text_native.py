"""Native text side of libsvx.so (include/svx_text.h), CPU: svx_fasta_fetch_batch against the per-call Python fetch of
svim_asm_amd/fasta.py (pysam.FastaFile.fetch semantics: 0-based half-open, end clipped to the sequence, line ends skipped)
and the argument validation of svx_vcf_format."""
import ctypes as C

import numpy as np
import pytest

from svim_asm_amd import _lib, SVIM_COMBINE
from svim_asm_amd.fasta import FastaFile, write_fasta
from svim_asm_amd.table import CandidateTable, NamePool, T_DEL, T_INS
from tests import helpers


@pytest.fixture(scope="module")
def fasta(tmp_path_factory):
    d = tmp_path_factory.mktemp("fa")
    rng = np.random.default_rng(1)
    names = ["chr1", "chr2", "tiny", "empty_tail"]
    seqs = [np.frombuffer(b"ACGTacgtNn", np.uint8)[rng.integers(0, 10, n)] for n in (12345, 7000, 7, 60)]
    path = str(d / "ref.fa")
    write_fasta(path, names, seqs, line=60)
    # a second file with CR LF line ends and another line length: the .fai's bytes-per-line column carries both
    path2 = str(d / "crlf.fa")
    with open(path2, "wb") as fh, open(path2 + ".fai", "w") as fai:
        for name, seq in zip(names, seqs):
            fh.write((">%s\r\n" % name).encode())
            off = fh.tell()
            for i in range(0, len(seq), 50):
                fh.write(seq[i:i + 50].tobytes() + b"\r\n")
            fai.write("%s\t%d\t%d\t50\t52\n" % (name, len(seq), off))
    return names, seqs, path, path2


@pytest.mark.parametrize("which", [0, 1])
def test_fetch_batch_equals_single_fetches(fasta, which):
    names, seqs, *paths = fasta
    fa = FastaFile(paths[which])
    rng = np.random.default_rng(2 + which)
    contigs, lo, hi = [], [], []
    for _ in range(3000):
        k = int(rng.integers(0, len(names)))
        n = len(seqs[k])
        a = int(rng.integers(0, n + 20))
        b = a + int(rng.choice([0, 1, 2, 49, 50, 51, 59, 60, 61, 119, 120, 121, 500, 20000]))
        contigs.append(names[k]); lo.append(a); hi.append(b)
    for upper in (False, True):
        pool, off = fa.fetch_batch(contigs, lo, hi, upper=upper)
        for i in range(len(contigs)):
            k = names.index(contigs[i])
            exp = seqs[k][lo[i]:hi[i]].tobytes()
            assert pool[off[i]:off[i + 1]].tobytes() == (exp.upper() if upper else exp), (contigs[i], lo[i], hi[i])
    # the same bytes as the per-call path (which the duck-typed references of the tests use)
    assert fa.fetch("chr1", 55, 130) == seqs[0][55:130].tobytes().decode()
    with pytest.raises(ValueError):
        fa.fetch_batch(["chr1"], [-1], [5])
    with pytest.raises(ValueError):
        fa.fetch_batch(["chr1"], [10], [5])
    with pytest.raises(KeyError):
        fa.fetch_batch(["nope"], [0], [5])
    fa.close()


def test_fetch_batch_refuses_a_file_shorter_than_its_index(fasta, tmp_path):
    names, seqs, path, _ = fasta
    cut = str(tmp_path / "cut.fa")
    data = open(path, "rb").read()
    open(cut, "wb").write(data[:len(data) // 2])
    open(cut + ".fai", "w").write(open(path + ".fai").read())
    fa = FastaFile(cut)
    with pytest.raises(ValueError):
        fa.fetch_batch(["empty_tail"], [0], [60])
    fa.close()


def _small_table():
    t = CandidateTable(["chr1", "chr2"], [1000, 1000], 2, names=NamePool.from_strings(["a", "b"]),
                       seqs=np.frombuffer(b"ACGT", np.uint8))
    t.type[:] = [T_DEL, T_INS]
    t.sc[0], t.ss[0], t.se[0] = 0, 10, 20
    t.dc[1], t.ds[1], t.de[1], t.q_off[1], t.q_len[1] = 1, 30, 34, 0, 4
    t.r_off, t.r_flat = np.arange(3, dtype=np.int64), np.arange(2, dtype=np.int64)
    return t


def test_vcf_format_validates_what_it_indexes_with():
    import argparse
    seqs = {"chr1": "A" * 1000, "chr2": "C" * 1000}
    o = argparse.Namespace(symbolic_alleles=False, query_names=True, tandem_duplications_as_insertions=False,
                           interspersed_duplications_as_insertions=False)
    types = ["DEL", "INS"]
    good = SVIM_COMBINE.vcf_body(_small_table(), types, helpers.FakeFasta(seqs), o).decode().split("\n")
    assert good[0].startswith("chr1\t10\tsvim_asm.DEL.1\t" + "A" * 11 + "\tA\t") and "READS=a" in good[0]
    assert good[1].startswith("chr2\t30\tsvim_asm.INS.1\tC\tCACGT\t") and good[2] == ""
    for damage in ("seq slice", "read name", "genotype", "contig"):
        t = _small_table()
        if damage == "seq slice":
            t.q_len[1] = 5                # one byte past the sequence pool
        elif damage == "read name":
            t.r_flat[1] = 2               # names hold two entries
        elif damage == "genotype":
            t.gt[0] = 9
        else:
            t.sc[0] = 7
        with pytest.raises((_lib.SvxError, IndexError)):
            SVIM_COMBINE.vcf_body(t, types, helpers.FakeFasta(seqs), o)
