"""Indexed FASTA access (.fai) for the host side: the subset of pysam.FastaFile the reference
uses (svim-asm:124; SVIM_COMBINE.py:45-99,467; SVCandidate.py:57-58,105,155,210,301-302):
0-based half-open `fetch`, `get_reference_length`, `close`, and the two error conditions
main() distinguishes (missing file → IOError, missing index → ValueError)."""
import mmap
import os

import numpy as np


class FastaFile(object):
    def __init__(self, path):
        if not os.path.exists(path):
            raise IOError("file `%s` not found" % path)
        if not os.path.exists(path + ".fai"):
            raise ValueError("no index (.fai) for %s" % path)
        self.filename = path
        self._idx = {}
        self.references, self.lengths = [], []
        with open(path + ".fai") as fh:
            for line in fh:
                f = line.rstrip("\n").split("\t")
                if len(f) < 5:
                    continue
                self._idx[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
                self.references.append(f[0])
                self.lengths.append(int(f[1]))
        self._fh = open(path, "rb")
        # fetches are tens of thousands of short windows (haplotype flanks, VCF alleles): slices of a
        # read-only mapping instead of a seek + read pair each
        try:
            self._map = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(path) else None
        except (OSError, ValueError):
            self._map = None
        self._native = None  # svx_fasta handle (libsvx.so), opened by the first batch fetch

    def _handle(self):
        if getattr(self, "_closed", False):
            raise ValueError("I/O operation on closed file")
        if self._native is None:
            import ctypes as C
            from svim_asm_amd import _lib
            lib = _lib.load()
            rows = [self._idx[name] for name in self.references]
            cols = [np.array([r[k] for r in rows], dtype=dt) for k, dt in ((0, np.int64), (1, np.int64), (2, np.int32), (3, np.int32))]
            h, err = C.c_void_p(), C.create_string_buffer(256)
            rc = lib.svx_fasta_open(os.fsencode(self.filename), len(rows), cols[0].ctypes.data, cols[1].ctypes.data,
                                    cols[2].ctypes.data, cols[3].ctypes.data, C.byref(h), err, len(err))
            if rc != 0:
                raise IOError(err.value.decode(errors="replace"))
            self._native = (lib, h)
            self._ref_index = {}
            for i, name in enumerate(self.references):
                self._ref_index.setdefault(name, i)
        return self._native

    def fetch_batch(self, contigs, start, end, upper=True, ids=None):
        """fetch(contigs[i], start[i], end[i]) for all i in one native call (threads, no str objects):
        (uint8 pool, int64 offsets [n + 1]); `upper` applies str.upper() to every slice.  With `ids`, interval i
        lies on contigs[ids[i]] (a table's contig names and its id column: names are looked up once each)."""
        lib, h = self._handle()
        if ids is not None:
            ids = np.asarray(ids, dtype=np.int64)
            n = len(ids)
            used = np.unique(ids) if n else ids
            table = np.full(len(contigs), -1, dtype=np.int32)
            for c in used.tolist():
                table[c] = self._ref_index[contigs[c]]   # KeyError for a contig the FASTA does not have, as fetch()
            ref = np.ascontiguousarray(table[ids])
        else:
            n = len(contigs)
            ref = np.fromiter((self._ref_index[c] for c in contigs), dtype=np.int32, count=n)
        start = np.ascontiguousarray(start, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        if n and (bool((start < 0).any()) or bool((end < start).any())):
            raise ValueError("fetch coordinates out of range")
        length = np.asarray(self.lengths, dtype=np.int64)[ref] if n else np.zeros(0, np.int64)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.maximum(np.minimum(end, length) - start, 0), out=off[1:])
        out = np.empty(int(off[-1]), dtype=np.uint8)
        rc = lib.svx_fasta_fetch_batch(h, ref.ctypes.data, start.ctypes.data, end.ctypes.data, n, 1 if upper else 0,
                                       off.ctypes.data, out.ctypes.data, 0)
        if rc != 0:
            raise ValueError("reference windows shorter than the index says (%s)" % self.filename)
        return out, off.astype(np.int64)

    def get_reference_length(self, name):
        return self._idx[name][0]

    def fetch(self, reference, start=None, end=None):
        return self.fetch_bytes(reference, start, end).decode("ascii")

    def fetch_bytes(self, reference, start=None, end=None):
        """fetch() without the str round trip (the GPU path uploads reference windows as bytes)."""
        if getattr(self, "_closed", False):
            raise ValueError("I/O operation on closed file")
        length, offset, line_bases, line_width = self._idx[reference]
        start = 0 if start is None else start
        end = length if end is None else end
        if start < 0:
            raise ValueError("start out of range (%i)" % start)
        if end < start:
            raise ValueError("end out of range (%i)" % end)
        end = min(end, length)
        if start >= end:
            return b""
        b0 = offset + (start // line_bases) * line_width + start % line_bases
        b1 = offset + ((end - 1) // line_bases) * line_width + (end - 1) % line_bases + 1
        if self._map is not None:
            raw = self._map[b0:b1]
        else:
            self._fh.seek(b0)
            raw = self._fh.read(b1 - b0)
        if line_width != line_bases:
            raw = raw.replace(b"\n", b"").replace(b"\r", b"")
        return raw

    def close(self):
        """The object is closed at once (fetches fail from here on); its mappings are released by release_deferred().
        Unmapping a genome-sized file whose pages were touched all over is tens of milliseconds of page-table work
        under the process's mapping lock (57 ms for the 3.1 GB of the full-size sample) — in the middle of
        write_final_vcf, where the reference closes its FastaFile (SVIM_COMBINE.py:466-467), it would stall the threads
        that format the record lines; a command that exits right after the VCF never needs it at all."""
        self._closed = True
        if getattr(self, "_native", None) is not None:
            _DEFERRED.append(("native",) + tuple(self._native))
            self._native = None
        if getattr(self, "_map", None) is not None:
            _DEFERRED.append(("map", self._map))
            self._map = None
        if self._fh:
            _DEFERRED.append(("file", self._fh))
            self._fh = None
        if len(_DEFERRED) > 12:  # a caller that opens and closes many genomes and never writes a VCF: bounded
            release_deferred()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass


_DEFERRED = []  # mappings and descriptors of closed FastaFile objects, not yet given back


def release_deferred(background=True):
    """Give back what closed FastaFile objects held (write_vcf_table calls this behind its last write; a long-lived
    caller may call it any time).  `background`: on a daemon thread, beside whatever the caller does next."""
    todo = _DEFERRED[:]
    del _DEFERRED[:len(todo)]
    if not todo:
        return

    def work():
        for item in todo:
            try:
                if item[0] == "native":
                    item[1].svx_fasta_close(item[2])
                else:
                    item[1].close()
            except Exception:  # noqa: BLE001 — nothing left to report to
                pass
    if background:
        import threading
        threading.Thread(target=work, daemon=True).start()
    else:
        work()


def write_fasta(path, names, seqs, line=60):
    """Write FASTA + .fai; `seqs` are ASCII byte strings / numpy uint8 arrays."""
    with open(path, "wb") as fh, open(path + ".fai", "w") as fai:
        for name, seq in zip(names, seqs):
            a = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
            hdr = (">%s\n" % name).encode()
            fh.write(hdr)
            off = fh.tell()
            n = len(a)
            full = n // line
            if full:
                body = np.empty((full, line + 1), dtype=np.uint8)
                body[:, :line] = a[:full * line].reshape(full, line)
                body[:, line] = 10
                fh.write(body.tobytes())
            if n % line:
                fh.write(a[full * line:].tobytes() + b"\n")
            fai.write("%s\t%d\t%d\t%d\t%d\n" % (name, n, off, line, line + 1))
