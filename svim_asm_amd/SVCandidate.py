"""SV candidate data model: the attribute / method surface of the reference's Candidate*
classes (SVCandidate.py:1-443) consumed by main() (svim-asm:133-148), pair_candidates
(SVIM_COMBINE.py:164-366), write_final_vcf (:431-464) and plot_sv_lengths.

Semantics restated from SURVEY.md A2/A5: constructors assert `end >= start` then clamp to
[0, contig length] (BND: each position clamped, endpoints ordered lexicographically with the
directions flipped when swapped); get_key() is (type, contig, position) with the midpoint
for DEL/INV/DUP_TAN, the destination start for INS/DUP_INT and the source start for BND.
VCF lines share one formatter instead of six copies.
"""

_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}
_FLIP = {"fwd": "rev", "rev": "fwd"}


def _vcf_line(chrom, pos, ref, alt, filters, info, fmt, sample):
    return "\t".join((chrom, str(pos), "PLACEHOLDERFORID", ref, alt, ".",
                      ";".join(filters) if filters else "PASS", info, fmt, sample))


def _revcomp_upper(seq):
    return "".join(_COMPLEMENT.get(b.upper(), b.upper()) for b in reversed(seq))


class Candidate(object):
    """Common behaviour; `source_*` describe the affected reference interval."""
    type = None

    def __init__(self, source_contig, source_start, source_end, genotype="1/1"):
        self.source_contig = source_contig
        self.source_start = source_start
        self.source_end = source_end
        self.genotype = genotype

    def _set_source(self, contig, start, end, bam, what, reads):
        assert end >= start, "{0} end ({1}:{2}) is smaller than its start ({1}:{3}). From read {4}".format(
            what, contig, end, start, reads)
        self.source_contig = contig
        self.source_start = max(0, start)
        self.source_end = min(bam.get_reference_length(contig), end)

    def _set_dest(self, contig, start, end, bam, what, reads):
        assert end >= start, "{0} end ({1}:{2}) is smaller than its start ({1}:{3}). From read {4}".format(
            what, contig, end, start, reads)
        self.dest_contig = contig
        self.dest_start = max(0, start)
        self.dest_end = min(bam.get_reference_length(contig), end)

    def get_source(self):
        return (self.source_contig, self.source_start, self.source_end)

    def get_key(self):
        return (self.type, self.source_contig, (self.source_start + self.source_end) // 2)

    def position_distance_to(self, other):
        c1, s1, e1 = self.get_source()
        c2, s2, e2 = other.get_source()
        if self.type == other.type and c1 == c2:
            return min(abs(s1 - s2), abs(e1 - e2), abs((s1 + e1) // 2 - (s2 + e2) // 2))
        return float("inf")

    def _reads_info(self, read_names):
        return ";READS={0}".format(",".join(self.reads)) if read_names else ""

    def get_vcf_entry(self):
        raise NotImplementedError


class CandidateDeletion(Candidate):
    type = "DEL"

    def __init__(self, source_contig, source_start, source_end, reads, bam, genotype="1/1"):
        self._set_source(source_contig, source_start, source_end, bam, "Deletion", reads)
        self.reads = reads
        self.genotype = genotype

    def get_vcf_entry(self, sequence_alleles=False, reference=None, read_names=False):
        contig, start, end = self.source_contig, self.source_start, self.source_end
        if sequence_alleles:
            first = start - 1 if start > 0 else 0
            ref = reference.fetch(contig, first, end).upper()
            # reference.fetch(contig, max(0, start - 1), start): the first base of `ref` (nothing when start is 0)
            alt = ref[:start - first]
        else:
            ref, alt = "N", "<DEL>"
        reads = self._reads_info(read_names) if read_names else ""
        return (f"{contig}\t{start if start > 1 else 1}\tPLACEHOLDERFORID\t{ref}\t{alt}\t.\tPASS\t"
                f"SVTYPE=DEL;END={end};SVLEN={start - end}{reads}\tGT\t{self.genotype}")


class CandidateInversion(Candidate):
    type = "INV"

    def __init__(self, source_contig, source_start, source_end, reads, complete, bam, genotype="1/1"):
        self._set_source(source_contig, source_start, source_end, bam, "Inversion", reads)
        self.reads = reads
        self.complete = complete
        self.genotype = genotype

    def get_vcf_entry(self, sequence_alleles=False, reference=None, read_names=False):
        contig, start, end = self.get_source()
        if sequence_alleles:
            ref = reference.fetch(contig, start, end).upper()
            alt = _revcomp_upper(ref)
        else:
            ref, alt = "N", "<INV>"
        info = "SVTYPE=INV;END={0}".format(end) + self._reads_info(read_names)
        return _vcf_line(contig, start + 1, ref, alt, [] if self.complete else ["incomplete_inversion"],
                         info, "GT", self.genotype)


class CandidateInsertion(Candidate):
    type = "INS"

    def __init__(self, dest_contig, dest_start, dest_end, reads, sequence, bam, genotype="1/1"):
        self._set_dest(dest_contig, dest_start, dest_end, bam, "Insertion", reads)
        self.reads = reads
        self.sequence = sequence
        self.genotype = genotype

    def get_destination(self):
        return (self.dest_contig, self.dest_start, self.dest_end)

    def get_key(self):
        return (self.type, self.dest_contig, self.dest_start)

    def get_vcf_entry(self, sequence_alleles=False, reference=None, read_names=False):
        contig, start, end = self.dest_contig, self.dest_start, self.dest_end
        if sequence_alleles:
            ref = reference.fetch(contig, start - 1 if start > 0 else 0, start).upper()
            alt = ref + self.sequence
        else:
            ref, alt = "N", "<INS>"
        reads = self._reads_info(read_names) if read_names else ""
        return (f"{contig}\t{start if start > 1 else 1}\tPLACEHOLDERFORID\t{ref}\t{alt}\t.\tPASS\t"
                f"SVTYPE=INS;END={start};SVLEN={end - start}{reads}\tGT\t{self.genotype}")


class CandidateDuplicationTandem(Candidate):
    type = "DUP_TAN"

    def __init__(self, source_contig, source_start, source_end, copies, fully_covered, reads, bam,
                 genotype="1/1"):
        self._set_source(source_contig, source_start, source_end, bam, "Tandem duplication", reads)
        self.copies = copies  # number of ADDITIONAL copies
        self.reads = reads
        self.fully_covered = fully_covered
        self.genotype = genotype

    def get_destination(self):
        contig, start, end = self.get_source()
        return (contig, end, end + self.copies * (end - start))

    def _filters(self):
        return [] if self.fully_covered else ["not_fully_covered"]

    def get_vcf_entry_as_ins(self, sequence_alleles=False, reference=None, read_names=False):
        contig, start, end = self.get_source()
        if sequence_alleles:
            ref = reference.fetch(contig, start, end).upper()
            alt = ref * (self.copies + 1)
        else:
            ref, alt = "N", "<INS>"
        info = "SVTYPE=INS;END={0};SVLEN={1}".format(end, (end - start) * self.copies) + self._reads_info(read_names)
        return _vcf_line(contig, start + 1, ref, alt, self._filters(), info, "GT", self.genotype)

    def get_vcf_entry_as_dup(self, read_names=False):
        contig, start, end = self.get_source()
        info = "SVTYPE=DUP:TANDEM;END={0};SVLEN={1}".format(end, end - start) + self._reads_info(read_names)
        return _vcf_line(contig, start + 1, "N", "<DUP:TANDEM>", self._filters(), info, "GT:CN",
                         "{0}:{1}".format(self.genotype, self.copies + 1))


class CandidateDuplicationInterspersed(Candidate):
    type = "DUP_INT"

    def __init__(self, source_contig, source_start, source_end, dest_contig, dest_start, dest_end, reads,
                 bam, cutpaste=False, genotype="1/1"):
        self._set_source(source_contig, source_start, source_end, bam, "Interspersed duplication source", reads)
        self._set_dest(dest_contig, dest_start, dest_end, bam, "Interspersed duplication destination", reads)
        self.cutpaste = cutpaste
        self.reads = reads
        self.genotype = genotype

    def get_destination(self):
        return (self.dest_contig, self.dest_start, self.dest_end)

    def get_key(self):
        return (self.type, self.dest_contig, self.dest_start)

    def _cutpaste_info(self):
        return "CUTPASTE;" if self.cutpaste else ""

    def get_vcf_entry_as_ins(self, sequence_alleles=False, reference=None, read_names=False):
        contig, start, end = self.get_destination()
        if sequence_alleles:
            ref = reference.fetch(contig, max(0, start - 1), start).upper()
            alt = ref + reference.fetch(self.source_contig, self.source_start, self.source_end).upper()
        else:
            ref, alt = "N", "<INS>"
        info = "SVTYPE=INS;{0}END={1};SVLEN={2}".format(self._cutpaste_info(), start, end - start) + \
               self._reads_info(read_names)
        return _vcf_line(contig, max(1, start), ref, alt, [], info, "GT", self.genotype)

    def get_vcf_entry_as_dup(self, read_names=False):
        contig, start, end = self.get_source()
        info = "SVTYPE=DUP:INT;{0}END={1};SVLEN={2}".format(self._cutpaste_info(), end, end - start) + \
               self._reads_info(read_names)
        return _vcf_line(contig, start + 1, "N", "<DUP:INT>", [], info, "GT", self.genotype)


class CandidateBreakend(Candidate):
    type = "BND"

    def __init__(self, source_contig, source_start, source_direction, dest_contig, dest_start, dest_direction,
                 reads, bam, genotype="1/1"):
        a = (source_contig, source_start, source_direction)
        b = (dest_contig, dest_start, dest_direction)
        if not (source_contig < dest_contig or (source_contig == dest_contig and source_start < dest_start)):
            # swapped endpoints read the junction from the other side: both directions flip
            a, b = (b[0], b[1], _FLIP[b[2]]), (a[0], a[1], _FLIP[a[2]])
        self.source_contig, self.source_direction = a[0], a[2]
        self.source_start = min(bam.get_reference_length(a[0]), max(0, a[1]))
        self.dest_contig, self.dest_direction = b[0], b[2]
        self.dest_start = min(bam.get_reference_length(b[0]), max(0, b[1]))
        self.reads = reads
        self.genotype = genotype

    def get_source(self):
        return (self.source_contig, self.source_start)

    def get_destination(self):
        return (self.dest_contig, self.dest_start)

    def get_key(self):
        return (self.type, self.source_contig, self.source_start)

    _ALT_FWD = {("fwd", "fwd"): "N[{0}:{1}[", ("fwd", "rev"): "N]{0}:{1}]",
                ("rev", "rev"): "]{0}:{1}]N", ("rev", "fwd"): "[{0}:{1}[N"}
    _ALT_REV = {("rev", "rev"): "N[{0}:{1}[", ("fwd", "rev"): "N]{0}:{1}]",
                ("fwd", "fwd"): "]{0}:{1}]N", ("rev", "fwd"): "[{0}:{1}[N"}

    def _entry(self, here, mate, table, read_names):
        alt = table[(self.source_direction, self.dest_direction)].format(mate[0], mate[1] + 1)
        info = "SVTYPE=BND" + self._reads_info(read_names)
        return _vcf_line(here[0], here[1] + 1, "N", alt, [], info, "GT", self.genotype)

    def get_vcf_entry(self, read_names=False):
        return self._entry(self.get_source(), self.get_destination(), self._ALT_FWD, read_names)

    def get_vcf_entry_reverse(self, read_names=False):
        return self._entry(self.get_destination(), self.get_source(), self._ALT_REV, read_names)
