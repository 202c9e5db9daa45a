#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference, read-only) in the
build container.  TEST INFRASTRUCTURE ONLY; cannot run on the GPU box (the reference does not
travel) — the committed outputs under tests/golden/ do.

    python oracle/make_golden.py            # (re)generate every fixture

pysam and edlib are not installed here, so the reference is imported with the stand-in
modules of oracle/refstub/ (htslib/pysam semantics restated from their documentation; edlib
replaced by an exact Levenshtein DP).  What is captured:
  * tests/golden/config1/          BAM + FASTA inputs (written by the build's own generator)
    and the reference's VCFs for `haploid` and `diploid` (##fileDate masked), default and
    alternative output options;
  * tests/golden/functions.json    function-level vectors: analyze_cigar_indel, is_similar,
    retrieve_other_alignments on the reference's chimeric_read*.bam fixtures;
  * tests/golden/pipeline_vectors.json.gz   function-level vectors of the callers either side of the
    kernels: analyze_alignment_file_coordsorted (COLLECT), analyze_read_segments per read,
    form_partitions, pair_candidates — inputs as plain records / candidate tuples, outputs as the
    canonical candidate tuples of the REAL reference's objects.
"""
import importlib.machinery
import importlib.util
import json
import logging
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")


def load_reference():
    """Import the reference package with the stub third-party modules in front."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference not available at " + REF)
    for p in (os.path.join(HERE, "refstub"), REF, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import matplotlib
        matplotlib.use("Agg")
    except Exception:
        pass
    import svim_asm.SVIM_COLLECT as COLLECT
    import svim_asm.SVIM_COMBINE as COMBINE
    import svim_asm.SVIM_inter as INTER
    import svim_asm.SVIM_intra as INTRA
    import svim_asm.SVCandidate as CAND
    import svim_asm.SVIM_input_parsing as PARSE
    return dict(COLLECT=COLLECT, COMBINE=COMBINE, INTER=INTER, INTRA=INTRA, CAND=CAND, PARSE=PARSE)


def run_reference_cli(argv):
    """Run the reference's `svim-asm` script main() with sys.argv = argv."""
    load_reference()
    path = os.path.join(REF, "svim_asm", "svim-asm")
    loader = importlib.machinery.SourceFileLoader("svim_asm_ref_main", path)
    spec = importlib.util.spec_from_loader("svim_asm_ref_main", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    # parse_arguments binds sys.argv[1:] as a default argument at import time: pass argv explicitly
    real_parse = mod.parse_arguments
    mod.parse_arguments = lambda program_version: real_parse(program_version, list(argv))
    old = sys.argv
    root = logging.getLogger()
    handlers = list(root.handlers)
    sys.argv = ["svim-asm"] + list(argv)
    try:
        mod.main()
    finally:
        sys.argv = old
        for h in list(root.handlers):
            if h not in handlers:
                root.removeHandler(h)
                try:
                    h.close()
                except Exception:
                    pass


def masked_vcf(path):
    with open(path) as fh:
        return "".join(line for line in fh if not line.startswith("##fileDate="))


def candidate_tuple(c):
    """Canonical tuple of a reference Candidate* object (same layout as oracle/svim_oracle.py)."""
    t = c.type
    if t == "DEL":
        return ("DEL", c.source_contig, c.source_start, c.source_end, tuple(c.reads), c.genotype)
    if t == "INS":
        return ("INS", c.dest_contig, c.dest_start, c.dest_end, tuple(c.reads), c.sequence, c.genotype)
    if t == "INV":
        return ("INV", c.source_contig, c.source_start, c.source_end, tuple(c.reads), bool(c.complete), c.genotype)
    if t == "DUP_TAN":
        return ("DUP_TAN", c.source_contig, c.source_start, c.source_end, c.copies, bool(c.fully_covered),
                tuple(c.reads), c.genotype)
    if t == "DUP_INT":
        return ("DUP_INT", c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end,
                tuple(c.reads), bool(c.cutpaste), c.genotype)
    return ("BND", c.source_contig, c.source_start, c.source_direction, c.dest_contig, c.dest_start,
            c.dest_direction, tuple(c.reads), c.genotype)


def make_config1():
    from svim_asm_amd import synth_bam
    out = os.path.join(GOLD, "config1")
    if os.path.isdir(out):
        shutil.rmtree(out)
    fasta, bams = synth_bam.write_dataset(out, seed=1)
    runs = {
        "haploid_default": ["haploid", "{wd}", bams[0], fasta],
        "haploid_options": ["haploid", "{wd}", bams[1], fasta, "--min_sv_size", "30", "--query_names",
                            "--tandem_duplications_as_insertions", "--interspersed_duplications_as_insertions",
                            "--sample", "S2"],
        "haploid_symbolic": ["haploid", "{wd}", bams[0], fasta, "--symbolic_alleles", "--types", "DEL,INS,BND",
                             "--max_sv_size", "3000"],
        "diploid_default": ["diploid", "{wd}", bams[0], bams[1], fasta],
        "diploid_options": ["diploid", "{wd}", bams[0], bams[1], fasta, "--query_names", "--max_edit_distance", "20",
                            "--partition_max_distance", "300", "--min_mapq", "0"],
        "diploid_symbolic_dup_as_ins": ["diploid", "{wd}", bams[0], bams[1], fasta, "--symbolic_alleles",
                                        "--tandem_duplications_as_insertions",
                                        "--interspersed_duplications_as_insertions", "--sample", "D2"],
        "diploid_tolerances": ["diploid", "{wd}", bams[0], bams[1], fasta, "--query_gap_tolerance", "0",
                               "--query_overlap_tolerance", "0", "--reference_gap_tolerance", "500",
                               "--reference_overlap_tolerance", "1000", "--min_sv_size", "50"],
        "diploid_strict_pairing": ["diploid", "{wd}", bams[0], bams[1], fasta, "--max_edit_distance", "0",
                                   "--partition_max_distance", "0"],
        "diploid_types_small": ["diploid", "{wd}", bams[1], bams[0], fasta, "--types", "DEL,DUP:TANDEM,INV",
                                "--max_sv_size", "500"],
        "haploid_mapq_overlap": ["haploid", "{wd}", bams[1], fasta, "--min_mapq", "30",
                                 "--reference_overlap_tolerance", "0", "--query_overlap_tolerance", "500"],
    }
    for name, argv in runs.items():
        wd = tempfile.mkdtemp(prefix="svimref_")
        run_reference_cli([a.format(wd=wd) for a in argv])
        with open(os.path.join(out, name + ".vcf"), "w") as fh:
            fh.write(masked_vcf(os.path.join(wd, "variants.vcf")))
        shutil.rmtree(wd)
    with open(os.path.join(out, "runs.json"), "w") as fh:
        rel = {k: [os.path.basename(a) if a.startswith(out) else a for a in v] for k, v in runs.items()}
        json.dump(rel, fh, indent=1)
    return out


MEDIUM = dict(seed=3, scale=0.02, n_shared=40, n_private=8, median_aln=300000, mean_m=2000)


def medium_contigs():
    from svim_asm_amd import synth
    return tuple((n, max(60000, int(l * MEDIUM["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))


def make_medium():
    """62 Mbp diploid sample (24 contigs at 1/50 of GRCh38): too big to commit as BAM, so only the
    reference's VCF and the SHA-256 of the generated inputs are stored; tests regenerate the inputs
    with the same seeds (svim_asm_amd/synth_bam.py is deterministic)."""
    import gzip
    import hashlib
    from svim_asm_amd import synth_bam
    d = tempfile.mkdtemp(prefix="svx_medium_")
    fasta, bams = synth_bam.write_dataset(d, seed=MEDIUM["seed"], contigs=medium_contigs(), n_shared=MEDIUM["n_shared"],
                                          n_private=MEDIUM["n_private"], median_aln=MEDIUM["median_aln"],
                                          mean_m=MEDIUM["mean_m"])
    wd = os.path.join(d, "wd")
    run_reference_cli(["diploid", wd, bams[0], bams[1], fasta])
    vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
    with gzip.GzipFile(os.path.join(GOLD, "medium_diploid.vcf.gz"), "wb", compresslevel=9, mtime=0) as fh:
        fh.write(vcf.encode())
    digest = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in [fasta] + bams}
    with open(os.path.join(GOLD, "medium_inputs.json"), "w") as fh:
        json.dump({"params": MEDIUM, "sha256": digest, "records": sum(1 for l in vcf.split("\n") if l and l[0] != "#")}, fh, indent=1)
    shutil.rmtree(d)


MEDIUM_OPTION_RUNS = {
    # the same 62 Mbp inputs through the option paths that change what the VCF writer and the pairing step do
    "medium_names_dupins": ["--query_names", "--tandem_duplications_as_insertions", "--interspersed_duplications_as_insertions",
                            "--min_sv_size", "30", "--max_edit_distance", "50", "--sample", "M2"],
    "medium_symbolic_strict": ["--symbolic_alleles", "--max_edit_distance", "0", "--partition_max_distance", "100",
                               "--types", "DEL,INS,BND,DUP:TANDEM", "--min_mapq", "0"],
}


def make_medium_options():
    """More goldens on the medium sample: the real reference with non-default options (read names in INFO,
    duplications written as insertions, symbolic alleles, strict pairing, a subset of types)."""
    import gzip
    from svim_asm_amd import synth_bam
    d = tempfile.mkdtemp(prefix="svx_medium_opt_")
    fasta, bams = synth_bam.write_dataset(d, seed=MEDIUM["seed"], contigs=medium_contigs(), n_shared=MEDIUM["n_shared"],
                                          n_private=MEDIUM["n_private"], median_aln=MEDIUM["median_aln"],
                                          mean_m=MEDIUM["mean_m"])
    meta = json.load(open(os.path.join(GOLD, "medium_inputs.json")))
    for f in [fasta] + bams:
        got = synth_bam.payload_digest(f) if f.endswith(".bam") else synth_bam.file_digest(f)
        if got != meta["payload_sha256"][os.path.basename(f)]:
            raise SystemExit("regenerated medium inputs differ from the committed digests")
    runs = {}
    for name, extra in MEDIUM_OPTION_RUNS.items():
        wd = os.path.join(d, "wd_" + name)
        run_reference_cli(["diploid", wd, bams[0], bams[1], fasta] + extra)
        vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
        with gzip.GzipFile(os.path.join(GOLD, name + ".vcf.gz"), "wb", compresslevel=9, mtime=0) as fh:
            fh.write(vcf.encode())
        runs[name] = {"options": extra, "records": sum(1 for l in vcf.split("\n") if l and l[0] != "#")}
    meta["option_runs"] = runs
    with open(os.path.join(GOLD, "medium_inputs.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    shutil.rmtree(d)


LARGE = dict(seed=3, scale=0.25, sv_per_mbp=8.0, median_aln=300000, mean_m=2000)


def large_dataset_args():
    """Arguments of synth_bam.write_dataset for the 772 Mbp diploid sample (config 3 at 1/4 of GRCh38: 24
    contigs, 2 x 231 MB BAM, ~0.79 M CIGAR ops per haplotype) — the generator tools/e2e_bench.py uses."""
    from svim_asm_amd import synth
    contigs = tuple((n, max(60000, int(l * LARGE["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    n_shared = max(4, int(LARGE["sv_per_mbp"] * max(c[1] for c in contigs) / 1e6))
    return dict(seed=LARGE["seed"], contigs=contigs, diploid=True, n_shared=n_shared, n_private=max(2, n_shared // 5),
                median_aln=LARGE["median_aln"], mean_m=LARGE["mean_m"])


def make_large():
    """`svim-asm diploid` of the REAL reference on the 772 Mbp sample.  The VCF is several MB: its SHA-256
    (##fileDate masked), its size, the record counts and the first / last records are committed together
    with the input digests; tests regenerate the inputs from the seeds and compare the digest of their VCF."""
    import hashlib
    from svim_asm_amd import synth_bam
    d = tempfile.mkdtemp(prefix="svx_large_")
    fasta, bams = synth_bam.write_dataset(d, **large_dataset_args())
    wd = os.path.join(d, "wd")
    run_reference_cli(["diploid", wd, bams[0], bams[1], fasta])
    vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
    digest = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in bams}
    body = [l for l in vcf.split("\n") if l and l[0] != "#"]
    kinds = {}
    for l in body:
        k = l.split("\t")[2].rsplit(".", 1)[0]
        kinds[k] = kinds.get(k, 0) + 1
    with open(os.path.join(GOLD, "large_inputs.json"), "w") as fh:
        json.dump({"params": LARGE, "sha256": digest, "records": len(body), "records_by_id_prefix": kinds,
                   "vcf_sha256": hashlib.sha256(vcf.encode()).hexdigest(), "vcf_bytes": len(vcf.encode()),
                   "first_records": [l[:200] for l in body[:3]], "last_records": [l[:200] for l in body[-3:]]}, fh, indent=1)
    shutil.rmtree(d)


FULL = dict(seed=3, scale=1.0, sv_per_mbp=8.0, median_aln=300000, mean_m=2000)


def full_dataset_args(prm=None):
    """synth_bam.write_dataset arguments of the full-size diploid sample (BASELINE config 3 as the metric quotes it:
    GRCh38 contig lengths, 3.1 Gbp, 2 x 924 MB BAM, ~3.2 M CIGAR ops per haplotype) — what bench.py's e2e leg and
    tools/e2e_bench.py generate at --scale 1.0."""
    from svim_asm_amd import synth
    prm = prm or FULL
    contigs = tuple((n, max(60000, int(l * prm["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    n_shared = max(4, int(prm["sv_per_mbp"] * max(c[1] for c in contigs) / 1e6))
    return dict(seed=prm["seed"], contigs=contigs, diploid=True, n_shared=n_shared, n_private=max(2, n_shared // 5),
                median_aln=prm["median_aln"], mean_m=prm["mean_m"])


def _vcf_summary(vcf):
    import hashlib
    body = [l for l in vcf.split("\n") if l and l[0] != "#"]
    kinds = {}
    for l in body:
        k = l.split("\t")[2].rsplit(".", 1)[0]
        kinds[k] = kinds.get(k, 0) + 1
    return {"records": len(body), "records_by_id_prefix": kinds, "vcf_sha256": hashlib.sha256(vcf.encode()).hexdigest(),
            "vcf_bytes": len(vcf.encode()), "first_records": [l[:200] for l in body[:3]],
            "last_records": [l[:200] for l in body[-3:]]}


def make_full(keep=None):
    """`svim-asm diploid` of the REAL reference on the full-size sample (the configuration BASELINE's BAM->VCF
    wall-clock is quoted on).  Committed: digest / size / counts / first and last records of its VCF, the
    reference's own wall-clock in this container, and the digests of the inputs' UNCOMPRESSED content
    (synth_bam.payload_digest — independent of the zlib build) so that bench.py and the tests can tell, on any
    box, that they regenerated the same inputs."""
    import time
    from svim_asm_amd import synth_bam
    keep = keep or os.environ.get("SVX_KEEP_FULL")
    d = keep or tempfile.mkdtemp(prefix="svx_full_")
    t0 = time.time()
    fasta, bams = synth_bam.write_dataset(d, **full_dataset_args())
    t_gen = time.time() - t0
    wd = os.path.join(d, "wd_reference")
    t0 = time.time()
    run_reference_cli(["diploid", wd, bams[0], bams[1], fasta])
    t_ref = time.time() - t0
    vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
    meta = {"params": FULL,
            "payload_sha256": dict([(os.path.basename(f), synth_bam.payload_digest(f)) for f in bams] +
                                   [(os.path.basename(fasta), synth_bam.file_digest(fasta))]),
            "bam_bytes_here": [os.path.getsize(b) for b in bams],
            "reference_wall_s_build_container": round(t_ref, 1), "generate_s_build_container": round(t_gen, 1),
            "reference_note": "the real reference (pure Python, /root/reference) with oracle/refstub pysam and edlib "
                              "(edlib = the C oracle's exact Levenshtein), 1 core of the build container"}
    meta.update(_vcf_summary(vcf))
    with open(os.path.join(GOLD, "full_inputs.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    if not keep:
        shutil.rmtree(d)


CONFIG5 = dict(seed=5, scale=0.05, sv_per_mbp=400.0, median_aln=300000, mean_m=400, min_gap=200)


def config5_dataset_args():
    """BASELINE config 5 as a diploid end-to-end case: 10x the small-indel density (mean M run 400 bp) and SV
    events dense enough (50x config 3's rate, at least 200 bp apart) that neighbouring events chain inside the
    pairing step's 1000 bp: > 131072 candidates (the pair sort takes its radix plan), tens of thousands of
    partitions with 3..10 members (the complete-linkage path) and hundreds with more than 10 (dropped,
    SVIM_COMBINE.py:126-128).  24 contigs at 1/20 of GRCh38 (154 Mbp)."""
    kw = full_dataset_args(CONFIG5)
    kw["min_gap"] = CONFIG5["min_gap"]
    return kw


def make_config5(keep=None):
    """`svim-asm diploid` of the REAL reference on the config-5 sample; committed like `full`: VCF digest, counts,
    first / last records, digests of the inputs' uncompressed content."""
    import time
    from svim_asm_amd import synth_bam
    keep = keep or os.environ.get("SVX_KEEP_CONFIG5")
    d = keep or tempfile.mkdtemp(prefix="svx_config5_")
    fasta, bams = synth_bam.write_dataset(d, **config5_dataset_args())
    wd = os.path.join(d, "wd_reference")
    t0 = time.time()
    run_reference_cli(["diploid", wd, bams[0], bams[1], fasta])
    t_ref = time.time() - t0
    vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
    log = "".join(open(os.path.join(wd, f)).read() for f in os.listdir(wd) if f.endswith(".log"))
    meta = {"params": CONFIG5,
            "payload_sha256": dict([(os.path.basename(f), synth_bam.payload_digest(f)) for f in bams] +
                                   [(os.path.basename(fasta), synth_bam.file_digest(fasta))]),
            "reference_wall_s_build_container": round(t_ref, 1),
            "reference_log_pairing_lines": [l.split("]  ", 1)[-1] for l in log.split("\n") if "Pairing " in l]}
    meta.update(_vcf_summary(vcf))
    with open(os.path.join(GOLD, "config5_inputs.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    if not keep:
        shutil.rmtree(d)


def refresh_payload_digests():
    """Add `payload_sha256` (uncompressed-content digests) to medium / large / longcigar metadata: the inputs are
    regenerated here, checked byte for byte against the compressed-file digests taken when the reference ran on
    them (same container, same zlib), and the zlib-independent digests of exactly those inputs are stored."""
    import hashlib
    from svim_asm_amd import synth_bam
    jobs = [("medium_inputs.json", dict(seed=MEDIUM["seed"], contigs=medium_contigs(), n_shared=MEDIUM["n_shared"],
                                        n_private=MEDIUM["n_private"], median_aln=MEDIUM["median_aln"], mean_m=MEDIUM["mean_m"])),
            ("large_inputs.json", large_dataset_args()),
            ("longcigar_inputs.json", dict(seed=LONGCIGAR["seed"], contigs=tuple((n, l) for n, l in LONGCIGAR["contigs"]),
                                           n_shared=LONGCIGAR["n_shared"], n_private=LONGCIGAR["n_private"],
                                           median_aln=LONGCIGAR["median_aln"], mean_m=LONGCIGAR["mean_m"]))]
    for name, kw in jobs:
        path = os.path.join(GOLD, name)
        meta = json.load(open(path))
        d = tempfile.mkdtemp(prefix="svx_digest_")
        fasta, bams = synth_bam.write_dataset(d, **kw)
        for f in [fasta] + bams:
            want = meta["sha256"].get(os.path.basename(f))
            if want is not None and hashlib.sha256(open(f, "rb").read()).hexdigest() != want:
                raise SystemExit("%s: regenerated %s differs from the file the golden was made from" % (name, f))
        meta["payload_sha256"] = dict([(os.path.basename(f), synth_bam.payload_digest(f)) for f in bams] +
                                      [(os.path.basename(fasta), synth_bam.file_digest(fasta))])
        with open(path, "w") as fh:
            json.dump(meta, fh, indent=1)
        shutil.rmtree(d)
        print(name, "ok")


LONGCIGAR = dict(seed=5, contigs=[["chr1", 1500000], ["chr2", 400000], ["chr3", 90000]], n_shared=30, n_private=6,
                 median_aln=40000000, mean_m=12)


def make_longcigar():
    """Diploid sample whose contig-spanning alignments have CIGARs of > 65535 operations (chr1:
    > 10^5), i.e. records stored with the `kSmN` placeholder + CG:B,I tag (SAM spec §4.2.2).  The
    real reference reads them through the stub pysam, which restores the CIGAR the way htslib does.
    Only the VCF and the input digests are committed; tests regenerate the inputs from the seeds."""
    import gzip
    import hashlib
    from svim_asm_amd import bamio, synth_bam
    d = tempfile.mkdtemp(prefix="svx_longcigar_")
    contigs = tuple((n, l) for n, l in LONGCIGAR["contigs"])
    fasta, bams = synth_bam.write_dataset(d, seed=LONGCIGAR["seed"], contigs=contigs, n_shared=LONGCIGAR["n_shared"],
                                          n_private=LONGCIGAR["n_private"], median_aln=LONGCIGAR["median_aln"],
                                          mean_m=LONGCIGAR["mean_m"])
    n_ops = [int(bamio.AlignmentFile(b, reader="python")._cols["n_cig"].max()) for b in bams]
    assert min(n_ops) > 100000, n_ops
    wd = os.path.join(d, "wd")
    run_reference_cli(["diploid", wd, bams[0], bams[1], fasta])
    vcf = masked_vcf(os.path.join(wd, "variants.vcf"))
    with gzip.GzipFile(os.path.join(GOLD, "longcigar_diploid.vcf.gz"), "wb", compresslevel=9, mtime=0) as fh:
        fh.write(vcf.encode())
    digest = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in [fasta] + bams}
    with open(os.path.join(GOLD, "longcigar_inputs.json"), "w") as fh:
        json.dump({"params": LONGCIGAR, "sha256": digest, "max_cigar_ops": n_ops,
                   "records": sum(1 for l in vcf.split("\n") if l and l[0] != "#")}, fh, indent=1)
    shutil.rmtree(d)


def make_function_vectors():
    import random
    ref = load_reference()
    import pysam  # the stub (oracle/refstub), on sys.path after load_reference()
    vec = {}
    # analyze_cigar_indel on random tuples (all op codes)
    rnd = random.Random(7)
    cases = []
    for _ in range(40):
        n = rnd.randint(0, 60)
        tuples = [(rnd.randint(0, 9), rnd.choice([0, 1, 5, 29, 30, 31, 39, 40, 41, 100, 5000])) for _ in range(n)]
        m = rnd.choice([1, 30, 40])
        cases.append({"tuples": tuples, "min_length": m, "out": ref["INTRA"].analyze_cigar_indel(tuples, m)})
    vec["analyze_cigar_indel"] = cases
    # is_similar (reference test_inter.py vectors + boundary)
    sims = [("chrI", 0, 100, "chrII", 0, 100), ("chrI", 0, 100, "chrI", 0, 100), ("chrI", 0, 100, "chrI", 10, 90),
            ("chrI", 0, 100, "chrI", 21, 100), ("c", 0, 0, "c", 19, 0), ("c", 0, 0, "c", 20, 0), ("c", 5, 40, "c", 5, 60)]
    vec["is_similar"] = [{"args": list(a), "out": bool(ref["INTER"].is_similar(*a))} for a in sims]
    # retrieve_other_alignments on the reference's own BAM fixtures (tests/test_satag.py)
    sa = {}
    for fn in ("chimeric_read.bam", "chimeric_read_errors.bam"):
        bam = pysam.AlignmentFile(os.path.join(REF, "tests", fn))
        rows = []
        for aln in bam.fetch(until_eof=True):
            if aln.is_supplementary:
                continue
            others = ref["COLLECT"].retrieve_other_alignments(aln, bam)
            rows.append([dict(cigarstring=o.cigarstring, reference_id=o.reference_id, reference_start=o.reference_start,
                              reference_end=o.reference_end, flag=o.flag, mapping_quality=o.mapping_quality,
                              query_alignment_start=o.query_alignment_start, query_alignment_end=o.query_alignment_end,
                              infer_read_length=o.infer_read_length()) for o in others])
        sa[fn] = rows
    vec["retrieve_other_alignments"] = sa
    with open(os.path.join(GOLD, "functions.json"), "w") as fh:
        json.dump(vec, fh, indent=0)

PIPE_NAMES = ["chr1", "chr10", "chr2", "chrX"]
PIPE_LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000]


def _ref_stub_bam(records, names, lengths):
    import pysam  # the stub

    class Bam(object):
        references = tuple(names)

        def __init__(self):
            self.lengths = tuple(lengths)
            self.recs = []
            for r in records:
                a = pysam.AlignedSegment()
                a.query_name = r["qname"]
                a.flag = r["flag"]
                a.reference_id = r["tid"]
                a.reference_start = r["pos"]
                a.mapping_quality = r["mapq"]
                a.cigartuples = [tuple(t) for t in r["cigar"]]
                a.query_sequence = r["seq"]
                if r.get("sa") is not None:
                    a.set_tags([("SA", r["sa"], "Z")])
                self.recs.append(a)

        def fetch(self, contig=None):
            tid = names.index(contig)
            return iter([a for a in self.recs if a.reference_id == tid])

        def get_tid(self, n):
            return names.index(n) if n in names else -1

        def get_reference_name(self, tid):
            if tid < 0:
                raise ValueError("bad tid")
            return names[tid]

        getrname = get_reference_name

        def get_reference_length(self, n):
            return lengths[names.index(n)]
    return Bam()


def make_pipeline_vectors():
    """Function-level vectors of COLLECT / analyze_read_segments / form_partitions / pair_candidates from
    the real reference.  Inputs come from the seeded generators of tests/helpers.py and are stored in
    full, so the file is self-contained."""
    import gzip
    import numpy as np
    ref = load_reference()
    from tests import helpers
    vec = {"names": PIPE_NAMES, "lengths": PIPE_LENGTHS}
    # ---- COLLECT + per-read analyze_read_segments
    option_sets = [dict(), dict(min_sv_size=30, max_sv_size=3000, min_mapq=0, query_gap_tolerance=500)]
    collect = []
    for seed, kw in enumerate(option_sets):
        rng = np.random.default_rng(7000 + seed)
        recs = helpers.random_records(rng, PIPE_NAMES, PIPE_LENGTHS, 10) + \
            helpers.engineered_split_records(rng, PIPE_NAMES, PIPE_LENGTHS, 36)
        recs.sort(key=lambda r: (r["tid"], r["pos"]))
        o = helpers.options(**kw)
        bam = _ref_stub_bam(recs, PIPE_NAMES, PIPE_LENGTHS)
        out = [candidate_tuple(c) for c in ref["COLLECT"].analyze_alignment_file_coordsorted(bam, o)]
        per_read = []
        for i, aln in enumerate(bam.recs):
            if aln.is_supplementary or aln.is_secondary or aln.is_unmapped or aln.mapping_quality < o.min_mapq:
                continue
            supp = [s for s in ref["COLLECT"].retrieve_other_alignments(aln, bam)
                    if not s.is_unmapped and s.mapping_quality >= o.min_mapq]
            if supp:
                per_read.append({"record": i,
                                 "out": [candidate_tuple(c) for c in ref["INTER"].analyze_read_segments(aln, supp, bam, o)]})
        collect.append({"options": kw, "records": recs, "out": out, "analyze_read_segments": per_read})
    vec["collect"] = collect
    # ---- form_partitions + pair_candidates
    pair = []
    for seed, kw in enumerate([dict(), dict(max_edit_distance=10, partition_max_distance=100, query_names=True)]):
        rng = np.random.default_rng(7100 + seed)
        names = PIPE_NAMES
        seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=25000)) for n in names}
        lengths = [25000] * len(names)
        fasta = helpers.FakeFasta(seqs)

        class Bam(object):
            references = tuple(names)

            def get_reference_length(self, n):
                return lengths[names.index(n)]
        bam = Bam()
        t1 = helpers.random_candidates(rng, names, lengths, seqs, 90, "h1")
        t2 = helpers.random_candidates(rng, names, lengths, seqs, 90, "h2")
        for c in t1[:50]:
            if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
                shift = int(rng.integers(-3, 4))
                lst = list(c)
                lst[2] = max(0, c[2] + shift)
                lst[3] = max(lst[2], c[3] + shift)
                lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
                t2.append(tuple(lst))
        o = helpers.options(**kw)
        c1 = [helpers.build_candidate(t, bam, ref["CAND"]) for t in t1]
        c2 = [helpers.build_candidate(t, bam, ref["CAND"]) for t in t2]
        parts = {}
        for typ in ("DEL", "INV", "INS", "DUP_TAN", "DUP_INT", "BND"):
            sub = [(1, c) for c in c1 if c.type == typ] + [(2, c) for c in c2 if c.type == typ]
            where = {id(c): k for k, (_, c) in enumerate(sub)}
            parts[typ] = [[where[id(c)] for _, c in p] for p in ref["COMBINE"].form_partitions(sub, o.partition_max_distance)]
        out = [candidate_tuple(c) for c in ref["COMBINE"].pair_candidates(c1, c2, fasta, bam, o)]
        pair.append({"options": kw, "seqs": seqs, "t1": t1, "t2": t2, "form_partitions": parts, "out": out})
    vec["pair"] = pair
    with gzip.GzipFile(os.path.join(GOLD, "pipeline_vectors.json.gz"), "wb", compresslevel=9, mtime=0) as fh:
        fh.write(json.dumps(vec).encode())  # (mtime 0: regenerating the fixture leaves the bytes unchanged)


def main():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.makedirs(GOLD, exist_ok=True)
    logging.getLogger().setLevel(logging.WARNING)
    if len(sys.argv) > 1:   # regenerate selected fixtures only: functions / config1 / medium / longcigar
        for what in sys.argv[1:]:
            {"functions": make_function_vectors, "config1": make_config1, "medium": make_medium,
             "longcigar": make_longcigar, "pipeline": make_pipeline_vectors, "large": make_large, "full": make_full, "config5": make_config5, "medium_options": make_medium_options,
             "digests": refresh_payload_digests}[what]()
        return
    make_longcigar()
    make_function_vectors()
    make_pipeline_vectors()
    out = make_config1()
    make_medium()
    # the two reference BAM fixtures are data, not source: keep copies for the GPU box
    for fn in ("chimeric_read.bam", "chimeric_read_errors.bam"):
        shutil.copy(os.path.join(REF, "tests", fn), os.path.join(GOLD, fn))
    print("golden vectors written under", GOLD, "and", out)


if __name__ == "__main__":
    main()
