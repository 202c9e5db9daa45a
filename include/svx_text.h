/*
 * svx_text.h — C-ABI of the native text side of libsvx.so: indexed-FASTA access and the VCF body
 * (SURVEY.md §8 f-3).  Host code (C++), no device work: the reference does this part through
 * pysam.FastaFile (svim-asm:124; SVIM_COMBINE.py:45-99,467; SVCandidate.py:57-58,105,155,210,301-302)
 * and one Python format call per record (SVCandidate.py get_vcf_entry*, SVIM_COMBINE.py:428-477); at human
 * scale that is 10^5 fetch() calls and 10^5 str.format() calls — here each is one batch call over columns.
 *
 * All functions return SVX_OK (0) or a negative svx_status (svx.h); nothing throws or aborts.
 */
#ifndef SVX_TEXT_H_
#define SVX_TEXT_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ FASTA ---- */
typedef struct svx_fasta svx_fasta;

/* Map `path` read-only.  The caller has parsed `<path>.fai` (five columns: name, length, offset of the first
 * base, bases per line, bytes per line) and passes columns 2-5 for the n_refs sequences.  On failure *out is
 * NULL and a message is copied into err[err_cap] when err != NULL. */
int svx_fasta_open(const char* path, int32_t n_refs, const int64_t* length, const int64_t* offset,
                   const int32_t* line_bases, const int32_t* line_width, svx_fasta** out, char* err, size_t err_cap);
void svx_fasta_close(svx_fasta* fa);

/*
 * pysam.FastaFile.fetch(reference, start, end) for n intervals at once: the bases of sequence ref[i] in
 * [start[i], min(end[i], length)) — line ends skipped — at out + out_off[i]; `upper` != 0 applies ASCII
 * str.upper() on the way (the reference upper-cases every slice it fetches, SVIM_COMBINE.py:45-99,
 * SVCandidate.py:57-58,105,155,210,301-302).  out_off[n + 1] must hold exactly the clipped lengths back to
 * back (the caller knows them from the .fai lengths).  start < 0, end < start or a sequence that is shorter on
 * disk than its index entry says: SVX_E_INVALID.  Intervals are spread over n_threads host threads (<= 0:
 * one per hardware thread, at most 16).
 */
int svx_fasta_fetch_batch(const svx_fasta* fa, const int32_t* ref, const int64_t* start, const int64_t* end,
                          uint32_t n, int upper, const uint64_t* out_off, uint8_t* out, int n_threads);

/* -------------------------------------------------------------------- VCF ---- */
/*
 * The record lines of write_final_vcf (SVIM_COMBINE.py:428-477): one call formats every entry, sorts them the
 * way sorted_nicely does (:369-376 — natural order of the contig names, then start, then end; stable) and
 * numbers the IDs per SV type label in sorted order (:470-475).
 *
 * Entries arrive in the reference's LIST order (:431-464: deletions, inversions, insertions, tandem
 * duplications, interspersed duplications, then two entries per breakend).  Entry e describes candidate row
 * row[e] of the candidate columns in the way kind[e] says:
 */
#define SVX_VCF_DEL 0          /* CandidateDeletion.get_vcf_entry              SVCandidate.py:53-78    */
#define SVX_VCF_INV 1          /* CandidateInversion.get_vcf_entry             :99-125                 */
#define SVX_VCF_INS 2          /* CandidateInsertion.get_vcf_entry             :151-176                */
#define SVX_VCF_DUPTAN_INS 3   /* CandidateDuplicationTandem.get_vcf_entry_as_ins  :203-232            */
#define SVX_VCF_DUPTAN_DUP 4   /*   ... get_vcf_entry_as_dup                   :234-261                */
#define SVX_VCF_DUPINT_INS 5   /* CandidateDuplicationInterspersed.get_vcf_entry_as_ins  :296-321      */
#define SVX_VCF_DUPINT_DUP 6   /*   ... get_vcf_entry_as_dup                   :323-347                */
#define SVX_VCF_BND 7          /* CandidateBreakend.get_vcf_entry              :389-415                */
#define SVX_VCF_BND_REV 8      /*   ... get_vcf_entry_reverse                  :417-443                */

typedef struct svx_vcf_in {
    /* ---- candidate columns (n_rows rows; svim_asm_amd/table.py) */
    uint32_t n_rows;
    const int32_t* sc;        /* source contig id  */
    const int64_t* ss;        /* source start      */
    const int64_t* se;        /* source end        */
    const int32_t* dc;        /* destination contig id */
    const int64_t* ds;
    const int64_t* de;
    const uint8_t* flag;      /* bit 0 complete / fully_covered / cutpaste; bit 1 source 'rev', bit 2 dest 'rev' */
    const int64_t* copies;
    const uint8_t* gt;        /* index into genotypes */
    const int64_t* q_off;     /* inserted sequence = seqs[q_off .. q_off + q_len) */
    const int64_t* q_len;
    const int64_t* r_off;     /* n_rows + 1: reads of row i = read names r_flat[r_off[i] .. r_off[i+1]) */
    const int64_t* r_flat;
    /* ---- stores (the sizes bound every offset above and below: a slice that leaves its pool is SVX_E_INVALID) */
    const uint8_t* seqs;
    uint64_t seqs_bytes;
    const char* names;        /* read-name pool (only read when read_names != 0) */
    const int64_t* name_off;  /* n_names + 1 */
    uint64_t n_names;
    const char* contigs;      /* contig-name pool */
    const int64_t* contig_off; /* n_contigs + 1 */
    const int32_t* contig_rank; /* rank of every contig name under sorted_nicely's natural key (equal keys share one) */
    uint32_t n_contigs;
    const char* genotypes;    /* genotype strings, pool + offsets */
    const int64_t* genotype_off;
    uint32_t n_genotypes;
    /* ---- entries (n_entries; list order) */
    uint32_t n_entries;
    const uint8_t* kind;
    const uint32_t* row;
    /* reference bases of entry e (sequence alleles only; already upper-cased by svx_fasta_fetch_batch):
     * bases[b_off[e] .. b_off[e] + b_len[e]) = the REF allele the entry's formatter fetches; for
     * SVX_VCF_DUPINT_INS a second slice bases[b2_off[e] ...) = the source interval appended to ALT */
    const uint8_t* bases;
    uint64_t bases_bytes;
    const int64_t* b_off;
    const int64_t* b_len;
    const int64_t* b2_off;
    const int64_t* b2_len;
    int sequence_alleles;     /* not options.symbolic_alleles */
    int read_names;           /* options.query_names */
} svx_vcf_in;

/* Formats into a buffer owned by the library: *text / *n_bytes (every line ends with '\n'); free with
 * svx_vcf_free.  n_lines receives the number of record lines. */
int svx_vcf_format(const svx_vcf_in* in, char** text, uint64_t* n_bytes, uint64_t* n_lines);
void svx_vcf_free(char* text);
/* The same lines written to the file descriptor `fd` behind its current position (where the caller's header lines
 * end; a regular file opened for writing, NOT in append mode): every formatting thread writes its own stretch with
 * pwrite at its final place — no joined buffer, the copies into the page cache run side by side.  The position of
 * `fd` is left behind the last line.  n_bytes receives the bytes written. */
int svx_vcf_write(const svx_vcf_in* in, int fd, uint64_t* n_bytes, uint64_t* n_lines);

#ifdef __cplusplus
}
#endif
#endif /* SVX_TEXT_H_ */
