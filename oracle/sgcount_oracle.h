/*
 * sgcount_oracle.h — CPU ORACLE for the sgcount count path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (noamteyssier/sgcount v0.1.35: src/library.rs, src/permutes.rs,
 * src/counter.rs, src/offsetter.rs, src/results.rs, src/genemap.rs,
 * src/utils.rs).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it — as the checker / reported baseline, never as
 * the product path.  Nothing under sgcount_amd/ links or imports it.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 * (counter.rs:283-382, permutes.rs:193-253, library.rs:119-136,
 * offsetter.rs:249-362, results.rs:134-177, genemap.rs:124-156,
 * utils.rs:55-104) re-expressed in tests/test_oracle_kat.py, plus the example/
 * fixtures whose read headers carry the generating guide sequence.
 * The reference binary itself is Rust and cannot be built here (no cargo, no
 * vendored crates), so there is no oracle/_ref.  Behaviour of the third-party
 * FASTX reader (fxread ^0.2.5) beyond what those tests pin — notably
 * Record::seq_rev_comp() on non-ACGT bytes — is UNPINNED and documented where
 * restated below.
 */
#ifndef SGCOUNT_ORACLE_H
#define SGCOUNT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (the reference panics / bails; the oracle returns codes) */
#define ORC_OK 0
#define ORC_E_DUPLICATE_SEQ (-1)   /* library.rs:91-96 panic */
#define ORC_E_INCONSISTENT (-2)    /* library.rs:83 "Library sequence sizes are inconsistent" */
#define ORC_E_EMPTY (-3)           /* library.rs:74 unwrap on empty */
#define ORC_E_FORMAT (-4)          /* malformed fastx text */
#define ORC_E_ARG (-5)
#define ORC_E_SHORT (-6)           /* offsetter.rs:154-156 / count.rs:98-100 */
#define ORC_E_NAN (-7)             /* offsetter.rs:123-141 argmin/min error → panic */

/* Position enum, counter.rs:7-12 */
enum { ORC_POS_PLUS = 0, ORC_POS_MINUS = 1, ORC_POS_CENTERED = 2, ORC_POS_NULL = 3 };

typedef struct orc_library orc_library;
typedef struct orc_permuter orc_permuter;
typedef struct orc_counter orc_counter;

/* ---- Library (library.rs) ---- */
/* Build from FASTA/FASTQ text (uncompressed).  err receives an ORC_E_* code. */
orc_library *orc_library_from_text(const uint8_t *buf, size_t len, int *err);
void orc_library_free(orc_library *);
size_t orc_library_size(const orc_library *);      /* library.rs:60 */
size_t orc_library_n(const orc_library *);         /* number of keys */
/* library.rs:34-40 contains(): returns id pointer/len or NULL */
const uint8_t *orc_library_contains(const orc_library *, const uint8_t *tok, size_t n, size_t *id_len);
/* i-th record in file order (the oracle's deterministic stand-in for HashMap order) */
const uint8_t *orc_library_seq(const orc_library *, size_t i);
const uint8_t *orc_library_id(const orc_library *, size_t i, size_t *id_len);

/* ---- Permuter (permutes.rs) ---- */
orc_permuter *orc_permuter_new(const orc_library *);                       /* permutes.rs:47-75 over library.keys() */
orc_permuter *orc_permuter_from_seqs(const uint8_t *seqs, size_t n, size_t L); /* raw n×L, for KATs */
void orc_permuter_free(orc_permuter *);
/* permutes.rs:55-57 contains(): child → parent sequence (L bytes) or NULL */
const uint8_t *orc_permuter_contains(const orc_permuter *, const uint8_t *tok, size_t n);
size_t orc_permuter_map_len(const orc_permuter *);
size_t orc_permuter_null_len(const orc_permuter *);
int orc_permuter_null_contains(const orc_permuter *, const uint8_t *tok, size_t n);

/* ---- Counter (counter.rs) ---- */
/* counter.rs:158-180 bounds(); returns 1 and sets min/max, or 0 for None */
int orc_bounds(size_t seq_len, size_t offset, size_t size, int position, size_t *min, size_t *max);

/* counter.rs:36-66 Counter::new split into new/feed so that callers can stream
 * FASTX text in chunks (each chunk must hold whole records).  permuter may be
 * NULL (exact). reverse: 0 = Offset::Forward(offset), 1 = Offset::Reverse(offset). */
orc_counter *orc_counter_new(const orc_library *, const orc_permuter *, int reverse, size_t offset,
                             size_t size, int position_recursion);
int orc_counter_feed_text(orc_counter *, const uint8_t *buf, size_t len);
/* feed one bare sequence (what Counter::assign sees as record.seq()) */
void orc_counter_feed_seq(orc_counter *, const uint8_t *seq, size_t n);
void orc_counter_free(orc_counter *);
uint64_t orc_counter_get_value(const orc_counter *, const uint8_t *id, size_t id_len); /* counter.rs:71-76 */
uint64_t orc_counter_total_reads(const orc_counter *);                                 /* counter.rs:239 */
uint64_t orc_counter_matched_reads(const orc_counter *);                               /* counter.rs:244 */
double orc_counter_fraction_mapped(const orc_counter *);                               /* counter.rs:249-251 */
/* convenience: counts_out[i] = get_value(id of i-th library record) */
void orc_counter_table(const orc_counter *, const orc_library *, uint64_t *counts_out);

/* one-shot: library text + reads text → per-guide counts in library file order */
int orc_count_text(const uint8_t *lib_buf, size_t lib_len, const uint8_t *reads_buf, size_t reads_len,
                   int reverse, size_t offset, int exact, int position_recursion,
                   uint64_t *counts_out, size_t n_counts, uint64_t *total, uint64_t *matched);

/* ---- Offsetter (offsetter.rs) ---- */
/* offsetter.rs:90-95 positional_entropy over at most `take` records of a fastx
 * text (take = SIZE_MAX for the library).  Returns length (size of first
 * record) and fills out[] (caller provides cap).  First record is consumed for
 * its size and not counted (offsetter.rs:37-39,57). */
/* offsetter.rs:55-79 position_counts: out is (size,4) row-major, cap_rows rows available */
int orc_position_counts(const uint8_t *buf, size_t len, size_t take, double *out, size_t cap_rows, size_t *n_out);
int orc_positional_entropy(const uint8_t *buf, size_t len, size_t take, double *out, size_t cap, size_t *n_out);
/* offsetter.rs:153-163 minimize_mse: returns ORC_OK and sets *reverse,*index */
int orc_minimize_mse(const double *ref, size_t n_ref, const double *cmp, size_t n_cmp, int *reverse, size_t *index);
/* offsetter.rs:165-183 entropy_offset for one input */
int orc_entropy_offset(const uint8_t *lib_buf, size_t lib_len, const uint8_t *reads_buf, size_t reads_len,
                       size_t subsample, int *reverse, size_t *index);

/* ---- utils.rs / genemap.rs / results.rs (the callers' side of the path) ---- */
/* utils.rs:18-49 generate_sample_names.  paths: n NUL-terminated strings back to back; out: names joined by
 * '\n'.  Returns ORC_OK, or ORC_E_ARG if cap is too small.  *fell_back = 1 if duplicate basenames forced
 * the "Sample.N" names. */
int orc_generate_sample_names(const char *paths, size_t n, char *out, size_t cap, int *fell_back);

typedef struct orc_genemap orc_genemap;
#define ORC_E_NOTAB (-8)        /* genemap.rs:58 panic "Missing '\t' in gene map" */
#define ORC_E_DUPKEY (-9)       /* genemap.rs:60-64 assert "Duplicate sgRNA key found in gene map" */
#define ORC_E_NOGENE (-10)      /* results.rs:59 panic "Missing sgrna -> gene mapping" */
/* genemap.rs:53-72 build() from text */
orc_genemap *orc_genemap_from_text(const uint8_t *buf, size_t len, int *err);
void orc_genemap_free(orc_genemap *);
const uint8_t *orc_genemap_get(const orc_genemap *, const uint8_t *sgrna, size_t n, size_t *gene_len);  /* :76-78 */
/* genemap.rs:81-86 missing_aliases over library.values() (file order here): index of first missing or -1 */
long orc_genemap_missing(const orc_genemap *, const orc_library *);

/* results.rs:32-43 generate_columns; names: n NUL-terminated strings back to back */
int orc_generate_columns(const char *names, size_t n, int with_genemap, char *out, size_t cap);
/* results.rs:71-99 write_results into a buffer: header line + one line per library alias (file order),
 * counts[s * n_guides + i] = per-guide counts of sample s (pooled by id like Counter::get_value).
 * Returns bytes written (excluding NUL), or a negative ORC_E_* code. */
long orc_format_results(const orc_library *, const uint64_t *counts, size_t n_samples, const char *names,
                        const orc_genemap *genemap, int include_zero, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
