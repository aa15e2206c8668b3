/*
 * sgcount_synth.h — synthetic workload generator (libsgcount_synth.so).
 *
 * Not part of the reference's interface: this produces the workloads SURVEY.md §8(d) /
 * BASELINE.json name (100k-guide library, 150-bp reads with a guide at offset 30, a fixed class
 * mix) so that bench.py, the parity tests and the CPU oracle all see byte-identical input.
 * Every read is a pure function of (seed, read index): the host and device generators agree
 * bit for bit, and any slice of a sample can be produced on its own.
 *
 * Read classes (percent of reads):
 *   85 exact guide · 5 one ACGT substitution · 1 one 'N' · 2 one base inserted before the guide
 *   (guide at +1) · 2 one base deleted from the prefix (guide at -1) · 4 random junk 20-mer ·
 *   1 truncated (read ends inside the guide).  1 % of the guides (every 100th) carry 50x weight.
 * mode SGS_MODE_FIXED: prefix 30 bp.  SGS_MODE_STAGGER: prefix 28..32 bp with weights
 *   5/10/70/10/5 % (the auto-offset workload, BASELINE.json configs[3]).
 */
#ifndef SGCOUNT_SYNTH_H
#define SGCOUNT_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGS_MODE_FIXED 0
#define SGS_MODE_STAGGER 1
/* | SGS_MODE_DOMINANT(pct): pct (1..99) percent of the reads draw ONE guide (index 7 % n) instead of the weighted pick — a sample
 * that a single guide dominates */
#define SGS_MODE_DOMINANT(pct) (((unsigned)(pct) & 127u) << 8)
#define SGS_READ_LEN 150
#define SGS_PREFIX_LEN 30

/* Library: n distinct L-mers (xoshiro256** seeded by `seed`), the last min(200, n/2*2) of them
 * replaced by planted pairs at Hamming distance 1 (even pairs) and 2 (odd pairs).
 * seqs_out: n*L ASCII bytes.  L <= 32. */
int sgs_library(uint64_t seed, uint32_t n, uint32_t L, uint8_t *seqs_out);
/* FASTA text of that library with ids sg000000...; returns bytes written (or needed if cap too small). */
size_t sgs_library_fasta(const uint8_t *seqs, uint32_t n, uint32_t L, uint8_t *out, size_t cap);

/* Length of read i (host). */
uint32_t sgs_read_len(uint64_t seed, uint64_t i, uint32_t L, uint32_t mode);
/* Class of read i: 0 exact, 1 substitution, 2 N, 3 insertion, 4 deletion, 5 junk, 6 truncated; and the guide drawn. */
uint32_t sgs_read_class(uint64_t seed, uint64_t i, uint32_t n_guides, uint32_t L, uint32_t mode, uint32_t *gid_out);

/* Reads [first, first+n) as contiguous bytes + offsets (n+1 entries, offsets[0] = 0).  Host. */
int sgs_reads_host(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs, uint32_t n_guides, uint32_t L,
                   uint32_t mode, uint8_t *seqs_out, uint64_t *offsets_out);
/* Same reads as FASTQ text ("@r<index>\n<seq>\n+\n<I...>\n").  Returns bytes written, or the size needed
 * when out is NULL / cap too small. */
size_t sgs_fastq_host(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs, uint32_t n_guides,
                      uint32_t L, uint32_t mode, uint8_t *out, size_t cap);

/* Device generators (pointers are device memory on the current device; stream = hipStream_t or NULL).
 * lens_out[i] = read length (u32).  The caller turns lengths into offsets (exclusive scan). */
int sgs_read_lens_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, uint32_t L, uint32_t mode,
                         uint32_t *lens_out);
int sgs_reads_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs,
                     uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *offsets, uint8_t *seqs_out);
/* FASTQ text on the device: rec_lens_out[i] = bytes of record i; then fill with record offsets. */
int sgs_fastq_lens_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, uint32_t L, uint32_t mode,
                          uint32_t *rec_lens_out);
int sgs_fastq_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs,
                     uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *rec_offsets, uint8_t *text_out);

const char *sgs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
