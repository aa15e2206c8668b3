/*
 * sgcount_hip.h — C ABI of the MI355X-native sgRNA count path (libsgcount_hip.so).
 *
 * This is the drop-in boundary for the hot path of noamteyssier/sgcount v0.1.35.
 * The reference has no FFI of its own (it is a single Rust binary crate); the
 * seam these entry points replace is the in-process call
 *
 *     Counter::new(reader, &Library, &Option<Permuter>, Offset, size, position_recursion)
 *                                                   (src/counter.rs:36-66, called at src/count.rs:26-33)
 *
 * plus the accessors the callers consume (src/counter.rs:71-76, :239-251).
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add to
 * bind them.  Conventions: 0 = OK, negative = error (sgc_last_error() holds a
 * thread-local message); no exceptions or aborts cross the ABI; every buffer
 * is caller-owned; plain pointers and sizes only.  Threading: the tables of a ctx are
 * read-only and shared, but its stream and scratch buffers are not — drive one ctx
 * (and its samples) from one thread at a time; use one ctx per device and per
 * concurrent host thread (the C++ host does: sgcount_amd/csrc/host/sgh.cpp count()).
 *
 * Division of labour (BASELINE.json north_star): the host streams FASTQ and
 * 2-bit-packs each read's guide window into one record (sgc_pack_reads_host,
 * or on-device from raw bytes with sgc_sample_push_reads / _push_fastq); the
 * device does the per-read offset scan (Centered / +1 / -1 windows), the
 * library lookup, the unambiguous single-mismatch probe and the count.
 */
#ifndef SGCOUNT_HIP_H
#define SGCOUNT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGC_OK 0
#define SGC_E_ARG (-1)          /* bad argument */
#define SGC_E_HIP (-2)          /* HIP runtime error / no device */
#define SGC_E_UNSUPPORTED (-3)  /* guide length outside 1..65535; packed records asked for a length outside 1..30 */
#define SGC_E_DUPLICATE (-4)    /* duplicate library sequence — src/library.rs:91-96 panics */
#define SGC_E_STATE (-5)        /* call order (no library set, sample finished, ...) */
#define SGC_E_OOM (-6)
#define SGC_E_FORMAT (-7)       /* malformed FASTQ text (a header line without '@', a separator without '+') — fxread panics */

#define SGC_MEM_HOST 0          /* pointer is host memory (pinned or pageable) */
#define SGC_MEM_DEVICE 1        /* pointer is device memory on the ctx's device */

#define SGC_MAX_GUIDE_LEN 30    /* L + 2 flanking bases must fit one 64-bit word */
#define SGC_REC8_MAX_LEN 23     /* up to here a read packs into ONE u64 (sgc_record_bytes() == 8) */

typedef struct sgc_ctx sgc_ctx;        /* stands for (&Library, &Option<Permuter>) — src/count.rs:87,103-107 */
typedef struct sgc_sample sgc_sample;  /* stands for one Counter — src/counter.rs:17-21 */

/* Per-kernel device timing of the last pushes (HIP events on the ctx stream). */
typedef struct {
    uint64_t launches;       /* kernel launches accumulated since the last reset */
    double   lookup_ms;      /* Σ duration of the lookup/count kernel(s) */
    double   hist_ms;        /* Σ duration of the histogram kernel(s), 0 if fused */
    double   pack_ms;        /* Σ duration of the on-device pack kernel(s) */
    double   part_ms;        /* Σ duration of the partition kernel (partitioned path) */
    double   miss_ms;        /* Σ duration of the miss-resolution kernel (partitioned path) */
    double   h2d_ms;         /* Σ duration of the host-to-device copies of FASTQ text (upload stream) */
} sgc_timing;

typedef struct {
    uint32_t n_guides;
    uint32_t guide_len;
    uint32_t record_bytes;     /* 8 or 16; 0 = byte-string path (non-ACGT library or L > 30): no packed records */
    uint32_t one_mismatch;     /* 1 if the permute table is resident */
    uint64_t lib_slots;        /* open-addressed slots of the library table */
    uint64_t perm_slots;       /* slots of the single-mismatch (permute) table, 0 if exact */
    uint64_t perm_entries;     /* unambiguous children stored (src/permutes.rs map.len(), ACGT children only) */
    uint64_t table_bytes;      /* device bytes of the tables and indexes */
    uint64_t core_partitions;  /* partitions of each core index of the in-LDS resolver (also built for exact
                                  mode, where one exact-only pass uses it); 0 = no core index (L outside 4..23, split-layout
                                  table, or the guides do not spread): probing resolver */
    uint32_t path;             /* which count path serves this library with the default options — the fallbacks are results-preserving but
                                  not equally fast (rates on the 100M-read bench sample, MI355X): 4 = partitioned pass +
                                  in-LDS core resolver (the shipped path: L <= 23; 100k guides 117-128 G reads/s, 150k-200k guides — 128
                                  slices — 100-109 G, 250k-300k guides — 128 slices of 2^13 slots — 90-95 G); 3 = partitioned pass + probing
                                  resolver (no core index: L = 24..30 never, guides that share a 9-base stretch by the thousand, or more
                                  than ~340k guides; 400k guides 47 G); 1 = guide-id array + LDS histogram (two-word records, split-layout
                                  tables, more than 128 slices of 2^13 slots = ~420k guides: 20-26 G); 0 = byte-string path
                                  (record_bytes == 0) */
    uint32_t slices;           /* library slices of the partitioned pass (0 if path < 3): 64 at 100k guides; up to 128 of 2^12 table
                                  slots (~210k guides), beyond that up to 128 of 2^13 slots (one slice-count workgroup per CU) */
    uint32_t slice_record_bytes; /* bytes of a clean record inside a slice block: 5, 6 or 8 (0 if path < 3) */
    uint32_t reserved_;
} sgc_lib_info;

/* ---- context: device + library tables -------------------------------------------------------- */

/* Number of HIP devices visible to the process (0 if none / no driver). */
int sgc_device_count(void);

/* Opens `device`, creates the ctx's stream.  Replaces nothing upstream (the reference has no device). */
int sgc_init(int device, sgc_ctx **out);
void sgc_free(sgc_ctx *);

/* A second ctx on the same device that SHARES the library tables of `src` (built once, read-only afterwards; reference
 * src/count.rs:103-136: one Library and one Permuter lent to every rayon worker) and has its own stream, scratch buffers and
 * samples: what a host with several worker threads per GPU uses instead of building the same tables once per worker.  The
 * tables live until the last ctx that shares them is freed or given another library. */
int sgc_ctx_clone(sgc_ctx *src, sgc_ctx **out);

/* Run everything on a caller-supplied hipStream_t instead of the ctx's own.  A NULL stream is the HIP
 * legacy default stream, exactly as hipStream_t 0 is; SGC_STREAM_OWN returns to the ctx's own stream. */
#define SGC_STREAM_OWN ((void *)(intptr_t)-1)
int sgc_set_stream(sgc_ctx *, void *hip_stream);
void *sgc_get_stream(sgc_ctx *);

/* Library::from_reader (src/library.rs:17-21,89-99) + Permuter::new (src/permutes.rs:47-75) unless
 * enable_1mm == 0 (the reference's -x/--exact, src/count.rs:103-107).  seqs = n rows of L ASCII bytes
 * in library-file order.  Builds the device tables once; they are read-only afterwards and shared by
 * every sample of the ctx.  SGC_E_DUPLICATE mirrors the duplicate-sequence panic.  A library with
 * bytes outside ACGT ('N', lowercase, anything: upstream compares raw bytes, src/library.rs:89-99) or
 * with L > SGC_MAX_GUIDE_LEN has no packed record format: it is served by the byte-string path
 * (hashed tables verified byte for byte, same rules, slower; sgc_library_info reports record_bytes
 * == 0) — such a ctx takes reads and FASTQ text (sgc_sample_push_reads / _push_fastq*), and
 * sgc_sample_push_packed / sgc_pack_reads_device return SGC_E_STATE.  On any error the ctx holds no
 * library afterwards. */
int sgc_set_library(sgc_ctx *, const uint8_t *seqs, uint32_t n, uint32_t L, int enable_1mm);
int sgc_library_info(sgc_ctx *, sgc_lib_info *out);

/* Point lookups against the resident tables (Library::contains src/library.rs:34-40 and
 * Permuter::contains src/permutes.rs:55-57 followed by Library::alias, as at src/counter.rs:111-117).
 * tokens = n rows of L ASCII bytes; gid_out[i] = guide index (library order) or -1.
 * which: 0 = library only, 1 = permuter only, 2 = library then permuter. */
int sgc_lookup(sgc_ctx *, const uint8_t *tokens, uint64_t n, int which, int32_t *gid_out);

/* ---- packing: read -> record ------------------------------------------------------------------ */

/* Bytes per packed record for guide length L: 8 (L <= 23), 16 (L <= 30), 0 = unsupported. */
uint32_t sgc_record_bytes(uint32_t L);

/* Host packer (pure CPU; usable without a GPU).  Applies Counter::apply_trim / bounds
 * (src/counter.rs:144-204) for all of Centered / Plus / Minus at once: read i is
 * seqs[offsets[i] .. offsets[i+1]).  reverse = Offset::Reverse (src/offsetter.rs:10-15).
 * position_recursion = 0 packs the Centered window only (Position::Null, src/counter.rs:44-48). */
int sgc_pack_reads_host(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L, int reverse,
                        uint32_t offset, int position_recursion, void *records_out);

/* Device packer on caller buffers: raw read bytes (device) -> records (device), the same bits as
 * sgc_pack_reads_host.  Asynchronous on the ctx stream.  Needs a library (for L). */
int sgc_pack_reads_device(sgc_ctx *, const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, int reverse,
                          uint32_t offset, int position_recursion, void *d_records_out);

/* ---- one sample = one Counter ------------------------------------------------------------------ */

/* Counter::new's (Offset, position_recursion) — src/counter.rs:36-43.  size is the library's L. */
int sgc_sample_begin(sgc_ctx *, sgc_sample **out, int reverse, uint32_t offset, int position_recursion);

/* The fold of Counter::count (src/counter.rs:211-236) over n packed records.  Asynchronous on the
 * ctx stream; a host buffer may be reused after sgc_sample_sync(). */
int sgc_sample_push_packed(sgc_sample *, const void *records, uint64_t n, int where);

/* Streaming form for a host that packs the records itself (the north star's split: the host streams FASTQ, 2-bit-packs
 * the reads and ships pinned batches; replaces the iterator of Counter::count, src/counter.rs:211-236, with
 * sgc_pack_reads_host / the scanner of the C++ host in the role of Counter::apply_trim).  `records` is HOST memory (pinned:
 * the copy is then truly asynchronous).  The records are uploaded on the ctx's upload stream into a device-side batch
 * buffer of the sample; a count pass runs whenever a batch ("batch_records" option, default 2^24) is full and at the next
 * sgc_sample_flush / _finish / _sync / _export_device, so that many small pushes cost one pass — each pass has a fixed cost
 * of a few launches and table stagings.  Two batch buffers alternate: the upload of one batch overlaps the count pass of the
 * batch before.  The host buffer may be reused once sgc_sample_wait_uploads says so.  Not to be interleaved on one ctx with
 * the asynchronous pushes of another sample. */
int sgc_sample_push_packed_async(sgc_sample *, const void *records, uint64_t n);

/* Same, from raw read bytes: the device packs (pack kernel) then counts. */
int sgc_sample_push_reads(sgc_sample *, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, int where);
/* The same for reads of which only the bytes that matter are shipped: entry i is the piece of read i that holds its windows — in
 * the read's orientation, bases [max(o - 1, 0), min(len, o + L + 1)) (o: the sample's offset; for a reverse-strand sample that is
 * the tail of the line as it stands in the file) — and window_offset is o seen from the start of that piece (1, or 0 when o is 0).
 * Every decision of Counter::assign (src/counter.rs:96-180) depends on the read only through those bytes and through whether the
 * windows fit, which the piece preserves; the C++ scanner ships the reads it routes to the byte-string chain of a hybrid library
 * this way: L + 2 bytes instead of the read. */
int sgc_sample_push_windows(sgc_sample *, const uint8_t *pieces, const uint64_t *offsets, uint64_t n, int where, uint32_t window_offset);

/* Same, from a chunk of FASTQ text holding whole 4-line records: the device finds the record
 * boundaries, packs and counts.  n_records_out may be NULL.  Waits for the device's line count (one stream
 * synchronisation per call); a streaming host uses sgc_sample_push_fastq_part instead. */
int sgc_sample_push_fastq(sgc_sample *, const uint8_t *text, uint64_t n_bytes, int where, uint64_t *n_records_out);

/* Streaming form (what replaces the fxread iterator inside Counter::count, src/counter.rs:211-236, for FASTQ input):
 * `text` is the next run of WHOLE LINES of the sample's FASTQ stream — it may begin and end anywhere in the 4-line
 * cycle — first_line is the 0-based number of its first line within the stream and n_newlines the number of '\n' in
 * it (the host counts them while it reads; UINT64_MAX = unknown: the call then waits for the device's count).  Only
 * the very last part of a stream may end without a '\n'.  Fully asynchronous when n_newlines is given: host text is
 * uploaded on a separate stream into alternating device buffers, so the upload of one part overlaps the ingest and
 * count kernels of the part before; the host buffer may be reused once sgc_sample_wait_uploads says so.  A '\r'
 * before a '\n' counts as part of the line terminator.  Lines 4k must start with '@' and lines 4k+2 with '+':
 * violations (and a wrong n_newlines) surface as SGC_E_FORMAT from the next sgc_sample_sync / sgc_sample_finish.
 * The caller checks at the end of the stream that the total number of lines is a multiple of 4. */
int sgc_sample_push_fastq_part(sgc_sample *, const uint8_t *text, uint64_t n_bytes, int where, uint64_t first_line,
                               uint64_t n_newlines, uint64_t *n_records_out);

/* Blocks until at most max_pending of the most recent asynchronous uploads (sgc_sample_push_fastq_part with host text,
 * sgc_sample_push_packed_async) are still in flight: the host buffers of all earlier ones can then be overwritten. */
int sgc_sample_wait_uploads(sgc_sample *, uint32_t max_pending);

int sgc_sample_sync(sgc_sample *);

/* Counter::get_value for every guide + total_reads / matched_reads (src/counter.rs:71-76,239-246).
 * counts has n_guides entries in library order (the host pools guides that share an id, mirroring
 * the id-keyed fold at src/counter.rs:232-235).  Synchronises. May be called repeatedly. */
int sgc_sample_finish(sgc_sample *, uint64_t *counts, uint64_t *total_reads, uint64_t *matched_reads);

/* Device pointer to the sample's u64 count vector (n_guides entries), valid after
 * sgc_sample_flush(); lets a multi-GPU host hand the vector to RCCL without a host round trip. */
int sgc_sample_flush(sgc_sample *);
void *sgc_sample_device_counts(sgc_sample *);
/* Flush, then copy the u64 count vector followed by {total_reads, matched_reads} (n_guides + 2 words)
 * into a caller-owned DEVICE buffer, asynchronously on the ctx stream (the RCCL hand-off). */
int sgc_sample_export_device(sgc_sample *, uint64_t *d_out);

/* Zero the sample's counts / totals (reuse between bench steps). */
int sgc_sample_reset(sgc_sample *);
void sgc_sample_free(sgc_sample *);

/* Page-locked host buffers for the push_* entry points (pageable memory works too, through a slower staging
 * copy inside the HIP runtime). */
void *sgc_alloc_pinned(size_t bytes);
void sgc_free_pinned(void *p);

/* Tuning knobs (all results-preserving): "variant" (count path variant, DESIGN.md §4: 4 = the shipped pass, default; 3 = the probing
 * resolver, what a library without a core index gets; 1 = the generic kernels, what a library without one-word records gets), "max_chunk"
 * (records per internal pass), "k1_wgs" (workgroups of the partition kernel), "host_build" /
 * "perm_bloom_bits" (how the next sgc_set_library builds the single-mismatch table and its filter), "force_bytes" (the next
 * sgc_set_library uses the byte-string path even for a library the packed path could serve), "place_trials" (1..64, default 1 = off:
 * where the block pool of a large pass — 32M records or more — falls in device memory decides whether its partition and
 * slice-count kernels run up to ~8 % faster or slower, so a host that counts many resident samples of that size on one ctx may
 * opt in: the first such pass then allocates up to that many candidate pools — all held at once, ~1 GB each for 100M records,
 * never more than a quarter of the device memory that is free at that moment —, times the partition kernel on each (under 1 ms
 * each; it stops at a candidate 11 % ahead of the slowest seen) and keeps the fastest: 10-30 ms once per ctx, which a single
 * sample never earns back; sgc_placement_info reports what the search did), "batch_records" (records per device-side batch of
 * sgc_sample_push_packed_async), "hybrid" (0: a library with a few guides outside ACGT is served by the byte-string path alone
 * instead of the hybrid of both; next sgc_set_library), "host_routes" (1: in a hybrid ctx — sgc_lib_info.path == 2 — packed
 * records are accepted after all: the host promises to push records ONLY for reads whose span region [o - 1, o + L + 1) is all
 * ACGT and none of whose windows equals a guide with exactly one other byte up to that byte, and to push every other read as
 * bytes (sgc_sample_push_reads); the C++ scanner does), "balanced" (default 1: the slice-count kernel deals the blocks of ALL library slices out in equal shares, so a sample that a few
 * guides dominate costs what any other does; 0: the same number of workgroups for every slice), "verbose"
 * (diagnostics on stderr).  No environment
 * variable changes what the library computes or which kernels it runs.  "dbg" sets
 * timing-only ablation flags / phase stamps of the kernels: they are compiled out of the shipped library, which refuses
 * a non-zero value (a profiling build — SGC_HIPCC_FLAGS=-DSGC_ABLATE=1 or -DSGC_STAMPS=1 — accepts it; results are
 * WRONG while an ablation is on). */
int sgc_set_option(sgc_ctx *, const char *key, int64_t value);

/* ---- diagnostics -------------------------------------------------------------------------------- */
/* Builds the host-side tables of a library exactly as sgc_set_library would (no device needed) and checks that every guide
 * is reachable where the kernels look for it: the open-addressed array, the two-choice image of the slices, or — for a
 * library of arbitrary bytes / L > 30 — the byte-string table.  stats[4]: [0] = 1 packed path, 2 byte-string path;
 * [1] = library slots; [2] = single-mismatch entries (0 unless enable_1mm); [3] = 1 if the two-choice image was built. */
int sgc_check_host_tables(const uint8_t *seqs, uint32_t n, uint32_t L, int enable_1mm, uint64_t *stats);
/* What the last placement search of the ctx did ("place_trials"): out4 = {candidate pools tried, peak transient bytes held
 * beyond the one kept, duration in microseconds, index of the candidate kept}; all zero if none ran. */
int sgc_placement_info(sgc_ctx *, uint64_t *out4);
int sgc_timing_enable(sgc_ctx *, int on);     /* record HIP events around every kernel (adds a sync at read) */
int sgc_timing_read(sgc_ctx *, sgc_timing *out, int reset);
const char *sgc_last_error(void);
const char *sgc_version(void);

#ifdef __cplusplus
}
#endif
#endif
