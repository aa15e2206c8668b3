// sgc_kernels.h — host-callable launchers of the gfx950 kernels (sgc_kernels.hip).
#pragma once
// Timing-only ablation branches of the kernels (tools/tune.py: "what does this phase cost?" — results are WRONG when one is
// taken) exist only in a library built with -DSGC_ABLATE=1 (SGC_HIPCC_FLAGS); in the shipped kernels SGC_DBG() is the
// constant false, the branches and the scalar registers they held are gone, and sgc_set_option("dbg", != 0) is refused.
#ifndef SGC_ABLATE
#define SGC_ABLATE 0
#endif
#define SGC_DBG(word, bits) (SGC_ABLATE && ((word) & (bits)))
// -DSGC_CHECK=1 (a second library, libsgcount_hip_check.so, that tests/test_check_gpu.py runs the pass ladder on): every index the
// partitioned pass computes into its scratch — pool blocks, slots inside a block, the miss runs — is compared with the bound of
// that buffer; a violation sets a bit of the sample's error word and SKIPS the access instead of walking off the buffer (a fault
// on this pool of GPUs takes the machine down), and sgc_sample_finish reports it as SGC_E_STATE.  In the shipped library
// SGC_BOUND() is the constant true and costs nothing.
#ifndef SGC_CHECK
#define SGC_CHECK 0
#endif
#define SGC_BOUND(ok, errp, bit) (!SGC_CHECK || (ok) || (atomicOr((errp), 1ull << (bit)), false))
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_format.h"

// slots per library-table slice staged in LDS by the partitioned path: 2^12 x 8 B = 32 KiB of keys plus
// 16 KiB of counters, so that two 1024-lane workgroups share a CU's 160 KiB
#define SGC_LDS_LOG2_SLICE 12u
// ... and for a library that would need more than 128 such slices (> ~210k guides: tiling and paired-guide libraries): slices of 2^13
// slots — 64 KiB of keys + 32 KiB of counters, ONE workgroup of k_count_slices per CU (it runs at 0.94 of the rate of two:
// tools/occupancy_probe.sh) — so that the partitioned pass serves up to ~420k guides (round 4; before, such a library fell to the generic kernels)
#define SGC_LDS_LOG2_SLICE_BIG 13u
#define SGC_PART_MAX_LOG2_SLICES 7u      // k_partition deals records to at most 128 slices (+ the generic partition)
// the slice size for a library of n guides (table load <= 0.4, as sgc_build_library_table allocates it)
static inline uint32_t sgc_choose_log2_slice(uint64_t n_guides) {
    uint64_t want = (uint64_t)((double)n_guides / 0.4) + 1;
    uint32_t l = 4;
    while ((1ull << l) < want) l++;
    return l > SGC_LDS_LOG2_SLICE + SGC_PART_MAX_LOG2_SLICES ? SGC_LDS_LOG2_SLICE_BIG : SGC_LDS_LOG2_SLICE;
}
// 64-bit words of the library Bloom filter the miss resolver stages in LDS (2^13 x 8 B = 64 KiB)
#define SGC_LIB_BLOOM_LOG2_WORDS 13u

struct sgc_bloom_view {
    const uint64_t *words;
    uint32_t log2_words;
    uint32_t pad_;
};

void sgc_launch_lookup(hipStream_t st, const uint64_t *keys, uint64_t n, const sgc_table_view &lib,
                       const sgc_table_view &perm, int which, bool has_perm, int32_t *out);
void sgc_launch_fold(hipStream_t st, uint32_t *c32, unsigned long long *c64, uint32_t n);
void sgc_launch_export(hipStream_t st, uint32_t *c32, unsigned long long *c64, const unsigned long long *matched,
                       unsigned long long total, uint32_t n, unsigned long long *out);
void sgc_launch_pack_reads(hipStream_t st, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L,
                           bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs);
void sgc_launch_lookup_gids(hipStream_t st, const uint64_t *recs, uint64_t n, uint32_t L, bool rec16,
                            const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm, uint32_t *gids,
                            unsigned long long *matched);
void sgc_launch_hist_slices(hipStream_t st, const uint32_t *gids, uint64_t n, uint32_t n_guides, uint32_t *counts);

// ---- partitioned count path (sgc_part.hip) ------------------------------------------------------
#define SGC_DESC_TAIL 69632u
struct sgc_part_geometry {
    uint32_t k1_wgs, blocks_per_wg, n_blocks, partitions, n_segs, block_records;
    uint64_t per_wg, pool_bytes, desc_bytes, gids_bytes;
    uint64_t wcnt_off, wlist_off; // per-(K1 workgroup, partition) block counts and lists, behind the descriptors
    uint64_t desc_tail_off;      // desc_bytes includes SGC_DESC_TAIL zeroed bytes at this offset (scratch counters of later stages)
};
bool sgc_part_supported(const sgc_table_view &lib, bool rec16);
void sgc_part_plan(uint64_t n, const sgc_table_view &lib, uint32_t max_wgs, sgc_part_geometry *g);
// sub_bits: with core-hashed slices, log2 (1..2) of core pass A's partitions per slice, tagged into the clean records; else 0
void sgc_launch_part_k1(hipStream_t st, unsigned long long *err /* the sample's error word (SGC_CHECK builds) */, const uint64_t *recs, uint64_t n, uint32_t L, const sgc_table_view &lib, uint32_t sub_bits,
                        const sgc_part_geometry &g, uint64_t *pool, uint32_t *desc,
                        int slice_rec /* what the slice blocks hold: 0 = 8-byte records, 1 = six-byte (direct runs, core-hashed slices, 2 (L + 2) + 2 <= 48), 2 = five-byte (2 (L + 2) - slice bits <= 40) */,
                        uint32_t dbg,
                        uint32_t *slice_tot /* NULL, or SGC_SLICE_TOT words, all zero: the blocks of every slice are added up there for the
                                               balanced shares of k_count_slices */);
#define SGC_SLICE_TOT 256u
struct sgc_runs;       // sgc_runs.h
// runs != NULL: the leftovers (misses, generic blocks) are laid out as the runs of core pass A by the kernel's epilogue
uint32_t sgc_part_k2_grid(const sgc_part_geometry &g);
uint32_t sgc_part_k2_shares(const sgc_part_geometry &g);     // workgroups per slice
uint32_t sgc_part_k2_direct_cols(const sgc_part_geometry &g, bool balanced);
void sgc_launch_part_k2(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_part_geometry &g,
                        uint64_t *pool, uint32_t *desc, uint32_t *counts, unsigned long long *matched, uint32_t dbg,
                        const sgc_runs *runs, const uint64_t *cuckoo /* two-choice image of the slices, or NULL */,
                        uint64_t *mrun /* with runs: buffer for the dense miss runs (as many records as the pool), or NULL */,
                        uint32_t *mcur /* its bump allocator, zeroed */,
                        bool direct_runs /* with mrun and tagged sub-partitions: the misses go straight to per-partition runs that are
                                            pass A's input (mrun: pool records << runs->sub_bits, inside runs->recs' allocation; the run
                                            matrices have shares + grid columns, or 2 x grid with slice_tot) */,
                        int slice_rec /* as given to sgc_launch_part_k1 */,
                        const uint32_t *slice_tot /* what k_partition added up (direct runs only), or NULL = static shares: the same
                                                     number of workgroups for every slice */,
                        uint32_t *slice_tot_next /* with slice_tot: SGC_SLICE_TOT words the kernel zeroes for the next pass */,
                        bool wide = true /* five-byte blocks of a 20-base library: 256 consecutive records per wave, 16-byte loads */);
void sgc_launch_part_k3(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm,
                        const sgc_bloom_view &bloom_lib, const sgc_bloom_view &bloom_perm, const sgc_part_geometry &g,
                        const uint64_t *pool, const uint32_t *desc, uint32_t *seg_cnt, uint32_t *gids, uint32_t dbg);
void sgc_launch_part_generic(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm,
                             const sgc_part_geometry &g, const uint64_t *pool, const uint32_t *desc, uint32_t *counts,
                             unsigned long long *matched);
void sgc_launch_part_k4(hipStream_t st, uint32_t n_guides, const sgc_part_geometry &g, const uint32_t *gids,
                        const uint32_t *seg_cnt, uint32_t *counts, unsigned long long *matched);

// ---- single-mismatch resolution in LDS (sgc_core.hip) -------------------------------------------
struct sgc_core_geometry {
    uint32_t w, grid_a, grid_b, pad_;       // w: producers of pass A's runs (the grid of k_count_slices)
    uint64_t runs_a_bytes, fwd_bytes, zero_bytes, small_bytes, mat_a, mat_b;
};
void sgc_core_plan(uint64_t n, const sgc_core_view &a, const sgc_core_view &b, uint32_t producers_a, sgc_core_geometry *g);
// buf0: runs_a_bytes (pass A's runs); buf1: fwd_bytes (pass A's forwarded runs); buf2: >= fwd_bytes, pass B's runs (the slice
// pool may serve: it is dead by then); zeroed: zero_bytes of zeros (stream-ordered before k_count_slices); small: small_bytes
sgc_runs sgc_core_runs_a(const sgc_core_geometry &g, const sgc_core_view &ca, uint32_t L, uint64_t *buf0, void *zeroed, void *small);
// pass: 0 = core A (reads buf0, forwards through buf1 into buf2), 1 = core B (reads buf2), 2 = the single exact-only pass of -x (reads buf0)
void sgc_launch_core(hipStream_t st, int pass, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm,
                     const sgc_core_view &ca, const sgc_core_view &cb, const uint64_t *amb, const sgc_core_geometry &g,
                     uint64_t *buf0, uint64_t *buf1, uint64_t *buf2, void *zeroed, void *small, uint32_t *counts,
                     unsigned long long *matched, uint32_t dbg);

void sgc_core_print_occupancy();
// -DSGC_STAMPS=1 builds: print and clear the workgroup timelines of the last pass (dbg 1048576; sgc_device.h)
void sgc_part_timeline_dump();
void sgc_core_timeline_dump();

// ---- device-side build of the single-mismatch table, its Bloom filter and the ambiguity masks (sgc_build.hip)
size_t sgc_device_build_scratch_bytes(uint32_t n, uint32_t L);
int sgc_device_build_permute(hipStream_t st, const uint64_t *d_keys, uint32_t n, uint32_t L, const sgc_table_view &lib,
                             uint64_t *d_slots, uint32_t log2_slots, uint32_t gid_bits, uint64_t *d_bloom,
                             uint32_t bloom_log2, uint64_t *d_amb, unsigned long long *d_entries, void *d_scratch);

// bit 31 of every guide id of a core index := "this guide has an ambiguous child" (after d_amb is complete on the stream)
void sgc_flag_ambiguous(hipStream_t st, uint32_t *d_gids, uint64_t n_entries, const uint64_t *d_amb);

// the children table of the byte-string path on the device (sgc_build.hip; the host builder is sgc_tables.cpp sgc_build_bytes_tables)
struct sgc_bytes_view;
bool sgc_device_bytes_children_supported(uint32_t n, uint32_t L);
size_t sgc_device_bytes_children_scratch(uint32_t n, uint32_t L);
int sgc_device_bytes_children(hipStream_t st, const sgc_bytes_view &v, uint64_t *perm_tag, uint32_t *perm_val, uint32_t *perm_pl, uint32_t perm_log2,
                              unsigned long long *d_entries, void *d_scratch);

// ---- FASTQ ingest (sgc_fastq.hip) -----------------------------------------------------------------
// tile_scratch: sgc_fastq_tiles(n) + 1 u32.  sgc_launch_fastq_count leaves the newlines before every tile there and
// tile_scratch[tiles] = number of '\n' in the text; sgc_launch_fastq_pack then writes the records of the part's
// sequence lines (sgc_fastq_records(first_line, n_lines) of them) and reports marker-byte errors through err[2].
uint32_t sgc_fastq_tiles(uint64_t n);
uint64_t sgc_fastq_records(uint64_t first_line, uint64_t n_lines);
void sgc_launch_fastq_count(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t *tile_scratch);
void sgc_launch_fastq_pack(hipStream_t st, const uint8_t *text, uint64_t n, const uint32_t *tile_scratch, uint64_t first_line,
                           uint32_t expect_nl, uint32_t n_lines, uint32_t L, bool rec16, int reverse, uint32_t o, int recursion,
                           uint64_t *recs, unsigned long long *err, uint32_t dbg);
void sgc_launch_pack_reads_lds(hipStream_t st, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L,
                               bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs);

// ---- generic byte-string path (sgc_bytes.hip; sgc_bytes.h) ------------------------------------------
struct sgc_bytes_view;
// starts/ends: sgc_fastq_records(first_line, n_lines) entries each, byte offsets of every sequence line of the part
void sgc_launch_fastq_lines(hipStream_t st, const uint8_t *text, uint64_t n, const uint32_t *tile_scratch, uint64_t first_line,
                            uint32_t expect_nl, uint32_t n_lines, uint64_t *starts, uint64_t *ends, unsigned long long *err);
void sgc_launch_bytes_count(hipStream_t st, const sgc_bytes_view &v, const uint8_t *text, const uint64_t *starts, const uint64_t *ends,
                            uint64_t n_reads, int reverse, uint32_t o, int recursion, bool one_mm, uint32_t *counts,
                            unsigned long long *matched, const uint8_t *flags = nullptr);
// hybrid libraries (sgc_bytes.hip): which reads need the byte-string chain over the whole library; their packed records are
// replaced by dead ones; the packed pass's counts (numbered over the ACGT guides) are folded through the guide map
void sgc_launch_bytes_route(hipStream_t st, const uint8_t *text, const uint64_t *starts, const uint64_t *ends, uint64_t n_reads, uint32_t L, bool rec16,
                            int reverse, uint32_t o, int recursion, uint64_t *recs, const sgc_bloom_view &shadow, uint8_t *flags);
void sgc_launch_fold_map(hipStream_t st, uint32_t *c32p, const uint32_t *map, unsigned long long *c64, uint32_t n);
void sgc_launch_bytes_lookup(hipStream_t st, const sgc_bytes_view &v, const uint8_t *tokens, uint64_t n, int which, bool one_mm, int32_t *out);
