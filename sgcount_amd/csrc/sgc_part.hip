// sgc_part.hip — the partitioned count path (variant 3), gfx950.
//
// Why: a random 8-byte gather served by the XCD L2 runs at ~0.26 T gathers/s chip-wide on MI355X and one
// served by the Infinity Cache at ~0.055 T/s (tools/ubench/gather.hip), and scattered device-scope atomics
// at ~0.02 T/s — while an LDS gather/atomic is an order of magnitude cheaper than any of them.  So the
// reads are first radix-partitioned by the hash of their Centered key; a workgroup then stages ONE slice of
// the library table (<= 64 KiB) in LDS and resolves and counts its partition there.  Only the reads that
// miss (Plus/Minus windows, single mismatch, 'N') — ~15 % of the synthetic mix — go on to global probes.
//
//   K1 k_partition     records -> blocks of B records, one partition per block.  Atomic-free: workgroup w
//                      owns the block range [w*M, (w+1)*M) and hands blocks to partitions as they fill; a
//                      descriptor per block records (partition, fill).  LDS stages each tile so that the
//                      global writes are contiguous runs.
//   K2 k_count_slices  workgroup (p, g): slice p of the library table -> LDS, per-slot LDS counters; every
//                      block of partition p: Centered-exact probe in LDS (Counter::assign's first and by far
//                      most frequent outcome, src/counter.rs:111); hits count in LDS, misses are compacted IN
//                      PLACE to the front of the block (descriptor fill := number of misses).  At the end the
//                      slot counters are flushed with one device-scope atomic per occupied slot.
//   K3 k_resolve_miss  the rest of the chain for the compacted misses (src/counter.rs:113-135) with global
//                      probes; the guide id replaces the record in place.
//   K4 k_hist_blocks   LDS histogram of those guide ids, slice by slice (as k_hist_slices).
//
// rec8 records and packed table slots only; other layouts use the generic kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_device.h"
#include "sgc_format.h"
#include "sgc_kernels.h"
#include "sgc_runs.h"

// s_memtime phase stamps + printf (dbg 512): compiled in only with -DSGC_STAMPS=1 (tools/evidence.sh builds such a library for
// profiles/*/stamps.txt); in the shipped kernels they would cost scalar registers the hot loops do not have
#ifndef SGC_STAMPS
#define SGC_STAMPS 0
#endif

#ifndef K1_THREADS
#define K1_THREADS 1024u        // one 16-wave workgroup per CU: few workgroups leave few half-empty blocks open
#endif
#define PART_TILE (K1_THREADS * 8u) // records staged per tile in K1 (8 per lane)
#ifndef PART_LOG2_BLOCK
#define PART_LOG2_BLOCK 9u
#endif
#define PART_BLOCK (1u << PART_LOG2_BLOCK) // records per block.  Every K1 workgroup leaves one part-filled block per partition: with
                                 // 1024-record blocks a sixth of all slots of a 100M-read pass is empty, and k_count_slices pays for
                                 // every block it walks, full or not (k1_wgs 256 instead of 512 halves the open blocks: K2 0.276 ->
                                 // 0.243 ms, but K1 then runs one workgroup per CU); smaller blocks do the same without that price
#ifndef PART_PAD
#define PART_PAD 0u              // experiment (DESIGN.md §6 "placement"): u64 words of padding behind every block of the pool
#endif
#define PART_STRIDE (PART_BLOCK + PART_PAD)      // distance between blocks of the pool, in 8-byte words
#define PART_MAXP 128u           // max library slices (~210k guides); partition index P_lib is the generic one
#define PART_ARR (PART_MAXP + 1u)
#define DESC_FILL_MASK 0xFFFFu

static_assert(PART_TILE % PART_BLOCK == 0 || PART_BLOCK % PART_TILE == 0, "tile/block sizes must nest");

// Partition of a record: the library slice of its Centered key, or — for records whose status is non-zero
// (an 'N', a dead window, a short read; ~2 % of reads) — the extra "generic" partition P_lib.  Those records
// need the serial generic chain (sgc_assign); kept apart, they are resolved by full waves in k_generic
// instead of stalling one or two lanes of almost every wave of the fast kernels.
// A clean record (status 0) has nothing above its span bits, so the partition kernel leaves the key's home slot
// inside its slice there (PART_TAG_SHIFT: 12 bits, 2 (L + 2) <= 50 for one-word records): k_count_slices then
// probes without hashing the key a second time — the 64-bit multiply is a dozen quarter-rate vector instructions,
// and that kernel is bound by vector issue.  Whoever hands a record on (the miss compaction) clears the tag.
// The misses of a slice block are compacted in place, but not to the block's first slots: every block is 8 KiB-aligned,
// so fronts that all start at offset 0 would land on the same few L2 / HBM channels (the first KiB of every 8 KiB) for
// the writer and for every reader.  The front of block b starts at record part_front(b) and wraps inside the block.
__device__ __forceinline__ uint32_t part_front(uint32_t b) { return ((b * 2654435761u) >> 26) << 4; }      // 64 starts, 128-B aligned
#define PART_TAG_SHIFT 52u
#define PART_SUB_SHIFT 50u        // below the bucket tag: which of the <= 4 partitions of core pass A inside the slice (sub_bits <= 2)
#define PART_TAG_MASK ((1ull << PART_SUB_SHIFT) - 1ull)
// With slices that follow the core hash (sgc_table_view::core_cl, sgc_home_bucket_ex) a slice's misses fall into the few
// partitions of core pass A that refine it, so the epilogue of k_count_slices writes a handful of streams, not hundreds.
// sub_bits (0..2, only with core-hashed slices): the next bits of the core hash below the slice — the partition of core
// pass A inside the slice — ride along too, so that k_count_slices can count its misses by that partition as it goes.
// MODE (compile time: the per-record code has no branches and touches nothing but its own record — with run-time
// modes the compiler copied the whole register tile around every branch, 32 moves per record):
//   0  slices by the full-key hash (sgc_home_slot_ex with core_cl == 0, or one slice), 8-byte records, slot tag
//   1  core-hashed slices, 8-byte records, slot tag + sub-partition tag
//   2  core-hashed slices, six-byte records (below): the home slot is not tagged (k_count_slices hashes the key again:
//      one multiply) and the sub-partition sits right above the span, so that a clean record is 2 (L + 2) + 2 <= 48 bits
//   3  core-hashed slices, FIVE-byte records (below): the slice index is a prefix of the mixed core value (sgc_core_mix, a
//      bijection), so a record inside a slice's block keeps only the rest of it in place of its core bases — 2 (L + 2) minus
//      the slice bits <= 40 bits; the sub-partition is the top of what is kept
// Returns the partition; `tag` is what to OR into the record (0 for the generic partition); MODE 3 replaces the record.
template <int MODE>
__device__ __forceinline__ uint32_t part_of(uint64_t &rec, uint64_t kmask, uint32_t sh, uint32_t log2_slots, uint32_t log2_slice,
                                            uint32_t core_cl, uint32_t sub_bits, uint64_t &tag) {
    const bool generic = (rec >> sh) != 0;
    const uint64_t key = (rec >> 2) & kmask;
    uint32_t p;
    if (MODE == 0) {
        const uint32_t hs = (uint32_t)(sgc_hash(key) >> (64 - log2_slots));
        tag = (uint64_t)(hs & ((1u << log2_slice) - 1u)) << PART_TAG_SHIFT;
        p = hs >> log2_slice;
    } else {
        // the slice and the sub-partition are prefixes of ONE mixed value of the core-A bases (log2_slots > log2_slice here)
        const uint32_t n = log2_slots - log2_slice, cb = 2u * core_cl;
        const uint32_t hm = sgc_core_mix((uint32_t)((key >> 2) & ((1ull << cb) - 1ull)), core_cl);
        const uint32_t part = hm >> (cb - n - sub_bits), sub = part & ((1u << sub_bits) - 1u);
        p = part >> sub_bits;
        if (MODE == 3) {
            tag = 0;
            // span bits [0, 4) | the mixed value below the slice index | span bits above the core, closed up
            if (!generic) rec = (rec & 0xFull) | ((uint64_t)(hm & ((1u << (cb - n)) - 1u)) << 4) | ((rec >> (4u + cb)) << (4u + cb - n));
        } else if (MODE == 2) tag = (uint64_t)sub << sh;
        else tag = ((uint64_t)(sgc_hash32(key) >> (32u - log2_slice)) << PART_TAG_SHIFT) | ((uint64_t)sub << PART_SUB_SHIFT);
    }
    if (generic) { tag = 0; p = 1u << (log2_slots - log2_slice); }
    return p;
}

// Six-byte slice blocks (p6; only with the direct-run path, where nothing but k_count_slices' probe loop reads them): a
// block of 1024 clean records is 1024 x u32 (low words) followed by 1024 x u16 (bits 32..47) in the first 6 KiB of its
// 8 KiB stride — a quarter less to write here and to read there, and both kernels run at what the memory system gives
// this access pattern.  Blocks of the generic partition keep whole 8-byte records (their status does not fit).
#define P6_HI_OFF (PART_BLOCK * 4u)
// Five-byte slice blocks (MODE 3): PART_BLOCK x u32 followed by PART_BLOCK x u8 (bits 32..39), 2.5 KiB of the block's stride.
// Records fall into the partitions evenly, so at any moment every open block of every workgroup of k_partition is filled to
// about the same level: tens of thousands of concurrent write streams of ~0.5 KB runs that all target the SAME offset inside
// their 4 KiB-aligned blocks — the same few memory channels (on physically contiguous memory, where nothing else scrambles
// the address bits, the kernel runs a quarter slower: DESIGN.md §6).  So record j of slice block b sits in slot
// (j + part_rot(b)) mod PART_BLOCK: the streams start at eight different 256-byte offsets, a whole wave still reads 64
// consecutive slots.
#ifndef PART_ROT
#define PART_ROT 0
#endif
__device__ __forceinline__ uint32_t part_rot(uint32_t b) { return PART_ROT ? ((b * 2654435761u) >> 29) << 6 : 0u; }

#if SGC_STAMPS
static __device__ sgc_tl_row tl_k1[SGC_TL_MAXWG], tl_k2loop[SGC_TL_MAXWG], tl_k2[SGC_TL_MAXWG];
#endif
void sgc_part_timeline_dump() {
#if SGC_STAMPS
    SGC_TIMELINE_DUMP(tl_k1, "K1"); SGC_TIMELINE_DUMP(tl_k2loop, "K2loop"); SGC_TIMELINE_DUMP(tl_k2, "K2");
#endif
}

// inclusive prefix sum over the 64 lanes of a wave with data-parallel primitives (row shifts inside the rows of 16 lanes, then the two
// row broadcasts of gfx9): six dependent vector instructions — the same scan written with __shfl_up goes through the LDS crossbar six
// times, 0.9 us per tile of k_partition with fifteen waves waiting for it (the phase stamps of round 4)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------------------------------ K1
template <int MODE>
__global__ void __launch_bounds__(K1_THREADS, 8) k_partition(const uint64_t *__restrict__ recs, uint64_t n, uint64_t per_wg,
                                                   uint32_t blocks_per_wg, uint32_t L, uint32_t log2_slots,
                                                   uint32_t log2_slice, uint32_t core_cl, uint32_t sub_bits, uint64_t *__restrict__ pool,
                                                   uint32_t *__restrict__ desc, uint32_t *__restrict__ tail,
                                                   uint32_t tail_words, uint32_t *__restrict__ wcnt,
                                                   uint32_t *__restrict__ wlist, unsigned long long *__restrict__ err, uint32_t dbg,
                                                   uint32_t *__restrict__ slice_tot) {
    constexpr bool p6 = MODE == 2, p5 = MODE == 3;
    SGC_TIMELINE_BEGIN(dbg);
    __shared__ uint64_t stage[PART_TILE];
    __shared__ uint8_t stage_p[PART_TILE];               // partition of every staged record
    // (the tile counts in two copies, used in turn: a tile's counts are zeroed while the tile before is staged, so that a tile needs three
    // barriers — ranked | scanned | staged — and neither one in front of its ranking nor one behind its writes: whoever enters the next
    // tile's scan or staging has passed that tile's first barrier, which every wave reaches only behind its own writes of this one)
    __shared__ uint32_t cnt2[2][PART_ARR], start[PART_ARR], blk[PART_ARR], fill[PART_ARR];
    __shared__ uint32_t nblk[PART_ARR];
    // where a partition's run of the current tile goes, in one 16-byte word per partition (one LDS read per record in the
    // write loop): x = first record of the run inside the staged tile, y = records that top up the open block, z = pool
    // position of the first of those MINUS x, w = pool position of the first record of the new block(s) MINUS (x + y)
    __shared__ uint4 runs[PART_ARR];
    __shared__ uint32_t next_free;
    const uint32_t P = 1u << (log2_slots - log2_slice), t = threadIdx.x;     // library slices; partition P = generic
    const uint32_t sh = 2 * (L + 2);
    const uint64_t kmask = sgc_key_mask(L);
    const uint64_t lo = (uint64_t)blockIdx.x * per_wg;
    const uint64_t hi = lo + per_wg < n ? lo + per_wg : n;
    const uint32_t block0 = blockIdx.x * blocks_per_wg;
    if (t < PART_ARR) { blk[t] = 0xFFFFFFFFu; fill[t] = PART_BLOCK; nblk[t] = 0; cnt2[0][t] = 0; cnt2[1][t] = 0; }
    if (t == 0) next_free = 0;
    __syncthreads();
    uint32_t tile_par = 0;
    // (-DSGC_STAMPS=1, dbg 1048576: thread 0's time in the phases of the tiles — loaded and ranked | scanned | staged | written; each ends at its barrier)
    unsigned long long k1_ph[4] = {0, 0, 0, 0}, k1_x = SGC_STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
#define K1_PHASE(i) if (SGC_STAMPS && (dbg & 1048576u)) { const unsigned long long y_ = __builtin_amdgcn_s_memtime(); k1_ph[i] += y_ - k1_x; k1_x = y_; }
    // two workgroups per CU (64 VGPRs each: __launch_bounds__(1024, 8)) — while one waits for its tile's records the other
    // ranks, stages or writes; a register prefetch of the next tile instead (one workgroup per CU, 80 VGPRs) measured slower
    for (uint64_t base = lo; base < hi; base += PART_TILE) {
        const uint32_t m = (uint32_t)(hi - base < PART_TILE ? hi - base : PART_TILE);
        uint32_t *const cnt = cnt2[tile_par];
        uint64_t rec[PART_TILE / K1_THREADS];
        uint32_t pr[PART_TILE / K1_THREADS];     // partition << 16 | rank inside the tile
#pragma unroll
        for (uint32_t k = 0; k < PART_TILE / K1_THREADS; k++) {
            const uint32_t j = k * K1_THREADS + t;
            if (j < m) rec[k] = __builtin_nontemporal_load(&recs[base + j]);
        }
#pragma unroll
        for (uint32_t k = 0; k < PART_TILE / K1_THREADS; k++) {
            const uint32_t j = k * K1_THREADS + t;
            if (j < m) {
                uint64_t tag;
                const uint32_t p = part_of<MODE>(rec[k], kmask, sh, log2_slots, log2_slice, core_cl, sub_bits, tag);
                rec[k] |= tag;
                pr[k] = (p << 16) | atomicAdd(&cnt[p], 1u);
            }
        }
        __syncthreads();
        K1_PHASE(0)
        // one lane per partition: exclusive scan of the tile counts, and where the tile's run goes.  A run
        // first tops up the partition's open block, the remainder opens a new one, so every closed block is
        // full and a workgroup never needs more than per_wg / BLOCK + P blocks.
        if (t < 64 || (t < PART_MAXP && P > 64) || t == PART_MAXP) {
            // lanes 0..127: library slices (scanned by wave 0, and by wave 1 when there are more than 64: it adds the
            // first 64 counts up for its base); lane 128 (wave 2): the generic partition, which starts where the
            // slices end
            const uint32_t q = t < PART_MAXP ? t : P, lane = t & 63u;
            const uint32_t c = (t < P || t == PART_MAXP) ? cnt[q] : 0;
            uint32_t st0;
            if (t < PART_MAXP) {
                uint32_t base = 0;
                const uint32_t incl = wave_inclusive_scan(c);
                if (t >= 64) {                           // wave 1: total of slices 0..63
                    base = cnt[lane];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) base += __shfl_xor(base, off, 64);
                }
                st0 = base + incl - c;
            } else {
                st0 = m - c;
            }
            if (t < P || t == PART_MAXP) {
                start[q] = st0;
                if (c) {
                    const uint32_t room = PART_BLOCK - fill[q];          // 0 when no block is open
                    const uint32_t head = c < room ? c : room;
                    uint4 rn = make_uint4(st0, head, (head ? blk[q] * PART_BLOCK + fill[q] : 0u) - st0, 0u);
                    fill[q] += head;
                    if (c > head) {
                        // the remainder goes to nb CONSECUTIVE new blocks (contiguous in the pool): all but the
                        // last are full
                        const uint32_t rem = c - head, nb = (rem + PART_BLOCK - 1) / PART_BLOCK;
                        if (blk[q] != 0xFFFFFFFFu) desc[blk[q]] = ((q + 1) << 16) | PART_BLOCK;    // close, full
                        const uint32_t x = block0 + atomicAdd(&next_free, nb);
                        for (uint32_t i = 0; i + 1 < nb; i++) desc[x + i] = ((q + 1) << 16) | PART_BLOCK;
                        // the partition's blocks from this workgroup, in order: k_count_slices walks these lists
                        // instead of scanning every descriptor
                        uint32_t *wl = wlist + ((size_t)blockIdx.x * PART_ARR + q) * blocks_per_wg + nblk[q];
                        // (block id << 11 | fill: every block is full except the one that is open when the workgroup ends, whose
                        // entry is corrected there — k_count_slices needs no second, dependent load of the descriptor)
                        for (uint32_t i = 0; i < nb; i++) wl[i] = ((x + i) << 11) | PART_BLOCK;
                        nblk[q] += nb;
                        blk[q] = x + nb - 1;
                        rn.w = x * PART_BLOCK - (st0 + head);
                        fill[q] = rem - (nb - 1) * PART_BLOCK;
                    }
                    runs[q] = rn;
                }
            }
        }
        __syncthreads();
        K1_PHASE(1)
#pragma unroll
        for (uint32_t k = 0; k < PART_TILE / K1_THREADS; k++) {
            const uint32_t j = k * K1_THREADS + t;
            if (j < m) {
                const uint32_t at = start[pr[k] >> 16] + (pr[k] & 0xFFFFu);
                stage[at] = rec[k];
                stage_p[at] = (uint8_t)(pr[k] >> 16);
            }
        }
        if (t < PART_ARR) cnt2[tile_par ^ 1u][t] = 0;        // the next tile's counts (last read by the scan of the tile before this one)
        tile_par ^= 1u;
        __syncthreads();
        K1_PHASE(2)
        for (uint32_t j = t; j < m; j += K1_THREADS) {
            const uint64_t r = stage[j];
            const uint32_t p = stage_p[j];
            const uint4 rn = runs[p];
            const uint32_t at = j + (j - rn.x < rn.y ? rn.z : rn.w);       // record index in the pool (< 2^29: byte offsets fit 32 bits)
            char *pb = reinterpret_cast<char *>(pool);
            if (!SGC_BOUND((at >> PART_LOG2_BLOCK) < gridDim.x * blocks_per_wg, err, 8)) continue;
            if ((p6 || p5) && p != P) {
                const uint32_t bo = (at >> PART_LOG2_BLOCK) * (PART_STRIDE * 8u), idx = (at + part_rot(at >> PART_LOG2_BLOCK)) & (PART_BLOCK - 1u);
                // (timing-only ablations, -DSGC_ABLATE=1: dbg 64 no high-byte store, dbg 128 no low-word store — what a second write stream per run costs)
                if (!SGC_DBG(dbg, 128u)) *reinterpret_cast<uint32_t *>(pb + (size_t)(bo + (idx << 2))) = (uint32_t)r;
                if (p5 && !SGC_DBG(dbg, 64u)) *reinterpret_cast<uint8_t *>(pb + (size_t)(bo + P6_HI_OFF + idx)) = (uint8_t)(r >> 32);
                else *reinterpret_cast<uint16_t *>(pb + (size_t)(bo + P6_HI_OFF + (idx << 1))) = (uint16_t)(r >> 32);
            } else {
                *reinterpret_cast<uint64_t *>(pb + (size_t)(((at >> PART_LOG2_BLOCK) * PART_STRIDE + (at & (PART_BLOCK - 1u))) << 3)) = r;
            }
        }
        K1_PHASE(3)
    }
#undef K1_PHASE
    __syncthreads();
    if (t <= P && blk[t] != 0xFFFFFFFFu) {
        desc[blk[t]] = ((t + 1) << 16) | fill[t];
        wlist[((size_t)blockIdx.x * PART_ARR + t) * blocks_per_wg + nblk[t] - 1u] = (blk[t] << 11) | fill[t];
    }
    // the blocks this workgroup never handed out read as "no partition, no records"; workgroup 0 also clears the
    // scratch counters behind the descriptors (no separate memset on the stream)
    for (uint32_t i = next_free + t; i < blocks_per_wg; i += K1_THREADS) desc[block0 + i] = 0;
    if (t < PART_ARR) wcnt[blockIdx.x * PART_ARR + t] = nblk[t];
    // blocks per slice over all workgroups: k_count_slices cuts equal shares of the whole from them (zeroed by the k_count_slices
    // of the pass before)
    if (slice_tot && t < P && nblk[t]) atomicAdd(&slice_tot[t], nblk[t]);
    if (blockIdx.x == 0)
        for (uint32_t i = t; i < tail_words; i += K1_THREADS) tail[i] = 0;
    SGC_TIMELINE_END4(dbg, tl_k1, next_free, k1_ph[0], k1_ph[1], k1_ph[2], k1_ph[3]);
}

// ------------------------------------------------------------------------------------------------ K2
#define K2_THREADS 1024u
#define K2_LIST 2048u            // capacity of the per-round block list
#ifndef K2_U
#define K2_U 8u 
#endif
//                 // blocks per group: one group is processed while the next is in flight
#ifndef K2_WU
#define K2_WU 1u                 // WIDE: steps (four records per lane each) per group: 1, 2, 3 measure the same (0.216-0.228 ms), 4 spills
#endif
#define K2_SCAN 16u              // descriptors examined per lane per scan chunk
#define K2_GLIST 64u             // generic blocks listed per epilogue window
// record j of slice block b
// REC: 0 = 8-byte records, 1 = six-byte, 2 = five-byte blocks
template <int REC>
__device__ __forceinline__ uint64_t k2_record(const uint64_t *__restrict__ pool, uint32_t b, uint32_t j) {
    if (REC == 0) return pool[(uint64_t)b * PART_STRIDE + j];
    const char *bb = reinterpret_cast<const char *>(pool) + (uint64_t)b * (PART_STRIDE * 8u);
    j = (j + part_rot(b)) & (PART_BLOCK - 1u);
    const uint32_t lo = reinterpret_cast<const uint32_t *>(bb)[j];
    const uint32_t hi = REC == 2 ? (uint32_t)reinterpret_cast<const uint8_t *>(bb + P6_HI_OFF)[j] : (uint32_t)reinterpret_cast<const uint16_t *>(bb + P6_HI_OFF)[j];
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// balanced shares (k_count_slices): a sequence of T blocks dealt to n workgroups in order, T / n blocks each and one more for the
// first T % n — so the workgroups with nothing to do (T < n) come last, and between the first and the last workgroup of a slice
// every workgroup holds blocks of it.  k2_share_lo: first block of workgroup j (j <= n); k2_owner: the workgroup that holds block x (< T)
__device__ __forceinline__ uint32_t k2_share_lo(uint32_t j, uint32_t n, uint32_t T) {
    const uint32_t base = T / n, rem = T % n;
    return j * base + (j < rem ? j : rem);
}
__device__ __forceinline__ uint32_t k2_owner(uint32_t x, uint32_t n, uint32_t T) {
    const uint32_t base = T / n, rem = T % n, big = rem * (base + 1u);
    return x < big ? x / (base + 1u) : rem + (x - big) / base;        // (x >= big implies base > 0: big = T when base = 0)
}

// workgroups whose shares hold blocks of slice p (spre: first block of every slice in the sequence)
__device__ __forceinline__ uint32_t k2_slice_wgs(const uint32_t *spre, uint32_t p, uint32_t n, uint32_t T) {
    const uint32_t b0 = spre[p], b1 = spre[p + 1u];
    return b1 > b0 ? k2_owner(b1 - 1u, n, T) - k2_owner(b0, n, T) + 1u : 0u;
}

// CUCKOO: the slice is staged from its two-choice image (`cuck`, sgc_format.h sgc_cuckoo_alt): both candidate slots are
// read at once (two 8-byte LDS reads) and the probe has no loop.
// DENSE (only with ep.recs): the misses do not go back into their blocks but, densely, into a stretch of `mrun` the workgroup
// takes at its start (as many records as its blocks hold: no miss can lack room).  Compacting in place needs a barrier per
// group of blocks — a wave must not overwrite slots another wave has yet to load — and the descriptors rewritten; a dense run
// needs neither (the position comes from ONE LDS atomic per wave step, after a ballot), and the epilogue reads it as one
// contiguous range instead of a thousand-odd fronts.
// DIRECT (with DENSE, only when k_partition tagged the sub-partitions): the stretch is cut into one run per partition of core
// pass A inside the slice (<= 4), every miss goes straight to the run of its partition (ONE LDS atomic per record: the position,
// which is also the count), and those runs ARE pass A's input — published as producer column g (the workgroup's index inside
// its slice) of the run matrices, nothing is read back or moved.  Only the workgroup's share of the generic blocks still goes
// through the epilogue's histogram + placement, as producer column G + blockIdx.x.  mrun must be ep.recs + (an offset that
// fits 32 bits); ep.W = G + gridDim.x.
// REC (with DIRECT): 1 = the slice blocks hold six-byte records (k_partition, P6_HI_OFF), 2 = five-byte records: a record is
// its span with the core-A bases replaced by the mixed core value below the slice index; the span comes back with one multiply
// (sgc_core_unmix) — this kernel waits for its pool bytes, not for its arithmetic.
// LT: 20 = the geometry of a 20-base library with 64 full slices as compile-time constants (guide length, core length 9, six slice
// bits, two sub-partition bits): the record decode and the probe become shifts and masks by immediates instead of 64-bit shifts
// by scalars — the loop of this kernel is as much vector issue as it is memory; 0 = everything from the arguments.
// WIDE (five-byte blocks, direct runs, two-choice image): a wave takes 256 CONSECUTIVE records of a block — lane l the four
// records 4 l .. 4 l + 3, one 16-byte load (their low words) and one 4-byte load (their high bytes) per lane — instead of one
// record per lane from eight different blocks (a 4-byte and a 1-byte load per record).  The same bytes in a quarter of the
// vector-memory instructions: what bounded the narrow loop was the number of memory instructions a CU keeps in flight, not the
// bytes (DESIGN.md §6 "What bounds k_count_slices", round 4).
template <int LOG2_SLICE, bool CUCKOO, bool DENSE, bool DIRECT, int REC, int LT = 0, bool WIDE = false>
__global__ void __launch_bounds__(K2_THREADS, 8) __attribute__((amdgpu_num_sgpr(80))) k_count_slices(uint64_t *__restrict__ pool, uint32_t *__restrict__ desc,
                                                             const uint32_t *__restrict__ wcnt, const uint32_t *__restrict__ wlist,
                                                             uint32_t k1_wgs, uint32_t blocks_per_wg, uint32_t G, uint32_t L_arg,
                                                             sgc_table_view lib, uint32_t *__restrict__ counts,
                                                             unsigned long long *__restrict__ matched, uint32_t dbg,
                                                             const sgc_runs ep, const uint64_t *__restrict__ cuck,
                                                             uint64_t *__restrict__ mrun, uint32_t *__restrict__ mcur,
                                                             const uint32_t *__restrict__ slice_tot, uint32_t *__restrict__ slice_tot_next) {
    constexpr uint32_t S = 1u << LOG2_SLICE;
    SGC_TIMELINE_BEGIN(dbg);
    constexpr bool P6 = REC != 0;                            // packed slice blocks: no slot tag in the record, the key is hashed here
    const uint32_t L = LT ? (uint32_t)LT : L_arg;
    // (LT: the launcher checked that the arguments say the same)
    const uint32_t core_cl = LT ? ((uint32_t)LT - 2u) / 2u : lib.core_cl, slice_bits = LT ? 6u : lib.log2_slots - lib.log2_slice;
    const uint32_t sub_bits = LT ? 2u : ep.sub_bits;
    // A step of the workgroup takes BPS blocks side by side: lanes [h PART_BLOCK, (h + 1) PART_BLOCK) take record jl of the
    // h-th of them (h is wave-uniform: a block is a whole number of waves); a group is K2_U steps.
    static_assert(PART_BLOCK <= K2_THREADS && PART_BLOCK >= 64u, "a block is 1..16 waves of the workgroup");
    constexpr uint32_t BPS = K2_THREADS / PART_BLOCK;        // blocks per step
    constexpr uint32_t Q = K2_U;                             // records per thread per group
    __shared__ ulonglong2 tab[S / 2];                        // the slice, bucket by bucket
    __shared__ uint32_t cnt[S];
    __shared__ uint32_t list[K2_LIST];                       // block id << 11 | (fill - 1)
    __shared__ uint32_t miss_cnt[2][K2_U * BPS], scratch[128], pre[K2_THREADS], wtmp[17];
    __shared__ uint32_t hn[RUN_MAXP], rcur[RUN_MAXP], rbase, preg[K2_THREADS];   // epilogue: leftovers by partition of core pass A
    __shared__ uint32_t wmiss, mbase;                                              // DENSE: misses so far, start of the stretch in mrun
    __shared__ uint32_t wmiss4[4];                                                 // DIRECT: misses so far, by sub-partition
    const uint32_t t = threadIdx.x;
    const uint32_t h = __builtin_amdgcn_readfirstlane(t / PART_BLOCK), jl = t % PART_BLOCK;
    const bool count_sub = ep.recs != nullptr && ep.sub_bits != 0xFFu;      // wave-uniform
    const uint32_t slice = (!LT && lib.log2_slice < (uint32_t)LOG2_SLICE) ? (1u << lib.log2_slice) : S;   // small libraries
    const uint32_t bmask = slice / 2 - 1u, gid_bits = lib.gid_bits;
    const uint64_t kmask = sgc_key_mask(L);
    const uint64_t *gslots = CUCKOO ? cuck : lib.slots;                // where the slice's slots (key << gid_bits | gid) live
    const uint32_t ls = (!LT && lib.log2_slice < (uint32_t)LOG2_SLICE) ? lib.log2_slice : (uint32_t)LOG2_SLICE;   // log2 slots per slice
    const uint64_t *tab1 = reinterpret_cast<const uint64_t *>(tab);                                       // the same keys, slot by slot
    for (uint32_t i = t; i < RUN_MAXP; i += K2_THREADS) hn[i] = 0;
    if (t < 2 * K2_U * BPS) miss_cnt[t / (K2_U * BPS)][t % (K2_U * BPS)] = 0;
    // diagnostic stamps (SGC_STAMPS && (dbg & 512)): cycle counts of the phases of a few workgroups, printed at the end
    unsigned long long ts0 = 0, ts_scan = 0, ts_loop = 0, ts_flush = 0;
    uint32_t n_groups_dbg = 0;
    if (SGC_STAMPS && (dbg & (512u | 1048576u))) ts0 = __builtin_amdgcn_s_memtime();
    // Which blocks of which slice.  K1 workgroup w handed slice p wcnt[w][p] blocks, listed in wlist[w][p][]: in that order
    // (w major) the slice's blocks form one sequence.
    //   static (slice_tot == nullptr): workgroup (p, g) = (blockIdx / G, blockIdx % G) takes the g-th of G equal shares of
    //     slice p — fine while the slices hold the same number of blocks;
    //   balanced (DIRECT, with k_partition's per-slice block totals): the sequences of all slices, one after the other, are
    //     ONE sequence of which every workgroup takes an equal share — a workgroup whose share crosses from one slice into
    //     the next works through two (rarely more) SEGMENTS, each with its own staging of the slice, flush of the counters
    //     and column of the run matrices.  Guides are not read equally often (a few hundred hot guides fall unevenly into 64
    //     slices: +-15 % of blocks per slice on the bench workload, any factor on a sample that a few guides dominate), and
    //     the kernel ends with its slowest workgroup.
    const bool balanced = DIRECT && slice_tot != nullptr;                // wave-uniform (kernel argument)
    const uint32_t n_slices = 1u << (lib.log2_slots - lib.log2_slice), NW = gridDim.x;
    __shared__ uint32_t spre[PART_MAXP + 1];                             // balanced: first block of every slice in the global sequence
    uint32_t gx = 0, gx_hi = 0, T_all = 0;
    if (balanced) {
        const uint32_t v = t < n_slices ? slice_tot[t] : 0u;
        const uint32_t e = wg_scan_1024(v, wtmp, &T_all);
        if (t <= n_slices) spre[t] = e;                                  // spre[n_slices] = T_all
        if (blockIdx.x == 0 && t < PART_ARR) slice_tot_next[t] = 0;      // the next pass's k_partition adds its totals up there
        __syncthreads();
        gx = k2_share_lo(blockIdx.x, NW, T_all);
        gx_hi = k2_share_lo(blockIdx.x + 1u, NW, T_all);
        // A row of the run matrices (a partition of core pass A inside slice p) holds, without gaps — the consumer walks over
        // empty columns one by one —: the G_p workgroups whose shares meet slice p (k2_slice_wgs), then every workgroup's part
        // of the generic blocks, then the G - G_p columns nobody fills: workgroup p clears those.
        if (blockIdx.x < n_slices) {
            const uint32_t used = k2_slice_wgs(spre, blockIdx.x, NW, T_all);
            const uint32_t rows = 1u << sub_bits, width = G - used;
            for (uint32_t i = t; i < rows * width; i += K2_THREADS)
                ep.cnt[(size_t)((blockIdx.x << sub_bits) + i / width) * ep.W + used + NW + i % width] = 0;
        }
    }
    uint64_t local = 0;                                                    // reads this lane's slots counted (all segments)
    bool first_seg = true;
    uint32_t p = 0, g = 0, s_lo = 0, s_hi = 0, run0 = 0, stretch = 0;
    for (;;) {
    if (balanced) {
        if (gx >= gx_hi) break;
        p = find_extent<7>(spre, n_slices, gx);                                   // last slice that starts at or before block gx: the one that holds it
        s_lo = gx - spre[p];
        s_hi = (gx_hi < spre[p + 1] ? gx_hi : spre[p + 1]) - spre[p];
        g = blockIdx.x - k2_owner(spre[p], NW, T_all);
        gx = spre[p] + s_hi;
    } else {
        if (!first_seg) break;
        p = blockIdx.x / G; g = blockIdx.x % G;
    }
    __syncthreads();                                                     // the previous segment is done with tab[], cnt[], pre[], wmiss4[]
    // What a segment needs from memory before its first record — the block counts of the slice and the slice itself — is requested
    // together and waited for once (the staging loop written load, store, load, store runs in that order: a round trip per iteration).
    const uint32_t wc = t < k1_wgs ? wcnt[t * PART_ARR + p] : 0u;
    {
        const ulonglong2 *gtab = reinterpret_cast<const ulonglong2 *>(gslots) + (uint64_t)p * (slice / 2);
        constexpr uint32_t NI = S / 2 / K2_THREADS;
        ulonglong2 v[NI];
#pragma unroll
        for (uint32_t k = 0; k < NI; k++) {
            const uint32_t i = k * K2_THREADS + t;
            v[k] = i < slice / 2 ? gtab[i] : make_ulonglong2(SGC_EMPTY, SGC_EMPTY);
        }
#pragma unroll
        for (uint32_t k = 0; k < NI; k++) {      // bare keys in LDS (a key is < 2^60, so SGC_EMPTY stays distinct)
            if (v[k].x != SGC_EMPTY) v[k].x >>= gid_bits;
            if (v[k].y != SGC_EMPTY) v[k].y >>= gid_bits;
            tab[k * K2_THREADS + t] = v[k];
        }
    }
    for (uint32_t i = t; i < S; i += K2_THREADS) cnt[i] = 0;
    {
        uint32_t Bp;
        pre[t] = wg_scan_1024(wc, wtmp, &Bp);
        __syncthreads();
        if (!balanced) { s_lo = (uint32_t)((uint64_t)Bp * g / G); s_hi = (uint32_t)((uint64_t)Bp * (g + 1) / G); }
        else if (!SGC_BOUND(s_hi <= Bp, reinterpret_cast<unsigned long long *>(matched) + 3, 12)) s_hi = s_lo;      // k_partition's totals and its lists disagree
    }
    first_seg = false;
    run0 = 0;
    stretch = (s_hi - s_lo) * PART_BLOCK;                         // no run can lack room: as many records as all blocks of the share hold
    if (DENSE) {
        if (t < 4) wmiss4[t] = 0;
        // where the runs go: with balanced shares the segments cut the sequence of all slice blocks into disjoint pieces, so the
        // piece's first block IS a bump allocator's answer (no atomic on one word from every workgroup of the launch)
        if (t == 0) { wmiss = 0; mbase = balanced ? ((spre[p] + s_lo) * PART_BLOCK) << sub_bits
                                                  : (s_hi > s_lo ? atomicAdd(mcur, DIRECT ? stretch << sub_bits : stretch) : 0u); }
        __syncthreads();
        run0 = mbase;
    }
    for (uint32_t win = s_lo; win < s_hi; win += K2_LIST) {
        const uint32_t nl = s_hi - win < K2_LIST ? s_hi - win : K2_LIST;
        __syncthreads();                               // the previous round is done with list[]
        for (uint32_t i = t; i < nl; i += K2_THREADS) {
            const uint32_t o = win + i, w = find_extent<10>(pre, k1_wgs, o);
            const uint32_t e = wlist[((size_t)w * PART_ARR + p) * blocks_per_wg + (o - pre[w])];       // block id << 11 | fill
            uint32_t b = e >> 11;
            unsigned long long *const err = reinterpret_cast<unsigned long long *>(matched) + 3;      // the sample's flag word behind `matched`
            if (!SGC_BOUND(b < k1_wgs * blocks_per_wg && w < k1_wgs, err, 9)) b = 0;
            const uint32_t fl = e & 2047u;
            list[i] = (b << 11) | ((SGC_BOUND(fl >= 1u && fl <= PART_BLOCK, err, 10) ? fl : 1u) - 1u);
        }
        __syncthreads();
        if (SGC_STAMPS && (dbg & (512u | 1048576u))) { ts_scan += __builtin_amdgcn_s_memtime() - ts0; ts0 = __builtin_amdgcn_s_memtime(); n_groups_dbg += (nl + K2_U * (K2_THREADS / PART_BLOCK) - 1) / (K2_U * (K2_THREADS / PART_BLOCK)); }
        if constexpr (WIDE) {
            static_assert(!WIDE || (REC == 2 && DIRECT && CUCKOO && PART_ROT == 0 && PART_BLOCK % 256u == 0), "wide loads: five-byte blocks, direct runs, two-choice image");
            constexpr uint32_t QPB = PART_BLOCK / 256u;                 // 256-record pieces per block
            constexpr uint32_t BPW = (K2_THREADS / 64u) / QPB;          // blocks per step of the workgroup
            constexpr uint32_t WU = K2_WU;                              // steps per group: one group is processed while the next is in flight
            const uint32_t wvi = __builtin_amdgcn_readfirstlane(t >> 6);
            const uint32_t hb = wvi / QPB, j0 = (wvi % QPB) * 256u + 4u * (t & 63u);
            const uint32_t nsteps = (nl + BPW - 1u) / BPW;
            const char *const pb = reinterpret_cast<const char *>(pool);
            // timing-only ablations (-DSGC_ABLATE=1; the table is wrong with any of them): dbg 16 the loads alone (every record dropped behind its
            // load: no unpacking, no LDS, no store), dbg 32 every step reads the share's FIRST blocks again (real records of the slice, served
            // by the L2: the probe without its HBM traffic), 4 no miss stores
#define K2W_ENTRY(s) __builtin_amdgcn_readfirstlane((s) * BPW + hb < nl ? list[SGC_DBG(dbg, 32u) ? hb : (s) * BPW + hb] : 0xFFFFFFFFu)
            // (unconditional loads: a lane past the block's fill reads the block's first 16 bytes again — a line that is on its way
            // anyway — and a wave without a block the pool's; loads inside branches make the compiler wait for every load in flight)
#define K2W_LOAD(e, lo, hi)                                                                                          \
            {                                                                                                        \
                const uint32_t e_ = (e);                                                                             \
                const char *bb_ = pb + (e_ != 0xFFFFFFFFu ? (uint64_t)(e_ >> 11) * (PART_STRIDE * 8u) : 0ull);       \
                const uint32_t jj_ = (e_ != 0xFFFFFFFFu && j0 <= (e_ & 2047u)) ? j0 : 0u;                            \
                lo = *reinterpret_cast<const uint4 *>(bb_ + 4u * jj_);                                               \
                hi = *reinterpret_cast<const uint32_t *>(bb_ + P6_HI_OFF + jj_);                                     \
            }
            uint4 clo[WU], nlo[WU];
            uint32_t chi[WU], nhi[WU], ce[WU];
#pragma unroll
            for (uint32_t u = 0; u < WU; u++) { ce[u] = K2W_ENTRY(u); K2W_LOAD(ce[u], clo[u], chi[u]) }
            for (uint32_t li = 0; li < nsteps; li += WU) {
#pragma unroll
                for (uint32_t u = 0; u < WU; u++) K2W_LOAD(K2W_ENTRY(li + WU + u), nlo[u], nhi[u])
#pragma unroll
                for (uint32_t u = 0; u < WU; u++) {
                    const uint32_t lw[4] = {clo[u].x, clo[u].y, clo[u].z, clo[u].w};
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const bool valid = ce[u] != 0xFFFFFFFFu && j0 + k <= (ce[u] & 2047u);
                        if (SGC_DBG(dbg, 16u)) { if (lw[k] == 0x12345678u && chi[u] == 0x9ABCDEFu) local++; continue; }      // (the load must not be optimised away)
                        // five-byte record -> span (as in the narrow loop below)
                        const uint32_t cb = 2u * core_cl, kb = cb - slice_bits;
                        const uint32_t raw_lo = lw[k];
                        const uint64_t raw = (uint64_t)raw_lo | ((uint64_t)((chi[u] >> (8u * k)) & 0xFFu) << 32);
                        const uint32_t hmix = (p << kb) | ((raw_lo >> 4) & ((1u << kb) - 1u));
                        const uint32_t corev = sgc_core_unmix(hmix, core_cl);
                        const uint64_t span = (uint64_t)(raw_lo & 0xFu) | ((uint64_t)corev << 4) | ((raw >> (4u + kb)) << (4u + cb));
                        const uint32_t sub = (hmix >> (kb - sub_bits)) & ((1u << sub_bits) - 1u);
                        const uint64_t key = (span >> 2) & kmask;
                        const uint32_t h32 = sgc_hash32(key);
                        const uint32_t s1 = h32 >> (32u - ls), s2 = sgc_cuckoo_alt_h(h32, s1, ls);
                        const uint64_t e1 = tab1[s1], e2 = tab1[s2];
                        const bool h2 = e2 == key, hit = e1 == key || h2;
                        const uint32_t slot = h2 ? s2 : s1;
                        const bool hv = valid && hit, mv = valid && !hit;
                        atomicAdd(hv ? &cnt[slot] : &scratch[t & 63u], 1u);
                        const uint32_t pos = atomicAdd(mv ? &wmiss4[sub] : &scratch[64u + (t & 63u)], 1u);
                        if (mv && !SGC_DBG(dbg, 4u) && SGC_BOUND(pos < stretch, reinterpret_cast<unsigned long long *>(matched) + 3, 11)) mrun[(uint64_t)run0 + sgc_mul24(sub, stretch) + pos] = span;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < WU; u++) { clo[u] = nlo[u]; chi[u] = nhi[u]; ce[u] = K2W_ENTRY(li + WU + u); }
            }
#undef K2W_LOAD
#undef K2W_ENTRY
            __syncthreads();
            if (SGC_STAMPS && (dbg & (512u | 1048576u))) { ts_loop += __builtin_amdgcn_s_memtime() - ts0; ts0 = __builtin_amdgcn_s_memtime(); }
            continue;
        }
        // software pipeline over groups of K2_U blocks: `cur` is processed while `nxt` is in flight.  Block ids and
        // fills are wave-uniform (scalar registers).
        uint64_t cur[Q], nxt[Q];
        uint32_t ce[K2_U];                // list entries (block id << 11 | fill - 1; all ones = none), wave-uniform
        // list entry of step s for this lane's block group (li counts steps, the list counts blocks)
#define K2_ENTRY(s) __builtin_amdgcn_readfirstlane((s) * BPS + h < nl ? list[(s) * BPS + h] : 0xFFFFFFFFu)
        const uint32_t nsteps = (nl + BPS - 1u) / BPS;
#pragma unroll
        for (uint32_t u = 0; u < K2_U; u++) {
            ce[u] = K2_ENTRY(u);
            // lanes past the block's fill have nothing to read (one open block per K1 workgroup and slice is part empty)
            cur[u] = (ce[u] != 0xFFFFFFFFu && jl <= (ce[u] & 2047u)) ? k2_record<REC>(pool, ce[u] >> 11, jl) : 0ull;
        }
        for (uint32_t li = 0; li < nsteps; li += K2_U) {
            const uint32_t par = (li / K2_U) & 1u;
            // all records of group li are in registers (this also publishes the previous group's miss counts)
            if (!DENSE) __syncthreads();
            if (!DENSE && t < K2_U * BPS) {
                if (li) {
                    const uint32_t e = (li - K2_U) * BPS + t < nl ? list[(li - K2_U) * BPS + t] : 0xFFFFFFFFu;
                    if (e != 0xFFFFFFFFu) desc[e >> 11] = ((p + 1) << 16) | miss_cnt[par ^ 1u][t];
                }
                miss_cnt[par ^ 1u][t] = 0;
            }
#pragma unroll
            for (uint32_t u = 0; u < K2_U; u++) {
                const uint32_t ne = K2_ENTRY(li + K2_U + u);
                nxt[u] = (ne != 0xFFFFFFFFu && jl <= (ne & 2047u)) ? k2_record<REC>(pool, ne >> 11, jl) : 0ull;
            }
            // Centered-exact probe (src/counter.rs:111) of the Q records against the slice in LDS.  The LDS copy
            // holds bare keys, so a bucket resolves with four 64-bit compares and no branches (K2 is bound by
            // instruction issue — one scalar unit per CU — not by memory); the rare longer chains are walked in
            // a loop.  An insert fills slot 0 before slot 1, so "slot 1 free" means "chain ends here".
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) {
                // K1 sends every record with a non-zero status to the generic partition, so a slice partition
                // only holds clean records: no status test here
                const uint32_t u = q;
                const bool valid = ce[u] != 0xFFFFFFFFu && jl <= (ce[u] & 2047u);
                if (REC == 2) {
                    // five-byte record -> span: bits [0, 4) | core value << 4 | the rest above; the core value is the inverse of
                    // (slice index : kept bits) under the mixing multiply
                    const uint32_t cb = 2u * core_cl, kb = cb - slice_bits;
                    const uint32_t raw_lo = (uint32_t)cur[q];
                    const uint32_t hmix = (p << kb) | ((raw_lo >> 4) & ((1u << kb) - 1u));
                    const uint32_t corev = sgc_core_unmix(hmix, core_cl);
                    cur[q] = (uint64_t)(raw_lo & 0xFu) | ((uint64_t)corev << 4) | ((cur[q] >> (4u + kb)) << (4u + cb)) |
                             ((uint64_t)((hmix >> (kb - sub_bits)) & ((1u << sub_bits) - 1u)) << (2u * (L + 2u)));      // + the sub-partition where the six-byte record has it
                }
                const uint64_t key = (cur[q] >> 2) & kmask;
                // home slot inside the slice: left there by k_partition, or (six-byte records) hashed again here
                const uint32_t h32 = sgc_hash32(key);
                // (k_partition's slot tag has twelve bits: with slices of 2^13 slots the slot is computed here, whatever the record format)
                const uint32_t s1 = P6 ? h32 >> (32u - ls) :
                                    LOG2_SLICE > 12 ? (sgc_home_slot_ex(key, lib.log2_slots, lib.log2_slice, lib.core_cl) & (slice - 1u)) : (uint32_t)(cur[q] >> PART_TAG_SHIFT);
                uint32_t b = s1 >> 1, slot;
                ulonglong2 wv;
                bool hit;
                if (CUCKOO) {
                    // the key is in its home slot or in the alternate one: read both, no chain
                    const uint32_t s2 = sgc_cuckoo_alt_h(h32, s1, ls);
                    const uint64_t e1 = tab1[s1], e2 = tab1[s2];
                    const bool h2 = e2 == key;
                    hit = e1 == key || h2;
                    slot = h2 ? s2 : s1;
                } else {
                    wv = tab[b];
                    hit = wv.x == key || wv.y == key;
                    bool cont = valid && !hit && wv.y != SGC_EMPTY;
                    while (cont) {
                        b = (b + 1) & bmask;
                        wv = tab[b];
                        hit = wv.x == key || wv.y == key;
                        cont = !hit && wv.y != SGC_EMPTY;
                    }
                    slot = 2 * b + (wv.x == key ? 0u : 1u);
                }
                // Predicated, not branched (every exec-mask change costs scalar instructions, and K2 is bound by
                // its scalar unit): lanes with nothing to add hit a scratch word of their own.
                const bool hv = valid && hit, mv = valid && !hit;
                atomicAdd(hv ? &cnt[slot] : &scratch[t & 63u], 1u);
                // with K1's sub-partition tag the misses are counted by pass A's partition right here, and the epilogue's
                // histogram sweep does not have to read the fronts once more
                if (count_sub && !DIRECT) atomicAdd(mv ? &hn[(p << sub_bits) | ((uint32_t)(cur[q] >> PART_SUB_SHIFT) & 3u)] : &scratch[t & 63u], 1u);
                if (DIRECT) {
                    const uint32_t sub = (uint32_t)(cur[q] >> (P6 ? 2u * (L + 2u) : PART_SUB_SHIFT)) & 3u;
                    const uint32_t pos = atomicAdd(mv ? &wmiss4[sub] : &scratch[64u + (t & 63u)], 1u);
                    if (mv && SGC_BOUND(pos < stretch, reinterpret_cast<unsigned long long *>(matched) + 3, 11)) mrun[(uint64_t)run0 + sgc_mul24(sub, stretch) + pos] = cur[q] & (P6 ? (1ull << (2u * (L + 2u))) - 1ull : PART_TAG_MASK);
                } else if (DENSE) {
                    // (a ballot + one atomic by the lowest missing lane measured 0.02 ms slower than this predicated add)
                    const uint32_t pos = atomicAdd(mv ? &wmiss : &scratch[64u + (t & 63u)], 1u);
                    if (mv) mrun[(uint64_t)run0 + pos] = cur[q] & PART_TAG_MASK;
                } else {
                    const uint32_t pos = atomicAdd(mv ? &miss_cnt[par][u * BPS + h] : &scratch[64u + (t & 63u)], 1u);
                    if (mv) pool[(uint64_t)(ce[u] >> 11) * PART_STRIDE + ((part_front(ce[u] >> 11) + pos) & (PART_BLOCK - 1u))] = cur[q] & PART_TAG_MASK;
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) cur[q] = nxt[q];
#pragma unroll
            for (uint32_t u = 0; u < K2_U; u++) ce[u] = K2_ENTRY(li + K2_U + u);
        }
#undef K2_ENTRY
        __syncthreads();
        if (nl && !DENSE) {
            const uint32_t last = ((nsteps - 1) / K2_U) * K2_U, par = (last / K2_U) & 1u;
            if (t < K2_U * BPS) {
                if (last * BPS + t < nl) {
                    const uint32_t e = list[last * BPS + t];
                    if (e != 0xFFFFFFFFu) desc[e >> 11] = ((p + 1) << 16) | miss_cnt[par][t];
                }
                miss_cnt[par][t] = 0;
            }
        }
        __syncthreads();
        if (SGC_STAMPS && (dbg & (512u | 1048576u))) { ts_loop += __builtin_amdgcn_s_memtime() - ts0; ts0 = __builtin_amdgcn_s_memtime(); }
    }
    if ((SGC_STAMPS && (dbg & 512)) && t == 0 && (blockIdx.x % 97) == 0)
        printf("K2 wg %u slice %u: scan %llu loop %llu cycles, %u groups\n", blockIdx.x, p, ts_scan, ts_loop, n_groups_dbg);
    // flush the slot counters: one atomic per occupied slot
    for (uint32_t i = t; i < slice; i += K2_THREADS) {
        const uint32_t c = SGC_DBG(dbg, 524288u) ? 0u : cnt[i];       // dbg 524288: timing-only, no flush
        if (c) {
            atomicAdd(&counts[(uint32_t)(gslots[(uint64_t)p * slice + i] & ((1ull << gid_bits) - 1ull))], c);
            local += c;
        }
    }
    if (DIRECT && t < (1u << sub_bits) && ep.recs) {
        // the direct runs: row (p << sub_bits | sub) of the run matrices belongs to this slice alone, so the workgroups that work
        // on it take columns 0 .. of it (the generic shares follow from column G): no empty columns for the consumer to walk over
        const uint32_t q = (p << sub_bits) | t, c = wmiss4[t];
        ep.cnt[(size_t)q * ep.W + g] = c;
        ep.off[(size_t)q * ep.W + g] = (uint32_t)(mrun - ep.recs) + run0 + t * stretch;
        if (c) atomicAdd(&ep.tot[q], c);
    }
    if (SGC_STAMPS && (dbg & 1048576u)) { __syncthreads(); ts_flush += __builtin_amdgcn_s_memtime() - ts0; ts0 = __builtin_amdgcn_s_memtime(); }
    }       // segments
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    __shared__ unsigned long long wsum;
    if (t == 0) wsum = 0;
    __syncthreads();
    if ((t & 63) == 0 && local) atomicAdd(&wsum, (unsigned long long)local);
    __syncthreads();
    if (t == 0 && wsum) atomicAdd(matched, wsum);
    SGC_TIMELINE_END4(dbg, tl_k2loop, s_hi - s_lo, ts_scan, ts_loop, ts_flush, 0);
    if (!ep.recs || SGC_DBG(dbg, 262144u)) return;
    // Epilogue (sgc_runs.h): what this workgroup could not settle — the misses it compacted to the fronts of its blocks —
    // and its share of the generic partition's blocks (records with an 'N' or a dead window: nothing to probe here) go to
    // core pass A, laid out by that pass's partitions in a region of ep.recs of the workgroup's own.  Two sweeps over the
    // same records: histogram, then placement.
    const uint32_t Pg = 1u << (lib.log2_slots - lib.log2_slice);                       // index of the generic partition
    uint32_t Bg;
    preg[t] = wg_scan_1024(t < k1_wgs ? wcnt[t * PART_ARR + Pg] : 0u, wtmp, &Bg);
    const uint32_t g_lo = (uint32_t)((uint64_t)Bg * blockIdx.x / gridDim.x), g_hi = (uint32_t)((uint64_t)Bg * (blockIdx.x + 1) / gridDim.x);
    __syncthreads();
    // Every dependent round trip to memory costs ~2.5 us here (the fronts left the L2 long ago), so the sweeps keep many
    // loads in flight: a wave takes eight block fronts at a time (a front is ~140 records: three 64-lane steps), the
    // generic blocks (full) are read one record per lane, four blocks at a time; the block lists are built once when
    // the workgroup's share fits one window (it does, short of extreme skew).
    __shared__ uint32_t glist[K2_GLIST];
    const bool one_window = (DENSE || s_hi - s_lo <= K2_LIST) && g_hi - g_lo <= K2_GLIST;
    const uint32_t lane = t & 63u, wave = t >> 6;
    for (uint32_t sweep = 0; sweep < 2; sweep++) {
        for (uint32_t ws = s_lo, wg = g_lo; (DENSE ? ws == s_lo : ws < s_hi) || wg < g_hi; ws += K2_LIST, wg += K2_GLIST) {
            const uint32_t nl = (!DENSE && ws < s_hi) ? (s_hi - ws < K2_LIST ? s_hi - ws : K2_LIST) : 0u;
            const uint32_t ng = wg < g_hi ? (g_hi - wg < K2_GLIST ? g_hi - wg : K2_GLIST) : 0u;
            if (!(one_window && sweep == 1)) {
                __syncthreads();
                for (uint32_t i = t; i < nl + ng; i += K2_THREADS) {
                    const bool gen = i >= nl;
                    const uint32_t *prefix = gen ? preg : pre;
                    const uint32_t o = gen ? wg + (i - nl) : ws + i, w = find_extent<10>(prefix, k1_wgs, o);
                    const uint32_t we = wlist[((size_t)w * PART_ARR + (gen ? Pg : p)) * blocks_per_wg + (o - prefix[w])], b = we >> 11;
                    // a generic block holds what k_partition put there; a slice block (not DENSE) the misses this kernel compacted to its front
                    const uint32_t e = gen ? we : (b << 11) | (desc[b] & DESC_FILL_MASK);
                    if (gen) glist[i - nl] = e; else list[i] = e;
                }
                __syncthreads();
            }
            if (SGC_DBG(dbg, 65536u << sweep)) continue;
            // DENSE: the misses are one contiguous run (walked once, with the first window), eight loads in flight per lane
            if (DENSE && !DIRECT && ws == s_lo && !(sweep == 0 && count_sub)) {
                const uint32_t M = wmiss;
                for (uint32_t j0 = 0; j0 < M; j0 += 8u * K2_THREADS) {
                    uint64_t r[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) { const uint32_t j = j0 + k * K2_THREADS + t; if (j < M) r[k] = mrun[(uint64_t)run0 + j]; }
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t j = j0 + k * K2_THREADS + t;
                        if (j >= M) continue;
                        if (sweep == 0) { const uint32_t q = run_part(ep, r[k]); if (q != RUN_DROP) atomicAdd(&hn[q], 1u); }
                        else run_place(ep, rcur, r[k]);
                    }
                }
            }
            // slice blocks: the compacted misses at the (staggered) fronts (already counted by the probe loop if K1 tagged them)
            for (uint32_t i0 = wave; i0 < nl && !DENSE && !(sweep == 0 && count_sub); i0 += 8u * (K2_THREADS / 64u)) {
                uint32_t e[8], mx = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
                    const uint32_t i = i0 + k * (K2_THREADS / 64u);
                    e[k] = i < nl ? list[i] : 0u;
                    const uint32_t f = e[k] & 2047u;
                    mx = f > mx ? f : mx;
                }
                for (uint32_t j = lane; j < mx; j += 64) {
                    uint64_t r[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (j < (e[k] & 2047u)) r[k] = pool[(uint64_t)(e[k] >> 11) * PART_STRIDE + ((part_front(e[k] >> 11) + j) & (PART_BLOCK - 1u))];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        if (j >= (e[k] & 2047u)) continue;
                        if (sweep == 0) { const uint32_t q = run_part(ep, r[k]); if (q != RUN_DROP) atomicAdd(&hn[q], 1u); }
                        else run_place(ep, rcur, r[k]);
                    }
                }
            }
            // generic blocks: filled from their first slot, usually full
            for (uint32_t i0 = 0; i0 < ng; i0 += 4) {
                uint64_t r[4];
                uint32_t f[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t e = i0 + k < ng ? glist[i0 + k] : 0u;
                    f[k] = e & 2047u;
                    if (t < f[k]) r[k] = pool[(uint64_t)(e >> 11) * PART_STRIDE + t];
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (t >= f[k]) continue;
                    if (sweep == 0) { const uint32_t q = run_part(ep, r[k]); if (q != RUN_DROP) atomicAdd(&hn[q], 1u); }
                    else run_place(ep, rcur, r[k]);
                }
            }
        }
        __syncthreads();
        // this workgroup's column in row t (thread t): behind the direct runs of the row's slice
        if (sweep == 0)
            run_reserve(ep, !DIRECT ? blockIdx.x : !balanced ? G + blockIdx.x :
                            (t < (n_slices << sub_bits) ? k2_slice_wgs(spre, t >> sub_bits, NW, T_all) : 0u) + blockIdx.x,
                        hn, rcur, wtmp, &rbase);
    }
    SGC_TIMELINE_END4(dbg, tl_k2, s_hi - s_lo, ts_scan, ts_loop, ts_flush, __builtin_amdgcn_s_memtime() - ts0);
}

// ------------------------------------------------------------------------------------------------ K3
// The rest of Counter::assign for the misses (their Centered-exact probe failed, or their status is
// non-zero).  Every gather of this kernel that leaves the XCD's L2 costs ~5x an L2 hit, and most probes of
// this stage MISS (a junk read probes the library twice and the permute table three times for nothing), so
// probes are screened by Bloom filters first: the library's (64 KiB) is staged in LDS, the permute table's
// (~4 MiB) is read through L2.  A clear bit proves absence, a set bit is followed by the real probe, so the
// outcome is exactly the reference's.
// A workgroup owns a SEGMENT of K3_SEG consecutive blocks: one wave per block copies the misses from the
// block fronts into LDS (the sparse fronts become one dense array); every lane then resolves K3_R at a time:
//   round A   filter(permute, C) + library(P) + library(M) buckets (the latter two if the LDS filter passes)
//   round A'  permute(C) bucket if its filter passed
//   round B   filter(permute, P), filter(permute, M) — only for records still unresolved after library(P)
//   round B'  permute(P), permute(M) buckets if their filters passed
// and the results are taken in the reference's order C-1mm, P-exact, P-1mm, M-exact, M-1mm
// (src/counter.rs:113-135).  Guide ids go to gids[segment * K3_SEG * BLOCK + i]; seg_cnt[segment] = count.
#define K3_SEG 64u
#define K3_THREADS 1024u
#define K3_STAGE 8192u
#define K3_R 2u
__device__ __forceinline__ bool bloom_hit(uint64_t word, uint64_t h2) {
    const uint64_t m = sgc_bloom_mask(h2);
    return (word & m) == m;
}
template <bool ONE_MM, bool PBLOOM>
__global__ void __launch_bounds__(K3_THREADS) k_resolve_miss(const uint64_t *__restrict__ pool,
                                                             const uint32_t *__restrict__ desc, uint32_t n_blocks,
                                                             uint32_t p_generic, uint32_t L, sgc_table_view lib,
                                                             sgc_table_view perm, sgc_bloom_view bl, sgc_bloom_view bp,
                                                             uint32_t *__restrict__ gids, uint32_t *__restrict__ seg_cnt,
                                                             uint32_t dbg) {
    __shared__ uint64_t st[K3_STAGE];
    __shared__ uint64_t lbf[1u << SGC_LIB_BLOOM_LOG2_WORDS];
    __shared__ uint32_t m_[K3_SEG], off_[K3_SEG + 1], n_fast, n_slow;
    const uint32_t t = threadIdx.x, seg = blockIdx.x, b0 = seg * K3_SEG;
    const uint32_t sh = 2 * (L + 2);
    const uint64_t smask = (1ull << sh) - 1ull, kmask = sgc_key_mask(L);
    if (t < K3_SEG) {          // the generic partition's blocks belong to k_generic
        const uint32_t d = b0 + t < n_blocks ? desc[b0 + t] : 0;
        m_[t] = (d >> 16) == p_generic + 1 ? 0 : d & DESC_FILL_MASK;
    }
    if (!SGC_DBG(dbg, 16u)) for (uint32_t i = t; i < (1u << SGC_LIB_BLOOM_LOG2_WORDS); i += K3_THREADS) lbf[i] = bl.words[i];
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t u = 0; u < K3_SEG; u++) { off_[u] = run; run += m_[u]; }
        off_[K3_SEG] = run;
        seg_cnt[seg] = run;
    }
    __syncthreads();
    const uint32_t T = off_[K3_SEG];
    uint32_t *out = gids + (uint64_t)seg * (K3_SEG * PART_BLOCK);
    for (uint32_t r0 = 0; r0 < T; r0 += K3_STAGE) {
        // stage [r0, r0 + K3_STAGE): wave u copies the front of block u.  Records that need the generic chain
        // (status != 0: an 'N', a dead window — ~2 % of reads) are packed at the END of the stage and resolved
        // in their own dense sweep: inline they would make almost every wave walk the serial generic chain
        // for one or two lanes.
        if (t == 0) { n_fast = 0; n_slow = 0; }
        __syncthreads();
        for (uint32_t u = t >> 6; u < K3_SEG; u += K3_THREADS / 64) {
            const uint32_t lane = t & 63u;
            const uint64_t *blkp = pool + (uint64_t)(b0 + u) * PART_STRIDE;
            const uint32_t fr = part_front(b0 + u);
            for (uint32_t j = lane; j < m_[u]; j += 64) {
                const uint32_t d = off_[u] + j;
                if (d >= r0 && d < r0 + K3_STAGE) {
                    const uint64_t rec = __builtin_nontemporal_load(&blkp[(fr + j) & (PART_BLOCK - 1u)]);
                    if ((rec >> sh) == 0 && !SGC_DBG(dbg, 128u)) st[atomicAdd(&n_fast, 1u)] = rec;
                    else st[K3_STAGE - 1u - atomicAdd(&n_slow, 1u)] = rec;
                }
            }
        }
        __syncthreads();
        const uint32_t cntr = n_fast, cnts = n_slow;
        for (uint32_t k = t; k < cnts; k += K3_THREADS) {
            const uint64_t rec = st[K3_STAGE - 1u - k];
            out[r0 + cntr + k] = SGC_DBG(dbg, 128u) ? SGC_NONE : sgc_assign<true>(rec & smask, rec >> sh, L, lib, perm, ONE_MM);
        }
        for (uint32_t d0 = 0; d0 < cntr; d0 += K3_THREADS * K3_R) {
            uint64_t rec[K3_R], fC[K3_R];
            ulonglong2 a1[K3_R], a3[K3_R];
            uint32_t x[K3_R], hP[K3_R], hM[K3_R];
            bool live[K3_R], needB[K3_R], mP[K3_R], mM[K3_R];
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                const uint32_t d = d0 + r * K3_THREADS + t;
                live[r] = d < cntr;
                rec[r] = live[r] ? st[d] : 0;
                x[r] = SGC_NONE;
                mP[r] = mM[r] = needB[r] = false;
            }
            // round A
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                if (!live[r]) continue;
                const uint64_t keyP = (rec[r] >> 4) & kmask, keyM = rec[r] & kmask;
                const uint64_t h2P = sgc_hash2(keyP), h2M = sgc_hash2(keyM);
                mP[r] = bloom_hit(lbf[sgc_bloom_word(h2P, SGC_LIB_BLOOM_LOG2_WORDS)], h2P);
                mM[r] = bloom_hit(lbf[sgc_bloom_word(h2M, SGC_LIB_BLOOM_LOG2_WORDS)], h2M);
                if (mP[r]) { hP[r] = bucket_of(lib, keyP); a1[r] = load_bucket(lib, hP[r]); }
                if (mM[r]) { hM[r] = bucket_of(lib, keyM); a3[r] = load_bucket(lib, hM[r]); }
                if (ONE_MM && PBLOOM) fC[r] = bp.words[sgc_bloom_word(sgc_hash2((rec[r] >> 2) & kmask), bp.log2_words)];
            }
            // round A'
            ulonglong2 a0[K3_R];
            uint32_t qC[K3_R];
            bool pC[K3_R];
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                pC[r] = false;
                if (!ONE_MM || !live[r]) continue;
                const uint64_t keyC = (rec[r] >> 2) & kmask;
                pC[r] = (!PBLOOM || bloom_hit(fC[r], sgc_hash2(keyC))) && !SGC_DBG(dbg, 64u);
                if (pC[r]) { qC[r] = bucket_of(perm, keyC); a0[r] = load_bucket(perm, qC[r]); }
            }
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                if (!live[r]) continue;
                if (pC[r]) x[r] = finish_find(perm, (rec[r] >> 2) & kmask, qC[r], a0[r]);
                if (x[r] == SGC_NONE && mP[r]) x[r] = finish_find(lib, (rec[r] >> 4) & kmask, hP[r], a1[r]);
                needB[r] = x[r] == SGC_NONE && !SGC_DBG(dbg, 32u);
            }
            // round B
            uint64_t fP[K3_R], fM[K3_R];
            if (ONE_MM && PBLOOM) {
#pragma unroll
                for (uint32_t r = 0; r < K3_R; r++) {
                    if (!needB[r]) continue;
                    fP[r] = bp.words[sgc_bloom_word(sgc_hash2((rec[r] >> 4) & kmask), bp.log2_words)];
                    fM[r] = bp.words[sgc_bloom_word(sgc_hash2(rec[r] & kmask), bp.log2_words)];
                }
            }
            // round B'
            ulonglong2 b2[K3_R], b4[K3_R];
            uint32_t qP[K3_R], qM[K3_R];
            bool pP[K3_R], pM[K3_R];
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                pP[r] = pM[r] = false;
                if (!ONE_MM || !needB[r]) continue;
                const uint64_t keyP = (rec[r] >> 4) & kmask, keyM = rec[r] & kmask;
                pP[r] = !PBLOOM || bloom_hit(fP[r], sgc_hash2(keyP));
                pM[r] = !PBLOOM || bloom_hit(fM[r], sgc_hash2(keyM));
                if (pP[r]) { qP[r] = bucket_of(perm, keyP); b2[r] = load_bucket(perm, qP[r]); }
                if (pM[r]) { qM[r] = bucket_of(perm, keyM); b4[r] = load_bucket(perm, qM[r]); }
            }
#pragma unroll
            for (uint32_t r = 0; r < K3_R; r++) {
                if (needB[r]) {
                    if (pP[r]) x[r] = finish_find(perm, (rec[r] >> 4) & kmask, qP[r], b2[r]);
                    if (x[r] == SGC_NONE && mM[r]) x[r] = finish_find(lib, rec[r] & kmask, hM[r], a3[r]);
                    if (x[r] == SGC_NONE && pM[r]) x[r] = finish_find(perm, rec[r] & kmask, qM[r], b4[r]);
                }
                if (live[r]) out[r0 + d0 + r * K3_THREADS + t] = x[r];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ Kg
// The generic partition: the full Counter::assign chain (sgc_assign), one record per lane, full waves.
// Workgroup w looks at the descriptors of K1 workgroup w's block range only.  ~2 % of the reads end up
// here, about half of them match: their counts go straight to the count vector with device-scope atomics.
template <bool ONE_MM>
__global__ void __launch_bounds__(256) k_generic(const uint64_t *__restrict__ pool, const uint32_t *__restrict__ desc,
                                                 uint32_t blocks_per_wg, uint32_t p_generic, uint32_t L,
                                                 sgc_table_view lib, sgc_table_view perm, uint32_t *__restrict__ counts,
                                                 unsigned long long *__restrict__ matched) {
    const uint32_t sh = 2 * (L + 2), t = threadIdx.x;
    const uint64_t smask = (1ull << sh) - 1ull;
    uint64_t local = 0;
    for (uint32_t k = 0; k < blocks_per_wg; k++) {
        const uint32_t b = blockIdx.x * blocks_per_wg + k;
        const uint32_t d = desc[b];
        if ((d >> 16) != p_generic + 1) continue;
        const uint32_t m = d & DESC_FILL_MASK;
        const uint64_t *blkp = pool + (uint64_t)b * PART_STRIDE;
        for (uint32_t j = t; j < m; j += 256) {
            const uint64_t rec = blkp[j];
            const uint32_t x = sgc_assign<true>(rec & smask, rec >> sh, L, lib, perm, ONE_MM);
            if (x != SGC_NONE) { atomicAdd(&counts[x], 1u); local++; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((t & 63) == 0 && local) atomicAdd(matched, (unsigned long long)local);
}

// ------------------------------------------------------------------------------------------------ K4
// LDS histogram of the segments' guide ids, one guide-range slice per pass; also tallies the matched reads.
// Four 256-lane groups walk different segments at once (a segment is short: a few hundred to ~1k ids).
#define K4_SLICE 36864u
__global__ void __launch_bounds__(1024) k_hist_segments(const uint32_t *__restrict__ gids, const uint32_t *__restrict__ seg_cnt,
                                                        uint32_t n_segs, uint32_t n_guides, uint32_t *__restrict__ counts,
                                                        unsigned long long *__restrict__ matched) {
    __shared__ uint32_t h[K4_SLICE];
    const uint32_t t = threadIdx.x;
    uint64_t local = 0;
    for (uint32_t base = 0; base < n_guides; base += K4_SLICE) {
        for (uint32_t j = t; j < K4_SLICE; j += 1024) h[j] = 0;
        __syncthreads();
        for (uint32_t sg = blockIdx.x * 4 + (t >> 8); sg < n_segs; sg += gridDim.x * 4) {
            const uint32_t c = seg_cnt[sg];
            const uint32_t *src = gids + (uint64_t)sg * (K3_SEG * PART_BLOCK);
            for (uint32_t i = t & 255u; i < c; i += 256) {
                const uint32_t r = src[i] - base;
                if (r < K4_SLICE) { atomicAdd(&h[r], 1u); local++; }
            }
        }
        __syncthreads();
        const uint32_t lim = n_guides - base < K4_SLICE ? n_guides - base : K4_SLICE;
        for (uint32_t j = t; j < lim; j += 1024) {
            const uint32_t v = h[j];
            if (v) atomicAdd(&counts[base + j], v);
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    __shared__ unsigned long long wsum;
    if (t == 0) wsum = 0;
    __syncthreads();
    if ((t & 63) == 0 && local) atomicAdd(&wsum, (unsigned long long)local);
    __syncthreads();
    if (t == 0 && wsum) atomicAdd(matched, wsum);
}

// ------------------------------------------------------------------------------------------------ host side
bool sgc_part_supported(const sgc_table_view &lib, bool rec16) {
    if (rec16 || lib.gid_bits == 0) return false;
    if (lib.log2_slice > SGC_LDS_LOG2_SLICE_BIG || lib.log2_slice < 1) return false;
    return (lib.log2_slots - lib.log2_slice) <= SGC_PART_MAX_LOG2_SLICES;      // <= PART_MAXP partitions
}

void sgc_part_plan(uint64_t n, const sgc_table_view &lib, uint32_t max_wgs, sgc_part_geometry *g) {
    const uint32_t P = 1u << (lib.log2_slots - lib.log2_slice);
    uint64_t tiles = (n + PART_TILE - 1) / PART_TILE;
    if (max_wgs == 0) max_wgs = 256;
    if (max_wgs > K2_THREADS) max_wgs = K2_THREADS;          // k_count_slices scans one count per K1 workgroup in one go
    uint32_t wgs = (uint32_t)(tiles < max_wgs ? (tiles ? tiles : 1) : max_wgs);
    uint64_t per = (tiles + wgs - 1) / wgs * PART_TILE;                 // records per K1 workgroup, whole tiles
    g->k1_wgs = wgs;
    g->per_wg = per;
    g->blocks_per_wg = (uint32_t)(per / PART_BLOCK) + P + 1;           // full blocks + one open block per partition (incl. generic)
    g->n_blocks = wgs * g->blocks_per_wg;
    g->pool_bytes = (uint64_t)g->n_blocks * PART_STRIDE * 8;
    g->desc_tail_off = ((uint64_t)g->n_blocks * 4 + 255) & ~255ull;
    g->wcnt_off = g->desc_tail_off + SGC_DESC_TAIL;          // k_partition zeroes the tail and writes every wcnt entry
    g->wlist_off = (g->wcnt_off + (uint64_t)wgs * PART_ARR * 4 + 255) & ~255ull;
    g->desc_bytes = g->wlist_off + (uint64_t)wgs * PART_ARR * g->blocks_per_wg * 4;
    g->n_segs = (g->n_blocks + K3_SEG - 1) / K3_SEG;
    g->gids_bytes = (uint64_t)g->n_segs * K3_SEG * PART_BLOCK * 4;
    g->partitions = P;
    g->block_records = PART_BLOCK;
}

void sgc_launch_part_k1(hipStream_t st, unsigned long long *err, const uint64_t *recs, uint64_t n, uint32_t L, const sgc_table_view &lib, uint32_t sub_bits,
                        const sgc_part_geometry &g, uint64_t *pool, uint32_t *desc, int slice_rec, uint32_t dbg, uint32_t *slice_tot) {
    const bool core_hashed = lib.core_cl != 0 && lib.log2_slice < lib.log2_slots;
#define K1_LAUNCH(MODE)                                                                                                            \
    hipLaunchKernelGGL((k_partition<MODE>), dim3(g.k1_wgs), dim3(K1_THREADS), sgc_extra_lds("K1"), st, recs, n, g.per_wg, g.blocks_per_wg, L,          \
                       lib.log2_slots, lib.log2_slice, lib.core_cl, sub_bits, pool, desc, (uint32_t *)((char *)desc + g.desc_tail_off), \
                       SGC_DESC_TAIL / 4, (uint32_t *)((char *)desc + g.wcnt_off), (uint32_t *)((char *)desc + g.wlist_off), err, dbg, slice_tot)
    if (slice_rec == 2 && core_hashed) K1_LAUNCH(3);
    else if (slice_rec == 1 && core_hashed) K1_LAUNCH(2);
    else if (core_hashed) K1_LAUNCH(1);
    else K1_LAUNCH(0);
#undef K1_LAUNCH
}

// workgroups of k_count_slices: G per slice, two per CU in all — one resident generation
static uint32_t k2_shares(const sgc_part_geometry &g) { return g.partitions >= 512 ? 1 : 512 / g.partitions; }
uint32_t sgc_part_k2_grid(const sgc_part_geometry &g) { return g.partitions * k2_shares(g); }
uint32_t sgc_part_k2_shares(const sgc_part_geometry &g) { return k2_shares(g); }
// columns of the run matrices for the direct runs
uint32_t sgc_part_k2_direct_cols(const sgc_part_geometry &g, bool balanced) {
    if (!balanced) return k2_shares(g);
    return g.partitions * k2_shares(g);          // any workgroup may meet any slice
}

void sgc_launch_part_k2(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_part_geometry &g,
                        uint64_t *pool, uint32_t *desc, uint32_t *counts, unsigned long long *matched, uint32_t dbg,
                        const sgc_runs *runs, const uint64_t *cuckoo, uint64_t *mrun, uint32_t *mcur, bool direct_runs, int slice_rec,
                        const uint32_t *slice_tot, uint32_t *slice_tot_next, bool wide) {
    const uint32_t grid = g.partitions * k2_shares(g);
    // balanced shares (slice_tot: k_partition's blocks per slice; direct runs only): any workgroup may work on any slice, so the run
    // matrices keep `grid` columns for the direct runs; static shares: the k2_shares(g) workgroups of a slice
    if (!(runs && mrun && direct_runs && runs->sub_bits != 0xFFu)) slice_tot = nullptr;
    const uint32_t G = slice_tot ? sgc_part_k2_direct_cols(g, true) : k2_shares(g);
    sgc_runs none{};
    const uint32_t *wcnt = (const uint32_t *)((const char *)desc + g.wcnt_off), *wlist = (const uint32_t *)((const char *)desc + g.wlist_off);
    const bool dense = runs && mrun;
    const bool direct = dense && direct_runs && runs->sub_bits != 0xFFu;
    // (slices of 2^13 slots — libraries beyond 128 slices of 2^12 — take the same kernel with twice the LDS: one workgroup per CU)
#define K2_LAUNCH(CK, DN, DR, REC)                                                                                                     \
    do { if (lib.log2_slice > SGC_LDS_LOG2_SLICE)                                                                                      \
        hipLaunchKernelGGL((k_count_slices<SGC_LDS_LOG2_SLICE_BIG, CK, DN, DR, REC>), dim3(grid), dim3(K2_THREADS), sgc_extra_lds("K2"), st, pool, desc, wcnt, wlist, \
                           g.k1_wgs, g.blocks_per_wg, G, L, lib, counts, matched, dbg, runs ? *runs : none, cuckoo, mrun, mcur, slice_tot, slice_tot_next); \
    else                                                                                                                               \
        hipLaunchKernelGGL((k_count_slices<SGC_LDS_LOG2_SLICE, CK, DN, DR, REC>), dim3(grid), dim3(K2_THREADS), sgc_extra_lds("K2"), st, pool, desc, wcnt, wlist, \
                           g.k1_wgs, g.blocks_per_wg, G, L, lib, counts, matched, dbg, runs ? *runs : none, cuckoo, mrun, mcur, slice_tot, slice_tot_next); } while (0)
    const int rec = direct ? slice_rec : 0;          // 0 = 8-byte, 1 = six-byte, 2 = five-byte slice blocks
    const bool l20 = L == 20 && rec == 2 && cuckoo && lib.core_cl == 9 && lib.log2_slice == SGC_LDS_LOG2_SLICE &&
                     lib.log2_slots == SGC_LDS_LOG2_SLICE + 6u && runs->sub_bits == 2;
    if (l20 && wide)
        hipLaunchKernelGGL((k_count_slices<SGC_LDS_LOG2_SLICE, true, true, true, 2, 20, true>), dim3(grid), dim3(K2_THREADS), sgc_extra_lds("K2"), st, pool, desc, wcnt, wlist,
                           g.k1_wgs, g.blocks_per_wg, G, L, lib, counts, matched, dbg, *runs, cuckoo, mrun, mcur, slice_tot, slice_tot_next);
    else if (l20)
        hipLaunchKernelGGL((k_count_slices<SGC_LDS_LOG2_SLICE, true, true, true, 2, 20>), dim3(grid), dim3(K2_THREADS), sgc_extra_lds("K2"), st, pool, desc, wcnt, wlist,
                           g.k1_wgs, g.blocks_per_wg, G, L, lib, counts, matched, dbg, *runs, cuckoo, mrun, mcur, slice_tot, slice_tot_next);
    else if (cuckoo) { if (rec == 2) K2_LAUNCH(true, true, true, 2); else if (rec == 1) K2_LAUNCH(true, true, true, 1); else if (direct) K2_LAUNCH(true, true, true, 0); else if (dense) K2_LAUNCH(true, true, false, 0); else K2_LAUNCH(true, false, false, 0); }
    else { if (rec == 2) K2_LAUNCH(false, true, true, 2); else if (rec == 1) K2_LAUNCH(false, true, true, 1); else if (direct) K2_LAUNCH(false, true, true, 0); else if (dense) K2_LAUNCH(false, true, false, 0); else K2_LAUNCH(false, false, false, 0); }
#undef K2_LAUNCH
}

void sgc_launch_part_k3(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm,
                        const sgc_bloom_view &bloom_lib, const sgc_bloom_view &bloom_perm, const sgc_part_geometry &g,
                        const uint64_t *pool, const uint32_t *desc, uint32_t *seg_cnt, uint32_t *gids, uint32_t dbg) {
    const unsigned grid = (g.n_blocks + K3_SEG - 1) / K3_SEG;
    if (one_mm && SGC_DBG(dbg, 256u))
        hipLaunchKernelGGL((k_resolve_miss<true, false>), dim3(grid), dim3(K3_THREADS), 0, st, pool, desc, g.n_blocks, g.partitions,
                           L, lib, perm, bloom_lib, bloom_perm, gids, seg_cnt, dbg);
    else if (one_mm)
        hipLaunchKernelGGL((k_resolve_miss<true, true>), dim3(grid), dim3(K3_THREADS), 0, st, pool, desc, g.n_blocks, g.partitions,
                           L, lib, perm, bloom_lib, bloom_perm, gids, seg_cnt, dbg);
    else
        hipLaunchKernelGGL((k_resolve_miss<false, false>), dim3(grid), dim3(K3_THREADS), 0, st, pool, desc, g.n_blocks, g.partitions,
                           L, lib, perm, bloom_lib, bloom_perm, gids, seg_cnt, dbg);
}

void sgc_launch_part_generic(hipStream_t st, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm,
                             const sgc_part_geometry &g, const uint64_t *pool, const uint32_t *desc, uint32_t *counts,
                             unsigned long long *matched) {
    if (one_mm)
        hipLaunchKernelGGL((k_generic<true>), dim3(g.k1_wgs), dim3(256), 0, st, pool, desc, g.blocks_per_wg, g.partitions, L,
                           lib, perm, counts, matched);
    else
        hipLaunchKernelGGL((k_generic<false>), dim3(g.k1_wgs), dim3(256), 0, st, pool, desc, g.blocks_per_wg, g.partitions, L,
                           lib, perm, counts, matched);
}

void sgc_launch_part_k4(hipStream_t st, uint32_t n_guides, const sgc_part_geometry &g, const uint32_t *gids,
                        const uint32_t *seg_cnt, uint32_t *counts, unsigned long long *matched) {
    const unsigned n_segs = (g.n_blocks + K3_SEG - 1) / K3_SEG;
    const unsigned grid = n_segs < 4 * 256 ? (n_segs + 3) / 4 : 256;
    hipLaunchKernelGGL(k_hist_segments, dim3(grid), dim3(1024), 0, st, gids, seg_cnt, n_segs, n_guides, counts, matched);
}
