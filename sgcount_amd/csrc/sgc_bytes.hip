// sgc_bytes.hip — the generic byte-string count path (sgc_bytes.h): libraries with bytes outside ACGT or longer than 30.
//
//   k_fastq_lines   FASTQ text -> (start, end) of every sequence line of the part, marker bytes verified (the same rules as
//                   k_fastq_pack in sgc_fastq.hip: '\r' before '\n' belongs to the terminator, '@' at lines 4k, '+' at 4k + 2)
//   k_bytes_count   one lane per read: Counter::assign (reference src/counter.rs:96-140) on the read's bytes — bounds
//                   (:158-180), trim (:184-204), Library::contains (src/library.rs:34-40) and Permuter::contains
//                   (src/permutes.rs:55-57) through the hashed tables, every hit verified byte for byte
//   k_bytes_lookup  point lookups for sgc_lookup
// A fallback: global probes, global atomics, no partitioning.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_bytes.h"
#include "sgc_device.h"
#include "sgc_kernels.h"

#define BL_THREADS 1024u
#define BL_TILE 65536u          // the tile of k_fastq_count (sgc_fastq.hip), whose per-tile newline prefix this kernel consumes

__host__ __device__ __forceinline__ uint32_t bl_seq_lines_before(uint32_t ph, uint32_t i) { return ((ph + i + 2u) >> 2) - ((ph + 2u) >> 2); }

__global__ void __launch_bounds__(BL_THREADS) k_fastq_lines(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ tile_base,
                                                            uint32_t tiles, uint64_t first_line, uint32_t expect_nl, uint32_t n_lines,
                                                            uint64_t *__restrict__ starts, uint64_t *__restrict__ ends,
                                                            unsigned long long *__restrict__ err) {
    __shared__ uint32_t wtmp[17];
    const uint32_t t = threadIdx.x, tile = blockIdx.x;
    const uint64_t b0 = (uint64_t)tile * BL_TILE + (uint64_t)t * 64u;
    const uint32_t ph = (uint32_t)(first_line & 3u);
    uint64_t mine = 0;
    for (uint32_t k = 0; k < 64; k++)
        if (b0 + k < n && text[b0 + k] == 0x0A) mine |= 1ull << k;
    uint32_t total;
    const uint32_t before = wg_scan_1024((uint32_t)__popcll(mine), wtmp, &total);
    uint32_t i = tile_base[tile] + before;                  // local number of the line my next newline ends
    if (tile == 0 && t == 0) {
        const uint32_t nl = tile_base[tiles];
        if (nl != expect_nl) atomicOr(&err[1], 1ull);
        const uint8_t c = text[0];
        if ((ph == 0 && c != '@') || (ph == 2 && c != '+')) atomicMax(&err[0], ~0ull - ((unsigned long long)first_line + 1ull));
        if (ph == 1u && n_lines) starts[0] = 0;             // the part starts with a sequence line
        if (n_lines > nl && ((ph + nl) & 3u) == 1u) {      // the stream's last line has no '\n' and is a sequence line
            uint64_t e = n;
            if (e && text[e - 1] == 0x0D) e--;
            ends[bl_seq_lines_before(ph, nl)] = e;
        }
    }
    while (mine) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(mine);
        mine &= mine - 1;
        const uint64_t pos = b0 + bit;
        const uint32_t phase = (ph + i) & 3u;
        if (pos + 1u < n && (phase == 3u || phase == 1u)) {
            const uint8_t c = text[pos + 1u];
            if (c != (phase == 3u ? '@' : '+')) atomicMax(&err[0], ~0ull - ((unsigned long long)first_line + i + 2ull));
        }
        if (phase == 1u && i < n_lines) {
            uint64_t e = pos;
            if (e && text[e - 1] == 0x0D) e--;
            ends[bl_seq_lines_before(ph, i)] = e;
        }
        if (phase == 0u && i + 1u < n_lines) starts[bl_seq_lines_before(ph, i + 1u)] = pos + 1u;
        i++;
    }
}

// oriented byte p of the read at s[0, n): forward, or the reverse complement (oracle ctr_trim; fxread's seq_rev_comp is unpinned)
__device__ __forceinline__ uint8_t bytes_at(const uint8_t *s, uint32_t n, uint32_t p, int reverse) {
    if (!reverse) return s[p];
    const uint8_t c = s[n - 1u - p];
    return (c & 2) ? (uint8_t)(c ^ 4) : (uint8_t)(c ^ 21);
}

// the guide whose sequence equals the window w[p, p + L) (Library::contains), or SGC_NONE
__device__ __forceinline__ uint32_t bytes_lib_find(const sgc_bytes_view &v, uint64_t h, const uint8_t *s, uint32_t n, uint32_t p, int reverse) {
    const uint32_t mask = (1u << v.lib_log2) - 1u;
    for (uint32_t slot = sgc_bytes_slot(h, v.lib_log2);; slot = (slot + 1u) & mask) {
        const uint64_t tag = v.lib_tag[slot];
        if (tag == SGC_BYTES_EMPTY) return SGC_NONE;
        if (tag != h) continue;
        const uint32_t g = v.lib_val[slot];
        const uint8_t *q = v.seqs + (size_t)g * v.L;
        bool same = true;
        for (uint32_t k = 0; k < v.L && same; k++) same = q[k] == bytes_at(s, n, p + k, reverse);
        if (same) return g;
    }
}

// the unique parent of the window as a child (Permuter::contains + Library::alias, src/counter.rs:113-116), or SGC_NONE
__device__ __forceinline__ uint32_t bytes_perm_find(const sgc_bytes_view &v, uint64_t h, const uint8_t *s, uint32_t n, uint32_t p, int reverse) {
    const uint32_t mask = (1u << v.perm_log2) - 1u;
    for (uint32_t slot = sgc_bytes_slot(h, v.perm_log2);; slot = (slot + 1u) & mask) {
        const uint64_t tag = v.perm_tag[slot];
        if (tag == SGC_BYTES_EMPTY) return SGC_NONE;
        if (tag != h) continue;
        const uint32_t g = v.perm_val[slot];
        const uint32_t pl = v.perm_pl[slot];
        const uint32_t j = pl & 0xFFFFFFu;
        const uint8_t b = (uint8_t)(pl >> 24);
        const uint8_t *q = v.seqs + (size_t)g * v.L;
        bool same = true;
        for (uint32_t k = 0; k < v.L && same; k++) same = (k == j ? b : q[k]) == bytes_at(s, n, p + k, reverse);
        if (same) return g;
    }
}

__device__ __forceinline__ uint32_t bytes_window(const sgc_bytes_view &v, const uint8_t *s, uint32_t n, uint32_t p, int reverse, bool one_mm) {
    uint64_t h = sgc_bytes_hash_init();
    for (uint32_t k = 0; k < v.L; k++) h = sgc_bytes_hash_step(h, bytes_at(s, n, p + k, reverse));
    h = sgc_bytes_hash_fin(h);
    uint32_t g = bytes_lib_find(v, h, s, n, p, reverse);                       // src/counter.rs:111
    if (g == SGC_NONE && one_mm) g = bytes_perm_find(v, h, s, n, p, reverse);   // :113-116
    return g;
}

__global__ void __launch_bounds__(256) k_bytes_count(const sgc_bytes_view v, const uint8_t *__restrict__ text, const uint64_t *__restrict__ starts,
                                                     const uint64_t *__restrict__ ends, uint64_t n_reads, int reverse, uint32_t o, int recursion,
                                                     bool one_mm, uint32_t *__restrict__ counts, unsigned long long *__restrict__ matched,
                                                     const uint8_t *__restrict__ flags /* hybrid library: only the reads flagged for this chain (null = all) */) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t g = SGC_NONE;
    if (i < n_reads && (!flags || flags[i])) {
        const uint64_t b = starts[i], e = ends[i];
        const uint8_t *s = text + b;
        const uint64_t len = e > b ? e - b : 0;
        // windows that could be in bounds start below o + 2; a read longer than 2^32 is cut to what they can see (forward strand)
        const uint32_t n = (uint32_t)(len < 0xFFFFFFFFull ? len : 0xFFFFFFFFull);
        const uint64_t L = v.L;
        // Counter::bounds (src/counter.rs:158-180): a window past the read's end ends the whole chain with None
        if ((uint64_t)o + L <= n) {
            g = bytes_window(v, s, n, o, reverse, one_mm);                                           // Centered / Null
            if (g == SGC_NONE && recursion && (uint64_t)o + 1u + L <= n) {
                g = bytes_window(v, s, n, o + 1u, reverse, one_mm);                                  // Plus  (:123-125)
                if (g == SGC_NONE && o >= 1u) g = bytes_window(v, s, n, o - 1u, reverse, one_mm);    // Minus (:128-130)
            }
        }
    }
    const bool hit = g != SGC_NONE;
    if (hit) atomicAdd(&counts[g], 1u);
    const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(matched, (unsigned long long)__popcll(m));
}

// which: 0 library only, 1 children only, 2 library then children (sgc_lookup)
__global__ void __launch_bounds__(256) k_bytes_lookup(const sgc_bytes_view v, const uint8_t *__restrict__ tokens, uint64_t n, int which,
                                                      bool one_mm, int32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint8_t *s = tokens + i * v.L;
    uint64_t h = sgc_bytes_hash_init();
    for (uint32_t k = 0; k < v.L; k++) h = sgc_bytes_hash_step(h, s[k]);
    h = sgc_bytes_hash_fin(h);
    uint32_t g = SGC_NONE;
    if (which != 1) g = bytes_lib_find(v, h, s, v.L, 0, 0);
    if (g == SGC_NONE && which != 0 && one_mm) g = bytes_perm_find(v, h, s, v.L, 0, 0);
    out[i] = g == SGC_NONE ? -1 : (int32_t)g;
}

void sgc_launch_fastq_lines(hipStream_t st, const uint8_t *text, uint64_t n, const uint32_t *tile_scratch, uint64_t first_line,
                            uint32_t expect_nl, uint32_t n_lines, uint64_t *starts, uint64_t *ends, unsigned long long *err) {
    if (n == 0) return;
    const uint32_t tiles = sgc_fastq_tiles(n);
    hipLaunchKernelGGL(k_fastq_lines, dim3(tiles), dim3(BL_THREADS), 0, st, text, n, tile_scratch, tiles, first_line, expect_nl, n_lines,
                       starts, ends, err);
}

void sgc_launch_bytes_count(hipStream_t st, const sgc_bytes_view &v, const uint8_t *text, const uint64_t *starts, const uint64_t *ends,
                            uint64_t n_reads, int reverse, uint32_t o, int recursion, bool one_mm, uint32_t *counts,
                            unsigned long long *matched, const uint8_t *flags) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(k_bytes_count, dim3((unsigned)((n_reads + 255u) / 256u)), dim3(256), 0, st, v, text, starts, ends, n_reads, reverse, o,
                       recursion, one_mm, counts, matched, flags);
}

// ---- hybrid libraries: mostly ACGT guides, a few with other bytes ('N' above all) -------------------------------------------------
// The packed pass is built over the ACGT guides alone; it is exact for every read that no other guide can influence.  A guide
// g with a byte outside ACGT is within one substitution of a window W (src/counter.rs:111-117, src/permutes.rs:127-144) only if
// W holds a byte outside ACGT itself, or g has exactly ONE such byte and W equals g everywhere else — one of four "shadow" keys
// per such guide.  So a read goes to the byte-string chain over the WHOLE library (k_bytes_count, flagged) iff its span region
// [o - 1, o + L + 1) holds a byte outside ACGT or one of its windows hits the Bloom filter of the shadow keys (a false positive
// only costs the slower chain, which is always right); its packed record is replaced by the all-dead one.  Everything else
// keeps its record: no guide outside ACGT is within distance one of any of its windows, so exact matches, unique parents and
// ambiguity are decided among the ACGT guides exactly as in the whole library.
template <bool REC16>
__global__ void __launch_bounds__(256) k_bytes_route(const uint8_t *__restrict__ text, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
                                                     uint64_t n_reads, uint32_t L, int reverse, uint32_t o, int recursion, uint64_t *__restrict__ recs,
                                                     sgc_bloom_view shadow, uint8_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_reads) return;
    const uint64_t b = starts[i], e = ends[i];
    const uint8_t *s = text + b;
    const uint64_t len = e > b ? e - b : 0;
    const uint32_t n = (uint32_t)(len < 0xFFFFFFFFull ? len : 0xFFFFFFFFull), K = L + 2u;
    bool route = false;
    for (uint32_t w = 0; w < K && !route; w++) {
        const int64_t p = (int64_t)o - 1 + (int64_t)w;
        if (p < 0 || (uint64_t)p >= n) continue;
        const uint8_t c = bytes_at(s, n, (uint32_t)p, reverse);
        route = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
    }
    if (!route && shadow.words) {
        const uint64_t span = REC16 ? recs[2 * i] : (recs[i] & ((1ull << (2u * K)) - 1ull)), kmask = sgc_key_mask(L);
        const bool c_ok = (uint64_t)o + L <= n, p_ok = c_ok && recursion && (uint64_t)o + 1u + L <= n, m_ok = p_ok && o >= 1u;
        const uint64_t keys[3] = {(span >> 2) & kmask, (span >> 4) & kmask, span & kmask};
        const bool ok[3] = {c_ok, p_ok, m_ok};
        for (int k = 0; k < 3 && !route; k++) {
            if (!ok[k]) continue;
            const uint64_t h2 = sgc_hash2(keys[k]);
            const uint64_t m = sgc_bloom_mask(h2);
            route = (shadow.words[sgc_bloom_word(h2, shadow.log2_words)] & m) == m;
        }
    }
    flags[i] = route ? 1 : 0;
    if (route) {
        const uint64_t dead = (uint64_t)SGC_STATE_DEAD * (1u + K + K * K);
        if (REC16) { recs[2 * i] = 0; recs[2 * i + 1] = dead; } else recs[i] = dead << (2u * K);
    }
}

void sgc_launch_bytes_route(hipStream_t st, const uint8_t *text, const uint64_t *starts, const uint64_t *ends, uint64_t n_reads, uint32_t L, bool rec16,
                            int reverse, uint32_t o, int recursion, uint64_t *recs, const sgc_bloom_view &shadow, uint8_t *flags) {
    if (n_reads == 0) return;
    const unsigned grid = (unsigned)((n_reads + 255u) / 256u);
    if (rec16) hipLaunchKernelGGL((k_bytes_route<true>), dim3(grid), dim3(256), 0, st, text, starts, ends, n_reads, L, reverse, o, recursion, recs, shadow, flags);
    else hipLaunchKernelGGL((k_bytes_route<false>), dim3(grid), dim3(256), 0, st, text, starts, ends, n_reads, L, reverse, o, recursion, recs, shadow, flags);
}

// counts64[map[i]] += packed counts32[i]; counts32[i] = 0  (the packed pass of a hybrid library numbers only its ACGT guides)
__global__ void k_fold_map(uint32_t *__restrict__ c32p, const uint32_t *__restrict__ map, unsigned long long *__restrict__ c64, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t v = c32p[i]; if (v) { c64[map[i]] += v; c32p[i] = 0; } }
}
void sgc_launch_fold_map(hipStream_t st, uint32_t *c32p, const uint32_t *map, unsigned long long *c64, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fold_map, dim3((n + 255) / 256), dim3(256), 0, st, c32p, map, c64, n);
}

void sgc_launch_bytes_lookup(hipStream_t st, const sgc_bytes_view &v, const uint8_t *tokens, uint64_t n, int which, bool one_mm, int32_t *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_bytes_lookup, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, st, v, tokens, n, which, one_mm, out);
}
