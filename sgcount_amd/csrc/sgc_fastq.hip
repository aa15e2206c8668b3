// sgc_fastq.hip — FASTQ ingest on the GPU: record boundaries, window extraction and 2-bit packing from raw
// FASTQ text resident in HBM (replaces fxread + Counter::apply_trim, reference src/counter.rs:144-204, for the
// sgc_sample_push_fastq* entry points), and the packer for raw reads (bytes + offsets).
//
// FASTQ is four lines per record, so record boundaries are a matter of counting newlines.  A *part* of a FASTQ
// stream is a run of whole lines that starts at (global) line number first_line — any phase of the 4-line cycle:
//   pass A  k_fastq_count   per 64 KiB tile: number of '\n'                      (reads the text once)
//           k_scan_tiles    exclusive prefix over the tiles (one workgroup)
//   pass B  k_fastq_pack    per tile: the tile (+ a halo of o + L + 2 bytes) is staged in LDS with coalesced
//                           16-byte loads; a 64-bit newline mask per 64 bytes + one workgroup scan give every
//                           newline its line number; the marker bytes ('@' at lines 4k, '+' at lines 4k + 2)
//                           are verified; the lanes whose newline opens (forward strand) or closes (reverse
//                           strand) a sequence line list it; one lane per listed line then packs the record
//                           from LDS.  Only o + L + 2 bytes of a sequence line are ever needed: the forward
//                           strand reads them from the line start (owner: the tile where the line starts), the
//                           reverse strand from the line end (owner: the tile where it ends), so the halo is
//                           bounded by the offset and not by the read length.
// A '\r' before the '\n' belongs to the line terminator (CRLF files; fxread's behaviour is unpinned — DESIGN.md §2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sgc_device.h"
#include "sgc_format.h"
#include "sgc_kernels.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FQ_THREADS 1024u
#define FQ_TILE 65536u                      // bytes per tile: 64 per lane
#define FQ_HALO_MAX 1024u                   // bytes staged beyond the tile (multiple of 64; two workgroups per CU must fit
                                            // 160 KiB of LDS); an offset beyond it makes the owning lanes read global memory
#define FQ_CAP 1024u                        // sequence lines listed per round (a tile of tiny records takes several rounds)
#define FQ_PIECES ((FQ_TILE + FQ_HALO_MAX) / 16u)

// 16-bit mask of '\n' among the 16 bytes at text[base ..) (bytes past n read as 0); optionally stages them in LDS
__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *__restrict__ text, uint64_t base, uint64_t n, uint8_t *stage) {
    uint32_t m = 0;
    if (base + 16 <= n && ((uintptr_t)(text + base) & 15) == 0) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(text + base));
        if (stage) *reinterpret_cast<u32x4 *>(stage) = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // exact per-byte test: bit 7 of every byte of y is set iff that byte of x is zero (no carries between bytes)
            const uint32_t x = w[k] ^ 0x0A0A0A0Au;
            const uint32_t y = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            // gather the four bit-7 flags into 4 adjacent bits
            m |= (((y >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * k);
        }
    } else {
        for (uint32_t k = 0; k < 16; k++) {
            const uint8_t c = base + k < n ? text[base + k] : 0;
            if (stage) stage[k] = c;
            if (c == 0x0A) m |= 1u << k;
        }
    }
    return m;
}

__global__ void __launch_bounds__(FQ_THREADS) k_fastq_count(const uint8_t *__restrict__ text, uint64_t n,
                                                            uint32_t *__restrict__ tile_nl) {
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    const uint32_t t = threadIdx.x;
    const uint64_t tile0 = (uint64_t)blockIdx.x * FQ_TILE;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t s = 0; s < FQ_TILE / (FQ_THREADS * 16u); s++) {      // 16 B per lane per load: fully coalesced
        const uint64_t base = tile0 + ((uint64_t)s * FQ_THREADS + t) * 16;
        if (base < n) c += __popc(nl_mask16(text, base, n, nullptr));
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((t & 63) == 0) wsum[t >> 6] = c;
    __syncthreads();
    if (t == 0) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < FQ_THREADS / 64; w++) sum += wsum[w];
        tile_nl[blockIdx.x] = sum;
    }
}

// exclusive scan of n u32 values in place (+ total at v[n]); one workgroup
__global__ void __launch_bounds__(1024) k_scan_tiles(uint32_t *__restrict__ v, uint32_t n) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t x = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
    if (t == 1023) v[n] = part[1023];
}

// lines j in [0, i) of the part with (ph + j) % 4 == 1, i.e. sequence lines before local line i
__host__ __device__ __forceinline__ uint32_t seq_lines_before(uint32_t ph, uint32_t i) { return ((ph + i + 2u) >> 2) - ((ph + 2u) >> 2); }

// first newline in stage-relative [a, b), or b
__device__ __forceinline__ uint32_t find_nl_fwd(const uint64_t *m64, uint32_t a, uint32_t b) {
    if (a >= b) return b;
    uint32_t w = a >> 6;
    uint64_t word = m64[w] & (~0ull << (a & 63u));
    for (;;) {
        if (word) { const uint32_t q = w * 64u + (uint32_t)__builtin_ctzll(word); return q < b ? q : b; }
        w++;
        if (w * 64u >= b) return b;
        word = m64[w];
    }
}
// last newline in stage-relative [a, b), or 0xFFFFFFFF
__device__ __forceinline__ uint32_t find_nl_bwd(const uint64_t *m64, uint32_t a, uint32_t b) {
    if (a >= b) return 0xFFFFFFFFu;
    uint32_t w = (b - 1u) >> 6;
    uint64_t word = m64[w] & (~0ull >> (63u - ((b - 1u) & 63u)));
    for (;;) {
        if (word) { const uint32_t q = w * 64u + 63u - (uint32_t)__builtin_clzll(word); return q >= a ? q : 0xFFFFFFFFu; }
        if (w * 64u <= a) return 0xFFFFFFFFu;
        w--;
        word = m64[w];
    }
}


// Four lanes per read: lane `sub` (0..3) classifies span bases w = sub, sub + 4, ... of the read at s[0, n) and the
// partial results are merged with two butterfly steps inside the quad — the same bits as sgc_pack_one (sgc_format.h),
// in a quarter of the serial steps.  All four lanes return the merged (span, status).  `valid` = the quad has a read.
template <class Ptr>
__device__ __forceinline__ void pack_quad(Ptr s, uint32_t n, bool valid, uint32_t sub, uint32_t L, int reverse, uint32_t o, int recursion,
                                          uint64_t &span, uint64_t &status) {
    const uint32_t K = L + 2;
    const bool c_ok = valid && o + L <= n;
    const bool p_ok = c_ok && recursion && (o + 1 + L <= n);
    const bool m_ok = p_ok && o >= 1;
    uint32_t lo = 0, hi = 0;          // span bits
    uint32_t cnt = 0;                 // invalid bases per window: M | C << 8 | P << 16; "bad" (non-ACGT, non-N) flags at bits 24..26
    uint32_t npos = 0;                // 2 + window position of an 'N', per window, 8 bits each
    if (c_ok) {
        for (uint32_t w = sub; w < K; w += 4) {
            const int32_t p = (int32_t)o - 1 + (int32_t)w;
            if (p < 0 || (uint32_t)p >= n) continue;      // only reachable for windows already out of bounds
            const uint8_t b = reverse ? s[n - 1 - (uint32_t)p] : s[p];
            const uint32_t code = reverse ? sgc_base_code_rc(b) : sgc_base_code(b);
            if (code < 4) { if (w < 16) lo |= code << (2 * w); else hi |= code << (2 * (w - 16)); continue; }
#pragma unroll
            for (int k = 0; k < 3; k++) {                 // span base w sits at window position w (M), w-1 (C), w-2 (P)
                const int32_t j = (int32_t)w - k;
                if (j < 0 || j >= (int32_t)L) continue;
                cnt += 1u << (8 * k);
                if (code == 4) npos |= (2u + (uint32_t)j) << (8 * k); else cnt |= 1u << (24 + k);
            }
        }
    }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        lo |= __shfl_xor(lo, m, 64); hi |= __shfl_xor(hi, m, 64); npos |= __shfl_xor(npos, m, 64);
        const uint32_t oc = __shfl_xor(cnt, m, 64);
        cnt = ((cnt & 0x00FFFFFFu) + (oc & 0x00FFFFFFu)) | ((cnt | oc) & 0x07000000u);
    }
    span = c_ok ? ((uint64_t)lo | ((uint64_t)hi << 32)) : 0ull;
    const bool ok[3] = {m_ok, c_ok, p_ok};
    uint32_t st[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t ninv = (cnt >> (8 * k)) & 0xFFu;
        if (!ok[k] || ninv >= 2 || ((cnt >> (24 + k)) & 1u)) st[k] = SGC_STATE_DEAD;
        else if (ninv == 0) st[k] = SGC_STATE_CLEAN;
        else st[k] = (npos >> (8 * k)) & 0xFFu;
    }
    status = (uint64_t)st[1] + (uint64_t)K * ((uint64_t)st[2] + (uint64_t)K * (uint64_t)st[0]);
}

struct fq_args {
    const uint8_t *text;          // the part: whole lines
    uint64_t n;                   // its bytes
    const uint32_t *tile_base;    // newlines before each tile; tile_base[tiles] = total
    uint64_t first_line;          // global line number of the part's first line (only for error reports and the phase)
    uint64_t *recs;               // record r of the part -> recs[r] (rec16: two words)
    unsigned long long *err;      // err[0]: ~0 - (smallest 1-based number of a line whose marker byte is wrong), 0 = none; err[1]: flags
    uint32_t tiles, expect_nl;    // expect_nl: newline count the host announced
    uint32_t n_lines;             // lines of the part as the host counts them (newlines + an unterminated last line): recs[] holds
                                  // exactly the sequence lines among them, and nothing beyond is ever written
    uint32_t L, o, halo;          // halo: bytes staged beyond the tile, multiple of 64, <= FQ_HALO_MAX
    int reverse, recursion;
    uint32_t dbg;                 // timing-only ablations (results wrong): 1 = no packing, 2 = no listing/validation either
};

template <bool REC16>
__global__ void __launch_bounds__(FQ_THREADS, 8) __attribute__((amdgpu_num_sgpr(80))) k_fastq_pack(const fq_args a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[FQ_TILE + FQ_HALO_MAX];
    __shared__ __attribute__((aligned(8))) uint16_t masks[FQ_PIECES];      // newline mask of every 16-byte piece
    __shared__ uint32_t list[FQ_CAP], wtmp[17];
    const uint32_t t = threadIdx.x, tile = blockIdx.x;
    const uint64_t tile0 = (uint64_t)tile * FQ_TILE;
    const uint32_t ph = (uint32_t)(a.first_line & 3u), cap = a.o + a.L + 2u;
    // staged region: absolute [r0, r0 + HB + FQ_TILE + HF); HB bytes before the tile (reverse strand), HF after (forward)
    const uint32_t HB = a.reverse ? (tile0 >= a.halo ? a.halo : 0u) : 0u, HF = a.reverse ? 0u : a.halo;
    const uint64_t r0 = tile0 - HB;
    const uint32_t span = HB + FQ_TILE + HF;
    const uint32_t staged = (uint32_t)(a.n - r0 < span ? a.n - r0 : span);          // bytes that exist
    const uint32_t n_pieces = span / 16u;
    uint32_t mk[FQ_PIECES / FQ_THREADS + 1];
#pragma unroll
    for (uint32_t k = 0; k < FQ_PIECES / FQ_THREADS + 1; k++) {
        const uint32_t piece = k * FQ_THREADS + t;
        if (piece < n_pieces) {
            const uint64_t base = r0 + (uint64_t)piece * 16;
            mk[k] = base < a.n ? nl_mask16(a.text, base, a.n, stage + piece * 16) : 0u;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < FQ_PIECES / FQ_THREADS + 1; k++) {
        const uint32_t piece = k * FQ_THREADS + t;
        if (piece < n_pieces) masks[piece] = (uint16_t)mk[k];
    }
    __syncthreads();
    const uint64_t *m64 = reinterpret_cast<const uint64_t *>(masks);
    // lane t owns tile bytes [64 t, 64 t + 64)
    const uint64_t mine = m64[(HB >> 6) + t];
    uint32_t total;
    const uint32_t before = wg_scan_1024((uint32_t)__popcll(mine), wtmp, &total);
    const uint32_t l0 = a.tile_base[tile], l1 = l0 + total;          // local line numbers ended by this tile's newlines
    if (tile == 0 && t == 0) {
        if (a.tile_base[a.tiles] != a.expect_nl) atomicOr(&a.err[1], 1ull);
        // marker byte of the part's first line
        const uint8_t c = stage[HB];
        if ((ph == 0 && c != '@') || (ph == 2 && c != '+')) atomicMax(&a.err[0], ~0ull - ((unsigned long long)a.first_line + 1ull));
    }
    // the sequence lines this tile owns: [rlo, rhi) as record numbers of the part
    const uint32_t n_lines = a.n_lines;
    uint32_t rlo, rhi;
    if (a.reverse) {
        rlo = seq_lines_before(ph, l0 < n_lines ? l0 : n_lines);
        rhi = seq_lines_before(ph, l1 < n_lines ? l1 : n_lines);
    } else {
        rlo = tile == 0 ? 0u : seq_lines_before(ph, l0 + 1u < n_lines ? l0 + 1u : n_lines);
        rhi = seq_lines_before(ph, l1 + 1u < n_lines ? l1 + 1u : n_lines);
    }
    if (SGC_DBG(a.dbg, 2u)) return;
    // Reverse strand: a sequence line is listed by the newline that CLOSES it, so a stream that breaks off inside a sequence line
    // (its last line has no '\n': a truncated record, which the host reports from the line count) would leave that record's slot
    // unwritten although it is counted — it gets the all-dead record (as k_fastq_lines does).
    if (a.reverse && tile == a.tiles - 1u && t == 0 && n_lines > a.tile_base[a.tiles] && ((ph + n_lines - 1u) & 3u) == 1u) {
        const uint64_t r = seq_lines_before(ph, n_lines - 1u), K = a.L + 2u;
        const uint64_t dead = SGC_STATE_DEAD * (1u + K + K * K);
        if (REC16) { a.recs[2 * r] = 0; a.recs[2 * r + 1] = dead; }
        else a.recs[r] = dead << (2 * K);
    }
    for (uint32_t rb = rlo; rb < rhi || rb == rlo; rb += FQ_CAP) {
        // list the owned sequence lines [rb, rb + FQ_CAP): forward = stage offset of the line start, reverse = of its '\n'
        uint64_t mm = mine;
        uint32_t i = l0 + before;                       // local number of the line my next newline ends
        while (mm) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
            const uint32_t pos = HB + 64u * t + bit;    // stage offset of the newline
            const uint32_t phase = (ph + i) & 3u;
            if (rb == rlo) {
                // marker byte of the next line (lines 4k start with '@', lines 4k + 2 with '+')
                const uint64_t nxt = r0 + pos + 1u;
                if (nxt < a.n && (phase == 3u || phase == 1u)) {
                    const uint8_t c = pos + 1u < staged ? stage[pos + 1u] : a.text[nxt];
                    if (c != (phase == 3u ? '@' : '+')) atomicMax(&a.err[0], ~0ull - ((unsigned long long)a.first_line + i + 2ull));
                }
            }
            if (a.reverse) {
                if (phase == 1u) { const uint32_t r = seq_lines_before(ph, i); if (r - rb < FQ_CAP) list[r - rb] = pos; }
            } else if (phase == 0u && r0 + pos + 1u < a.n) {
                const uint32_t r = seq_lines_before(ph, i + 1u);
                if (r - rb < FQ_CAP) list[r - rb] = pos + 1u;
            }
            i++;
        }
        if (!a.reverse && tile == 0 && t == 0 && ph == 1u && rb == 0 && staged) list[0] = HB;    // the part starts with a sequence line
        __syncthreads();
        const uint32_t nrec = rhi - rb < FQ_CAP ? rhi - rb : FQ_CAP;
        // four lanes per listed line (a tile of ordinary reads holds ~200: one sweep of the 1024 lanes)
        for (uint32_t i0 = 0; i0 < nrec && !SGC_DBG(a.dbg, 1u); i0 += FQ_THREADS / 4u) {
            const uint32_t i = i0 + (t >> 2), sub = t & 3u;
            const bool have = i < nrec;
            uint32_t so = 0, len = 0; uint64_t go = 0; bool in_lds = true;
            if (have && !a.reverse) {
                const uint32_t st = list[i];
                const uint32_t lim = st + cap < staged ? st + cap : staged;
                const uint32_t q = find_nl_fwd(m64, st, lim);
                so = st;
                if (q < lim || (lim == staged && r0 + staged == a.n)) {
                    len = q - st;
                    if (len && stage[q - 1] == 0x0D) len--;
                } else if (lim - st == cap) {
                    len = cap;
                } else {
                    const uint64_t g0 = r0 + st;
                    uint64_t g = g0;
                    while (g < a.n && g - g0 < cap && a.text[g] != 0x0A) g++;
                    len = (uint32_t)(g - g0);
                    if (len && len < cap && a.text[g - 1] == 0x0D) len--;
                    go = g0; in_lds = false;
                }
            } else if (have) {
                uint32_t e = list[i];
                if (e && stage[e - 1] == 0x0D) e--;
                const uint32_t lo = e > cap ? e - cap : 0u;
                const uint32_t p = find_nl_bwd(m64, lo, e);
                if (p != 0xFFFFFFFFu) { so = p + 1u; len = e - (p + 1u); }
                else if (e - lo == cap) { so = lo; len = cap; }
                else if (r0 == 0) { so = 0; len = e; }
                else {
                    const uint64_t ge = r0 + e;
                    uint64_t g = ge;
                    while (g > 0 && ge - g < cap && a.text[g - 1] != 0x0A) g--;
                    go = g; len = (uint32_t)(ge - g); in_lds = false;
                }
            }
            uint64_t spn, sts, spn2, sts2;
            pack_quad(stage + so, len, have && in_lds, sub, a.L, a.reverse, a.o, a.recursion, spn, sts);
            if (__builtin_amdgcn_ballot_w64(have && !in_lds)) {
                pack_quad(a.text + go, len, have && !in_lds, sub, a.L, a.reverse, a.o, a.recursion, spn2, sts2);
                if (!in_lds) { spn = spn2; sts = sts2; }
            }
            if (have && sub == 0) {
                const uint64_t r = (uint64_t)rb + i;
                if (REC16) { a.recs[2 * r] = spn; a.recs[2 * r + 1] = sts; }
                else a.recs[r] = spn | (sts << (2 * (a.L + 2)));
            }
        }
        if (rhi - rb <= FQ_CAP) break;
        __syncthreads();                                 // list[] is rewritten by the next round
    }
}

uint32_t sgc_fastq_tiles(uint64_t n) { return (uint32_t)((n + FQ_TILE - 1) / FQ_TILE); }

uint64_t sgc_fastq_records(uint64_t first_line, uint64_t n_lines) {
    const uint64_t ph = first_line & 3u;
    return ((ph + n_lines + 2u) >> 2) - ((ph + 2u) >> 2);
}

void sgc_launch_fastq_count(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t *tile_scratch) {
    if (n == 0) return;
    const uint32_t tiles = sgc_fastq_tiles(n);
    hipLaunchKernelGGL(k_fastq_count, dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tile_scratch, tiles);
}

void sgc_launch_fastq_pack(hipStream_t st, const uint8_t *text, uint64_t n, const uint32_t *tile_scratch, uint64_t first_line,
                           uint32_t expect_nl, uint32_t n_lines, uint32_t L, bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs,
                           unsigned long long *err, uint32_t dbg) {
    if (n == 0) return;
    fq_args a;
    a.dbg = dbg;
    a.text = text; a.n = n; a.tile_base = tile_scratch; a.first_line = first_line; a.recs = recs; a.err = err;
    a.tiles = sgc_fastq_tiles(n); a.expect_nl = expect_nl; a.n_lines = n_lines; a.L = L; a.o = o;
    const uint64_t want = ((uint64_t)o + L + 2u + 63u) & ~63ull;
    a.halo = (uint32_t)(want < FQ_HALO_MAX ? want : FQ_HALO_MAX);
    a.reverse = reverse; a.recursion = recursion;
    if (rec16) hipLaunchKernelGGL((k_fastq_pack<true>), dim3(a.tiles), dim3(FQ_THREADS), 0, st, a);
    else hipLaunchKernelGGL((k_fastq_pack<false>), dim3(a.tiles), dim3(FQ_THREADS), 0, st, a);
}

// ------------------------------------------------------------------------------------------------
// raw reads (bytes + offsets) -> records.  A workgroup owns PR_READS consecutive reads; their bytes are one
// contiguous range of the input, which is streamed through LDS with coalesced 16-byte loads; each lane then
// packs its read from LDS.  Ranges larger than the stage (long reads) fall back to global loads per read.
// ------------------------------------------------------------------------------------------------
#define PR_THREADS 256u
#define PR_READS 256u
#define PR_STAGE 65536u
template <bool REC16>
__global__ void __launch_bounds__(PR_THREADS) k_pack_reads_lds(const uint8_t *__restrict__ seqs,
                                                               const uint64_t *__restrict__ offsets, uint64_t n, uint32_t L,
                                                               int reverse, uint32_t o, int recursion,
                                                               uint64_t *__restrict__ recs) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[PR_STAGE];
    const uint32_t t = threadIdx.x;
    const uint64_t i0 = (uint64_t)blockIdx.x * PR_READS;
    const uint64_t i1 = i0 + PR_READS < n ? i0 + PR_READS : n;
    const uint64_t b0 = offsets[i0], b1 = offsets[i1];
    const uint64_t a0 = b0 & ~15ull;                       // aligned start of the staged range
    const bool staged = b1 - a0 <= PR_STAGE;
    if (staged) {
        for (uint64_t x = a0 + (uint64_t)t * 16; x < b1; x += PR_THREADS * 16) {
            // the last 16-byte piece may run past b1 but never past the next 16-byte boundary of the allocation:
            // read it byte-wise when it would cross the end of the input
            if (x + 16 <= b1 || x + 16 <= offsets[n])
                *reinterpret_cast<u32x4 *>(stage + (x - a0)) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(seqs + x));
            else
                for (uint32_t k = 0; k < 16 && x + k < b1; k++) stage[x - a0 + k] = seqs[x + k];
        }
    }
    __syncthreads();
    // one lane per read: every lane has work, so the serial per-lane packer keeps all lanes busy
    const uint64_t i = i0 + t;
    if (i >= i1) return;
    const uint64_t b = offsets[i], e = offsets[i + 1];
    uint64_t span, status;
    if (staged) sgc_pack_one(stage + (b - a0), e - b, L, reverse, o, recursion, span, status);
    else sgc_pack_one(seqs + b, e - b, L, reverse, o, recursion, span, status);
    if (REC16) { recs[2 * i] = span; recs[2 * i + 1] = status; }
    else recs[i] = span | (status << (2 * (L + 2)));
}

void sgc_launch_pack_reads_lds(hipStream_t st, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L,
                               bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs) {
    if (n == 0) return;
    const unsigned grid = (unsigned)((n + PR_READS - 1) / PR_READS);
    if (rec16)
        hipLaunchKernelGGL((k_pack_reads_lds<true>), dim3(grid), dim3(PR_THREADS), 0, st, seqs, offsets, n, L, reverse, o, recursion, recs);
    else
        hipLaunchKernelGGL((k_pack_reads_lds<false>), dim3(grid), dim3(PR_THREADS), 0, st, seqs, offsets, n, L, reverse, o, recursion, recs);
}
