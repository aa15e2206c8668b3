// sgc_fastq.hip — FASTQ ingest on the GPU: record boundaries, window extraction and 2-bit packing from raw
// FASTQ text resident in HBM (replaces fxread + Counter::apply_trim, reference src/counter.rs:144-204, for the
// sgc_sample_push_fastq entry point), and the packer for raw reads (bytes + offsets).
//
// FASTQ is four lines per record, so record boundaries are a matter of counting newlines:
//   pass A  k_fastq_count   per 64 KiB tile: number of '\n'                      (reads the text once)
//           k_scan_tiles    exclusive prefix over the tiles (one workgroup)
//   pass B  k_fastq_pack    per tile: the tile is staged in LDS; global line number of every '\n'; the lane that
//                           owns the newline ENDING A SEQUENCE LINE (line % 4 == 1) finds the line start, extracts
//                           the L+2-base span at the sample's offset from LDS and writes record r = line / 4
// The text chunk must start at a record boundary and hold whole records; a missing final '\n' is tolerated.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sgc_format.h"
#include "sgc_kernels.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FQ_THREADS 1024u
#define FQ_SUB 4u                           // sub-tiles per tile; one sub-tile = 16 bytes per lane
#define FQ_TILE (FQ_THREADS * 16u * FQ_SUB) // 64 KiB

// 16-bit mask of '\n' among the 16 bytes at text[base ..) (bytes past n read as 0); optionally stages them in LDS
__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *__restrict__ text, uint64_t base, uint64_t n, uint8_t *stage) {
    uint32_t m = 0;
    if (base + 16 <= n && ((uintptr_t)(text + base) & 15) == 0) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(text + base));
        if (stage) *reinterpret_cast<u32x4 *>(stage) = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // zero-byte detection on (w ^ 0x0A0A0A0A) as a cheap "any newline in this word?" test (it may also
            // fire for a 0x0B right after a newline, so the bytes are then checked one by one)
            const uint32_t x = w[k] ^ 0x0A0A0A0Au;
            if ((x - 0x01010101u) & ~x & 0x80808080u) {
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (((w[k] >> (8 * b)) & 0xFFu) == 0x0Au) m |= 1u << (4 * k + b);
            }
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

// ------------------------------------------------------------------------------------------------
// Cooperative packing: the L+2 bytes of one read's span are classified by L+2 lanes of a half-wave at once
// (wave ballots give the 2-bit codes and the non-ACGT masks), instead of one lane looping over them.  Same bits
// as sgc_pack_one (sgc_format.h); all lanes of the half-wave return the same (span, status).
// s: start of the read (LDS or global), n: its length.  Must be called by all 64 lanes of the wave; the two
// halves may work on different reads (or idle with n = 0, s = any valid pointer).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread_bits(uint32_t x) {       // bit i -> bit 2i
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

template <class Ptr>
__device__ __forceinline__ void pack_coop(Ptr s, uint64_t n, uint32_t L, int reverse, uint32_t o, int recursion,
                                          uint64_t &span, uint64_t &status) {
    const uint32_t lane = threadIdx.x & 63u, w = lane & 31u, half = lane >> 5;
    const uint32_t K = L + 2;
    const int64_t pz = (int64_t)o - 1 + (int64_t)w;
    const bool exists = w < K && pz >= 0 && (uint64_t)pz < n;
    uint32_t code = 0;
    if (exists) {
        const uint8_t b = reverse ? s[n - 1 - (uint64_t)pz] : s[pz];
        code = reverse ? sgc_base_code_rc(b) : sgc_base_code(b);
    }
    const uint32_t b0 = (uint32_t)(__ballot(exists && code < 4 && (code & 1u)) >> (32 * half));
    const uint32_t b1 = (uint32_t)(__ballot(exists && code < 4 && (code & 2u)) >> (32 * half));
    const uint32_t isn = (uint32_t)(__ballot(exists && code == 4) >> (32 * half));
    const uint32_t bad = (uint32_t)(__ballot(exists && code == 5) >> (32 * half));
    const bool c_ok = (uint64_t)o + L <= n;
    const bool p_ok = c_ok && recursion && ((uint64_t)o + 1 + L <= n);
    const bool m_ok = p_ok && o >= 1;
    span = c_ok ? (spread_bits(b0) | (spread_bits(b1) << 1)) : 0;
    const uint32_t wmask = L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
    const bool ok[3] = {m_ok, c_ok, p_ok};
    uint32_t st[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t inv = ((isn | bad) >> k) & wmask, nn = (isn >> k) & wmask, bb = (bad >> k) & wmask;
        const uint32_t cnt = __popc(inv);
        if (!ok[k] || !c_ok || cnt >= 2 || bb) st[k] = SGC_STATE_DEAD;
        else if (cnt == 0) st[k] = SGC_STATE_CLEAN;
        else st[k] = 2u + (uint32_t)(__ffs(nn) - 1);
    }
    status = (uint64_t)st[1] + (uint64_t)K * ((uint64_t)st[2] + (uint64_t)K * (uint64_t)st[0]);
}

__global__ void __launch_bounds__(FQ_THREADS) k_fastq_count(const uint8_t *__restrict__ text, uint64_t n,
                                                            uint32_t *__restrict__ tile_nl) {
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    const uint32_t t = threadIdx.x;
    const uint64_t tile0 = (uint64_t)blockIdx.x * FQ_TILE;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t s = 0; s < FQ_SUB; s++) {                 // 16 B per lane per load: fully coalesced
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

#define FQ_LIST 1024u                       // sequence lines found per sub-tile (<= one per 16-byte piece)
template <bool REC16>
__global__ void __launch_bounds__(FQ_THREADS) k_fastq_pack(const uint8_t *__restrict__ text, uint64_t n,
                                                           const uint32_t *__restrict__ tile_base, uint32_t L,
                                                           int reverse, uint32_t o, int recursion,
                                                           uint64_t *__restrict__ recs, uint32_t dbg) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[FQ_TILE];   // the whole tile, filled sub-tile by sub-tile
    __shared__ __attribute__((aligned(8))) uint16_t masks[FQ_TILE / 16];   // newline mask of every 16-byte piece
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    __shared__ uint32_t l_q[FQ_LIST], l_line[FQ_LIST], n_list;   // newline offset in the tile, global line index
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint64_t tile0 = (uint64_t)blockIdx.x * FQ_TILE;
    const uint64_t *masks64 = reinterpret_cast<const uint64_t *>(masks);
    uint32_t carry = tile_base[blockIdx.x];                 // lines completed before the current sub-tile
    for (uint32_t s = 0; s < FQ_SUB; s++) {
        const uint32_t piece = s * FQ_THREADS + t;
        const uint64_t base = tile0 + (uint64_t)piece * 16;
        const uint32_t m = base < n ? nl_mask16(text, base, n, stage + piece * 16) : 0;
        const uint32_t c = __popc(m);
        masks[piece] = (uint16_t)m;
        if (t == 0) n_list = 0;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        if (lane == 63) wsum[t >> 6] = incl;
        __syncthreads();
        uint32_t before = incl - c, total = 0;
        for (uint32_t w = 0; w < FQ_THREADS / 64; w++) {
            if (w < (t >> 6)) before += wsum[w];
            total += wsum[w];
        }
        // every newline that ends a SEQUENCE line (line % 4 == 1) goes to the list; at most one per piece can be
        // missed only if a piece held two of them, which needs a record shorter than 16 bytes: handled inline
        uint32_t line = carry + before;                      // global index of the line ended by my first newline
        uint32_t mm = m, mine = 0xFFFFFFFFu;
        while (mm) {
            const uint32_t k = __ffs(mm) - 1;
            mm &= mm - 1;
            if ((line & 3u) == 1u) {
                if (mine == 0xFFFFFFFFu) mine = (k << 28) | 0;   // placeholder, filled below
                const uint32_t at = atomicAdd(&n_list, 1u);
                l_q[at] = piece * 16 + k;
                l_line[at] = line;
            }
            line++;
        }
        __syncthreads();
        // one half-wave per listed sequence line
        const uint32_t nl = n_list;
        for (uint32_t i0 = 0; i0 < nl; i0 += FQ_THREADS / 32) {
            const uint32_t idx = i0 + (t >> 5);
            const bool have = idx < nl;
            const uint32_t qo = have ? l_q[idx] : 0;             // offset of the '\n' inside the tile
            // start of the line: one past the previous newline.  Search the piece masks backwards, four pieces
            // (one u64) at a time; if the tile holds none, the line began in an earlier tile: walk back in memory.
            uint32_t po = 0;
            bool in_tile = true;
            if (have) {
                const uint32_t pc = qo >> 4, kq = qo & 15u;
                const uint32_t below = (uint32_t)masks[pc] & ((1u << kq) - 1u);
                if (below) {
                    po = pc * 16 + (31 - __clz(below)) + 1;
                } else {
                    // pieces [0, pc): first the remainder of pc's group of four, then whole groups
                    int g = (int)(pc >> 2);
                    uint64_t word = masks64[g] & ((1ull << (16 * (pc & 3u))) - 1ull);
                    while (word == 0 && g > 0) word = masks64[--g];
                    if (word) {
                        const uint32_t hb = 63 - __clzll((long long)word);       // highest newline bit in the group
                        po = (uint32_t)g * 64 + hb + 1;
                    } else {
                        in_tile = false;
                    }
                }
            }
            uint64_t span = 0, status = 0;
            const uint64_t q = tile0 + qo;
            if (in_tile) {
                pack_coop(stage + po, have ? (uint64_t)(qo - po) : 0, L, reverse, o, recursion, span, status);
            } else {
                uint64_t p = tile0;
                while (p > 0 && text[p - 1] != 0x0A) p--;
                pack_coop(text + p, q - p, L, reverse, o, recursion, span, status);
            }
            if (have && (t & 31u) == 0) {
                const uint64_t r = l_line[idx] >> 2;
                if (REC16) { recs[2 * r] = span; recs[2 * r + 1] = status; }
                else recs[r] = span | (status << (2 * (L + 2)));
            }
        }
        carry += total;
        __syncthreads();                                     // wsum / list are reused by the next sub-tile
    }
    (void)dbg;
}

void sgc_launch_fastq(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t *tile_scratch, uint32_t L, bool rec16,
                      int reverse, uint32_t o, int recursion, uint64_t *recs) {
    if (n == 0) return;
    const uint32_t tiles = (uint32_t)((n + FQ_TILE - 1) / FQ_TILE);
    hipLaunchKernelGGL(k_fastq_count, dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tile_scratch, tiles);
    static const uint32_t dbg = getenv("SGC_FQ_DBG") ? (uint32_t)atoi(getenv("SGC_FQ_DBG")) : 0;   // timing ablations only
    if (rec16)
        hipLaunchKernelGGL((k_fastq_pack<true>), dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch, L, reverse, o,
                           recursion, recs, dbg);
    else
        hipLaunchKernelGGL((k_fastq_pack<false>), dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch, L, reverse, o,
                           recursion, recs, dbg);
}

uint32_t sgc_fastq_tiles(uint64_t n) { return (uint32_t)((n + FQ_TILE - 1) / FQ_TILE); }

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
    // one lane per read: here every lane has work (unlike the FASTQ kernel, where a sequence line ends in only
    // one of ~20 lanes' pieces), so the serial per-lane packer keeps all lanes busy
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
