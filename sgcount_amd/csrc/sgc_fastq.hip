// sgc_fastq.hip — FASTQ ingest on the GPU: record boundaries, window extraction and 2-bit packing from raw
// FASTQ text resident in HBM (replaces fxread + Counter::apply_trim, reference src/counter.rs:144-204, for the
// sgc_sample_push_fastq entry point).
//
// FASTQ is four lines per record, so record boundaries are a matter of counting newlines:
//   pass A  k_fastq_count   per 4 KiB tile: number of '\n'                       (reads the text once)
//           k_scan_tiles    exclusive prefix over the tiles (one workgroup)
//   pass B  k_fastq_pack    per tile: global line number of every '\n'; the lane that owns the newline ENDING
//                           A SEQUENCE LINE (line % 4 == 1) finds the line start, extracts the L+2-base span
//                           at the sample's offset and writes record r = line / 4   (reads the text again)
// The text chunk must start at a record boundary and hold whole records; a missing final '\n' is tolerated.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_format.h"
#include "sgc_kernels.h"

#define FQ_TILE 4096u
#define FQ_THREADS 256u
#define FQ_BPT (FQ_TILE / FQ_THREADS)      // 16 bytes per lane

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *__restrict__ text, uint64_t base, uint64_t n) {
    // 16-bit mask of '\n' among text[base .. base+16) (bytes past n read as 0)
    uint32_t m = 0;
    if (base + 16 <= n && ((uintptr_t)(text + base) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (((w[k] >> (8 * b)) & 0xFFu) == '\n') m |= 1u << (4 * k + b);
    } else {
        for (uint32_t k = 0; k < 16; k++)
            if (base + k < n && text[base + k] == '\n') m |= 1u << k;
    }
    return m;
}

__global__ void __launch_bounds__(FQ_THREADS) k_fastq_count(const uint8_t *__restrict__ text, uint64_t n,
                                                            uint32_t *__restrict__ tile_nl) {
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * FQ_TILE + (uint64_t)t * FQ_BPT;
    uint32_t c = base < n ? __popc(nl_mask16(text, base, n)) : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((t & 63) == 0) wsum[t >> 6] = c;
    __syncthreads();
    if (t == 0) tile_nl[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
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

template <bool REC16>
__global__ void __launch_bounds__(FQ_THREADS) k_fastq_pack(const uint8_t *__restrict__ text, uint64_t n,
                                                           const uint32_t *__restrict__ tile_base, uint32_t L,
                                                           int reverse, uint32_t o, int recursion,
                                                           uint64_t *__restrict__ recs) {
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    __shared__ uint32_t masks[FQ_THREADS];
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint64_t tile0 = (uint64_t)blockIdx.x * FQ_TILE;
    const uint64_t base = tile0 + (uint64_t)t * FQ_BPT;
    const uint32_t m = base < n ? nl_mask16(text, base, n) : 0;
    const uint32_t c = __popc(m);
    masks[t] = m;
    // exclusive prefix of c inside the tile: wave scan, then the 4 wave totals
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t w = 0; w < (t >> 6); w++) before += wsum[w];
    uint32_t line = tile_base[blockIdx.x] + before;          // global index of the line ended by my first newline
    uint32_t mm = m;
    while (mm) {
        const uint32_t k = __ffs(mm) - 1;
        mm &= mm - 1;
        if ((line & 3u) == 1u) {                              // this newline ends a sequence line
            const uint64_t q = base + k;                      // position of the '\n'
            // start of the line = one past the previous newline: in my own 16 bytes, else in the tile (the
            // lanes' masks are in LDS), else — the line began in an earlier tile — walk back through memory
            uint64_t p;
            const uint32_t below = m & ((1u << k) - 1u);
            if (below) {
                p = base + (31 - __clz(below)) + 1;
            } else {
                int tt = (int)t - 1;
                while (tt >= 0 && masks[tt] == 0) tt--;
                if (tt >= 0) {
                    p = tile0 + (uint64_t)tt * FQ_BPT + (31 - __clz(masks[tt])) + 1;
                } else {
                    p = tile0;
                    while (p > 0 && text[p - 1] != '\n') p--;
                }
            }
            uint64_t span, status;
            sgc_pack_one(text + p, q - p, L, reverse, o, recursion, span, status);
            const uint64_t r = line >> 2;
            if (REC16) { recs[2 * r] = span; recs[2 * r + 1] = status; }
            else recs[r] = span | (status << (2 * (L + 2)));
        }
        line++;
    }
}

// the last record of a chunk that does not end with '\n' has an unterminated quality line only: every
// sequence line is newline-terminated, so nothing else is needed.

void sgc_launch_fastq(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t *tile_scratch, uint32_t L, bool rec16,
                      int reverse, uint32_t o, int recursion, uint64_t *recs) {
    if (n == 0) return;
    const uint32_t tiles = (uint32_t)((n + FQ_TILE - 1) / FQ_TILE);
    hipLaunchKernelGGL(k_fastq_count, dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tile_scratch, tiles);
    if (rec16)
        hipLaunchKernelGGL((k_fastq_pack<true>), dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch, L, reverse, o,
                           recursion, recs);
    else
        hipLaunchKernelGGL((k_fastq_pack<false>), dim3(tiles), dim3(FQ_THREADS), 0, st, text, n, tile_scratch, L, reverse, o,
                           recursion, recs);
}

uint32_t sgc_fastq_tiles(uint64_t n) { return (uint32_t)((n + FQ_TILE - 1) / FQ_TILE); }
