// sgc_kernels.hip — gfx950 (CDNA4, wave64) kernels of the sgRNA count path.
//
// What they restate (reference noamteyssier/sgcount v0.1.35):
//   sgc_assign          Counter::assign             src/counter.rs:96-140  (C-exact, C-1mm, P-exact, P-1mm, M-exact, M-1mm)
//   table_find          Library::contains / alias   src/library.rs:34-46 ; Permuter::contains src/permutes.rs:55-57
//   count kernels       Counter::count fold         src/counter.rs:211-236
//   pack kernels        Counter::apply_trim/bounds  src/counter.rs:144-204
// This is integer / hash / atomic work: HBM- and L2-latency-bound, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_device.h"
#include "sgc_format.h"
#include "sgc_kernels.h"

// ------------------------------------------------------------------------------------------------
// variant 1: lookup kernel writes one gid per read (coalesced u32 stores, no count atomics); a second
// kernel builds the histogram in LDS, one guide-range slice per pass, and flushes each slice with
// contiguous (256 B per wave-instruction) device-scope atomics.  Scattered device-scope atomics run at
// ~20 G requests/s chip-wide on MI355X (they execute at the memory side), LDS atomics do not.
// ------------------------------------------------------------------------------------------------
template <bool PACKED, bool REC16>
__global__ void __launch_bounds__(256) k_lookup_gids(const uint64_t *__restrict__ recs, uint64_t n, uint32_t L,
                                                     sgc_table_view lib, sgc_table_view perm, int one_mm,
                                                     uint32_t *__restrict__ gids,
                                                     unsigned long long *__restrict__ matched) {
    uint64_t local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t span, status;
        load_record<REC16>(recs, i, L, span, status);
        const uint32_t g = sgc_assign<PACKED>(span, status, L, lib, perm, one_mm != 0);
        gids[i] = g;
        local += g != SGC_NONE;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(matched, (unsigned long long)local);
}

#define SGC_HIST_SLICE 36864u   // u32 counters per LDS slice (144 KiB of the CU's 160 KiB)
__global__ void __launch_bounds__(1024) k_hist_slices(const uint32_t *__restrict__ gids, uint64_t n, uint32_t n_guides,
                                                      uint32_t *__restrict__ counts) {
    __shared__ uint32_t h[SGC_HIST_SLICE];
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = 0; base < n_guides; base += SGC_HIST_SLICE) {
        for (uint32_t j = threadIdx.x; j < SGC_HIST_SLICE; j += 1024) h[j] = 0;
        __syncthreads();
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) {
            const uint32_t r = gids[i] - base;
            if (r < SGC_HIST_SLICE) atomicAdd(&h[r], 1u);
        }
        __syncthreads();
        const uint32_t lim = n_guides - base < SGC_HIST_SLICE ? n_guides - base : SGC_HIST_SLICE;
        for (uint32_t j = threadIdx.x; j < lim; j += 1024) {
            const uint32_t v = h[j];
            if (v) atomicAdd(&counts[base + j], v);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// point lookups (sgc_lookup): keys already packed by the host
// ------------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ void k_lookup(const uint64_t *__restrict__ keys, uint64_t n, sgc_table_view lib, sgc_table_view perm,
                         int which, int has_perm, int32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint32_t g = SGC_NONE;
    if (key != SGC_EMPTY) {
        if (which == 0 || which == 2) g = table_find<PACKED>(lib, key);
        if (g == SGC_NONE && (which == 1 || which == 2) && has_perm) g = table_find<PACKED>(perm, key);
    }
    out[i] = g == SGC_NONE ? -1 : (int32_t)g;
}

// counts64 += counts32; counts32 = 0
__global__ void k_fold(uint32_t *__restrict__ c32, unsigned long long *__restrict__ c64, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { c64[i] += c32[i]; c32[i] = 0; }
}

// fold and export in one pass: out = counts64 (+= counts32) | total_reads | matched_reads
__global__ void k_export(uint32_t *__restrict__ c32, unsigned long long *__restrict__ c64,
                         const unsigned long long *__restrict__ matched, unsigned long long total, uint32_t n,
                         unsigned long long *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const unsigned long long v = c64[i] + c32[i]; c64[i] = v; c32[i] = 0; out[i] = v; }
    if (i == n) out[n] = total;
    if (i == n + 1) out[n + 1] = *matched;
}

// ------------------------------------------------------------------------------------------------
// pack kernel, raw read bytes -> records (one read per thread; v0)
// ------------------------------------------------------------------------------------------------
template <bool REC16>
__global__ void k_pack_reads(const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ offsets, uint64_t n,
                             uint32_t L, int reverse, uint32_t o, int recursion, uint64_t *__restrict__ recs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = offsets[i], e = offsets[i + 1];
    uint64_t span, status;
    sgc_pack_one(seqs + b, e - b, L, reverse, o, recursion, span, status);
    if (REC16) { recs[2 * i] = span; recs[2 * i + 1] = status; }
    else recs[i] = span | (status << (2 * (L + 2)));
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap) {
    uint64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}

void sgc_launch_lookup_gids(hipStream_t st, const uint64_t *recs, uint64_t n, uint32_t L, bool rec16,
                            const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm, uint32_t *gids,
                            unsigned long long *matched) {
    if (n == 0) return;
    const unsigned block = 256, grid = grid_for(n, block, 256 * 8 * 4);
    const bool packed = lib.gid_bits != 0;
#define SGC_GO(P, R) \
    hipLaunchKernelGGL((k_lookup_gids<P, R>), dim3(grid), dim3(block), 0, st, recs, n, L, lib, perm, (int)one_mm, gids, matched)
    if (packed) { if (rec16) SGC_GO(true, true); else SGC_GO(true, false); }
    else        { if (rec16) SGC_GO(false, true); else SGC_GO(false, false); }
#undef SGC_GO
}

void sgc_launch_hist_slices(hipStream_t st, const uint32_t *gids, uint64_t n, uint32_t n_guides, uint32_t *counts) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_hist_slices, dim3(256), dim3(1024), 0, st, gids, n, n_guides, counts);
}

void sgc_launch_lookup(hipStream_t st, const uint64_t *keys, uint64_t n, const sgc_table_view &lib,
                       const sgc_table_view &perm, int which, bool has_perm, int32_t *out) {
    if (n == 0) return;
    const unsigned block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    if (lib.gid_bits != 0)
        hipLaunchKernelGGL((k_lookup<true>), dim3(grid), dim3(block), 0, st, keys, n, lib, perm, which, (int)has_perm, out);
    else
        hipLaunchKernelGGL((k_lookup<false>), dim3(grid), dim3(block), 0, st, keys, n, lib, perm, which, (int)has_perm, out);
}

void sgc_launch_fold(hipStream_t st, uint32_t *c32, unsigned long long *c64, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fold, dim3((n + 255) / 256), dim3(256), 0, st, c32, c64, n);
}

void sgc_launch_export(hipStream_t st, uint32_t *c32, unsigned long long *c64, const unsigned long long *matched,
                       unsigned long long total, uint32_t n, unsigned long long *out) {
    hipLaunchKernelGGL(k_export, dim3((n + 2 + 255) / 256), dim3(256), 0, st, c32, c64, matched, total, n, out);
}

void sgc_launch_pack_reads(hipStream_t st, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L,
                           bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs) {
    if (n == 0) return;
    const unsigned block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    if (rec16)
        hipLaunchKernelGGL((k_pack_reads<true>), dim3(grid), dim3(block), 0, st, seqs, offsets, n, L, reverse, o, recursion, recs);
    else
        hipLaunchKernelGGL((k_pack_reads<false>), dim3(grid), dim3(block), 0, st, seqs, offsets, n, L, reverse, o, recursion, recs);
}
