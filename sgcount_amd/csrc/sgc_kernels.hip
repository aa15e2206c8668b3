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

#include "sgc_format.h"
#include "sgc_kernels.h"

// ------------------------------------------------------------------------------------------------
// open-addressed lookup (linear probing; load factor <= 0.5 so a free slot always ends the chain)
// ------------------------------------------------------------------------------------------------
template <bool PACKED>
__device__ __forceinline__ uint32_t table_find(const sgc_table_view &t, uint64_t key) {
    const uint32_t mask = (1u << t.log2_slots) - 1u;
    uint32_t h = (uint32_t)(sgc_hash(key) >> (64 - t.log2_slots));
    for (;;) {
        const uint64_t s = t.slots[h];
        if (s == SGC_EMPTY) return SGC_NONE;
        if (PACKED) {
            if ((s >> t.gid_bits) == key) return (uint32_t)(s & ((1ull << t.gid_bits) - 1ull));
        } else {
            if (s == key) return t.vals[h];
        }
        h = (h + 1) & mask;
    }
}

// One window: exact, then single mismatch.  state: 0 clean, 1 dead, 2+j single 'N' at j.
template <bool PACKED>
__device__ __forceinline__ uint32_t window_assign(uint64_t key, uint32_t state, const sgc_table_view &lib,
                                                  const sgc_table_view &perm, bool one_mm) {
    if (state == SGC_STATE_CLEAN) {
        uint32_t g = table_find<PACKED>(lib, key);                 // src/counter.rs:111
        if (g == SGC_NONE && one_mm) g = table_find<PACKED>(perm, key);   // :113-116 (child -> parent -> alias)
        return g;
    }
    if (state == SGC_STATE_DEAD || !one_mm) return SGC_NONE;
    // exactly one 'N' at position j: the Hamming-1 guides are the (up to 4) substitutions at j;
    // src/permutes.rs:127-144 keeps the child only if its parent is unique.
    const uint32_t j = state - 2u;
    uint32_t hit = SGC_NONE, cnt = 0;
#pragma unroll
    for (uint64_t b = 0; b < 4; b++) {
        const uint32_t g = table_find<PACKED>(lib, key | (b << (2 * j)));
        if (g != SGC_NONE) { hit = g; cnt++; }
    }
    return cnt == 1 ? hit : SGC_NONE;
}

template <bool PACKED>
__device__ __forceinline__ uint32_t sgc_assign(uint64_t span, uint64_t status, uint32_t L, const sgc_table_view &lib,
                                               const sgc_table_view &perm, bool one_mm) {
    const uint64_t kmask = sgc_key_mask(L);
    uint32_t sC = 0, sP = 0, sM = 0;
    if (status != 0) {
        const uint32_t K = L + 2, st = (uint32_t)status;
        sC = st % K; sP = (st / K) % K; sM = st / (K * K);
    }
    uint32_t g = window_assign<PACKED>((span >> 2) & kmask, sC, lib, perm, one_mm);        // Centered
    if (g != SGC_NONE) return g;
    g = window_assign<PACKED>((span >> 4) & kmask, sP, lib, perm, one_mm);                  // Plus  (:123-125)
    if (g != SGC_NONE) return g;
    return window_assign<PACKED>(span & kmask, sM, lib, perm, one_mm);                      // Minus (:128-130)
}

template <bool REC16>
__device__ __forceinline__ void load_record(const uint64_t *recs, uint64_t i, uint32_t L, uint64_t &span,
                                            uint64_t &status) {
    if (REC16) {
        const ulonglong2 r = reinterpret_cast<const ulonglong2 *>(recs)[i];
        span = r.x; status = r.y;
    } else {
        const uint64_t r = recs[i];
        const uint32_t sh = 2 * (L + 2);
        span = r & ((1ull << sh) - 1ull);     // sh <= 50
        status = r >> sh;
    }
}

// ------------------------------------------------------------------------------------------------
// v0 count kernel: one record per thread (grid-stride), device-scope atomics on the count vector.
// Kept as the simple reference variant (SGC_VARIANT=0); the tuned path is below.
// ------------------------------------------------------------------------------------------------
template <bool PACKED, bool REC16>
__global__ void __launch_bounds__(256) k_count_direct(const uint64_t *__restrict__ recs, uint64_t n, uint32_t L,
                                                      sgc_table_view lib, sgc_table_view perm, int one_mm,
                                                      uint32_t *__restrict__ counts,
                                                      unsigned long long *__restrict__ matched) {
    uint64_t local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t span, status;
        load_record<REC16>(recs, i, L, span, status);
        const uint32_t g = sgc_assign<PACKED>(span, status, L, lib, perm, one_mm != 0);
        if (g != SGC_NONE) { atomicAdd(&counts[g], 1u); local++; }
    }
    // wave reduction of the matched tally, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(matched, (unsigned long long)local);
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

void sgc_launch_count_direct(hipStream_t st, const uint64_t *recs, uint64_t n, uint32_t L, bool rec16,
                             const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm, uint32_t *counts,
                             unsigned long long *matched) {
    if (n == 0) return;
    const unsigned block = 256, grid = grid_for(n, block, 256 * 8 * 4);
    const bool packed = lib.gid_bits != 0;
#define SGC_GO(P, R) \
    hipLaunchKernelGGL((k_count_direct<P, R>), dim3(grid), dim3(block), 0, st, recs, n, L, lib, perm, (int)one_mm, counts, matched)
    if (packed) { if (rec16) SGC_GO(true, true); else SGC_GO(true, false); }
    else        { if (rec16) SGC_GO(false, true); else SGC_GO(false, false); }
#undef SGC_GO
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
