// sgc_device.h — device-side helpers shared by the gfx950 kernels: table probes and the restatement of
// Counter::assign (reference src/counter.rs:96-140).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sgc_format.h"

// -DSGC_STAMPS=1 builds, dbg 1048576: every workgroup leaves one row — where it ran (XCD; HW_ID: wave / SIMD / CU / SE) and when
// (the constant 100 MHz clock) — in a buffer of its kernel, which sgc_*_timeline_dump() prints after the pass ("TL" lines:
// tools/wg_timeline.py draws the timeline of a kernel from them).  (A printf from the kernel itself stretches the lifetimes.)
#ifndef SGC_STAMPS
#define SGC_STAMPS 0
#endif
struct sgc_tl_row { unsigned long long begin, end, cycles, ph[4]; uint32_t xcc, hw, extra, used; };     // cycles: s_memtime ticks over the lifetime
#define SGC_TL_MAXWG 2048u
__device__ __forceinline__ uint32_t sgc_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20); }      // hwreg(HW_REG_XCC_ID, 0, 4)
__device__ __forceinline__ uint32_t sgc_hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }       // hwreg(HW_REG_HW_ID)
#if SGC_STAMPS
#define SGC_TIMELINE_BEGIN(dbg)                                                                                                  \
    const unsigned long long tl_begin_ = (SGC_STAMPS && ((dbg) & 1048576u)) ? __builtin_amdgcn_s_memrealtime() : 0ull,             \
                             tl_cyc_ = (SGC_STAMPS && ((dbg) & 1048576u)) ? __builtin_amdgcn_s_memtime() : 0ull
#define SGC_TIMELINE_END(dbg, buf, extra_) SGC_TIMELINE_END4(dbg, buf, extra_, 0, 0, 0, 0)
#define SGC_TIMELINE_END4(dbg, buf, extra_, p0, p1, p2, p3)                                                                     \
    if (SGC_STAMPS && ((dbg) & 1048576u) && threadIdx.x == 0 && blockIdx.x < SGC_TL_MAXWG) {                                      \
        sgc_tl_row row_;                                                                                                          \
        row_.ph[0] = (p0); row_.ph[1] = (p1); row_.ph[2] = (p2); row_.ph[3] = (p3);                                               \
        row_.begin = tl_begin_; row_.end = __builtin_amdgcn_s_memrealtime(); row_.xcc = sgc_xcc_id(); row_.hw = sgc_hw_id();     \
        row_.extra = (uint32_t)(extra_); row_.used = 1; row_.cycles = __builtin_amdgcn_s_memtime() - tl_cyc_;                                                                           \
        buf[blockIdx.x] = row_;                                                                                                   \
    }
#else
#define SGC_TIMELINE_BEGIN(dbg)
#define SGC_TIMELINE_END(dbg, buf, extra_)
#define SGC_TIMELINE_END4(dbg, buf, extra_, p0, p1, p2, p3)
#endif
// SGC_EXTRA_LDS_K1 / _K2 / _CORE (environment, -DSGC_STAMPS=1 builds only): dynamic LDS added to the launches of that kernel, to
// hold it to one workgroup per CU — the occupancy experiments of DESIGN.md
static inline unsigned sgc_extra_lds(const char *which) {
#if SGC_STAMPS
    char name[64];
    snprintf(name, sizeof name, "SGC_EXTRA_LDS_%s", which);
    const char *v = getenv(name);
    return v ? (unsigned)atoi(v) : 0u;
#else
    (void)which;
    return 0u;
#endif
}
// host side of a .hip file that owns such buffers
#define SGC_TIMELINE_DUMP(buf, name)                                                                                             \
    do {                                                                                                                          \
        static sgc_tl_row h_[SGC_TL_MAXWG];                                                                                       \
        (void)hipDeviceSynchronize();                                                                                             \
        if (hipMemcpyFromSymbol(h_, HIP_SYMBOL(buf), sizeof h_) != hipSuccess) break;                                             \
        for (uint32_t i_ = 0; i_ < SGC_TL_MAXWG; i_++)                                                                            \
            if (h_[i_].used) printf("TL %s wg %u xcc %u hw %x begin %llu end %llu extra %u cycles %llu ph %llu %llu %llu %llu\n", name, i_, h_[i_].xcc, h_[i_].hw, h_[i_].begin, h_[i_].end, h_[i_].extra, h_[i_].cycles, h_[i_].ph[0], h_[i_].ph[1], h_[i_].ph[2], h_[i_].ph[3]); \
        memset(h_, 0, sizeof h_);                                                                                                 \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(buf), h_, sizeof h_);                                                                  \
    } while (0)

// ------------------------------------------------------------------------------------------------
// bucketised open addressing: one 16-byte load reads both slots of a bucket
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ ulonglong2 load_bucket(const sgc_table_view &t, uint32_t b) {
    return reinterpret_cast<const ulonglong2 *>(t.slots)[b];
}

// 0 = keep probing, 1 = resolved (g holds the guide id or SGC_NONE); packed layout
__device__ __forceinline__ bool bucket_resolve(const ulonglong2 &v, uint64_t key, uint32_t gid_bits, uint32_t &g) {
    const uint64_t gmask = (1ull << gid_bits) - 1ull;
    if ((v.x >> gid_bits) == key && v.x != SGC_EMPTY) { g = (uint32_t)(v.x & gmask); return true; }
    if (v.x == SGC_EMPTY) { g = SGC_NONE; return true; }
    if ((v.y >> gid_bits) == key && v.y != SGC_EMPTY) { g = (uint32_t)(v.y & gmask); return true; }
    if (v.y == SGC_EMPTY) { g = SGC_NONE; return true; }
    return false;
}

template <bool PACKED>
__device__ __forceinline__ uint32_t table_find(const sgc_table_view &t, uint64_t key) {
    uint32_t b = sgc_home_bucket_ex(key, t.log2_slots, t.log2_slice, t.core_cl);
    for (;;) {
        const ulonglong2 v = load_bucket(t, b);
        if (PACKED) {
            uint32_t g;
            if (bucket_resolve(v, key, t.gid_bits, g)) return g;
        } else {
            if (v.x == SGC_EMPTY) return SGC_NONE;
            if (v.x == key) return t.vals[2 * b];
            if (v.y == SGC_EMPTY) return SGC_NONE;
            if (v.y == key) return t.vals[2 * b + 1];
        }
        b = sgc_next_bucket(b, t.log2_slice);
    }
}

// continue a probe whose home bucket `v` (index b) has already been loaded (packed layout)
__device__ __forceinline__ uint32_t finish_find(const sgc_table_view &t, uint64_t key, uint32_t b, ulonglong2 v) {
    for (;;) {
        uint32_t g;
        if (bucket_resolve(v, key, t.gid_bits, g)) return g;
        b = sgc_next_bucket(b, t.log2_slice);
        v = load_bucket(t, b);
    }
}
__device__ __forceinline__ uint32_t bucket_of(const sgc_table_view &t, uint64_t key) {
    return sgc_home_bucket_ex(key, t.log2_slots, t.log2_slice, t.core_cl);
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
// workgroup helpers shared by the partitioned kernels
// ------------------------------------------------------------------------------------------------
// exclusive scan of one value per thread over a 1024-thread workgroup; returns the prefix, *total = sum
__device__ __forceinline__ uint32_t wg_scan_1024(uint32_t v, uint32_t *wsum /*[17] in LDS*/, uint32_t *total) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += x;
    }
    __syncthreads();                       // wsum may still be read from a previous call
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < 16; w++) { const uint32_t x = wsum[w]; wsum[w] = run; run += x; }
        wsum[16] = run;
    }
    __syncthreads();
    *total = wsum[16];
    return wsum[wave] + incl - v;
}

// last u in [0, ns) with off[u] <= d (entries with nothing in them share their offset with the next one);
// branch-free with a fixed trip count
template <uint32_t LOG2_MAX>
__device__ __forceinline__ uint32_t find_extent(const uint32_t *off, uint32_t ns, uint32_t d) {
    uint32_t lo = 0;                     // off[0] == 0 <= d
#pragma unroll
    for (uint32_t step = 1u << (LOG2_MAX - 1); step; step >>= 1) {
        const uint32_t idx = lo + step;
        const uint32_t v = off[idx < ns ? idx : 0];
        lo = (idx < ns && v <= d) ? idx : lo;
    }
    return lo;
}

