// sgc_runs.h — workgroup-private partitioned runs: how the records a stage could not settle reach the next
// core pass (sgc_core.hip) without a counting pass and a scatter pass of their own.
//
// A *producer* workgroup (k_count_slices for its misses and the generic blocks, k_core<A> for what it forwards)
// ends with an epilogue: it knows how many of its leftover records fall into every partition of the consuming
// pass (an LDS histogram), takes ONE contiguous region of the shared record buffer with a single bump-allocator
// atomic, lays its records out partition by partition inside that region (LDS cursors), and publishes
// (count, start) per partition in two [P][W] matrices plus the per-partition totals.  A *consumer* workgroup of
// the core pass that works on partition p walks row p of the matrices: its input is the concatenation of W short
// segments, addressed through a prefix sum it keeps in LDS.  Nothing is moved twice, no kernel runs in between.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_device.h"
#include "sgc_format.h"

#define RUN_DROP 0xFFFFu
#define RUN_MAXP (1u << SGC_CORE_MAX_LOG2_P)

struct sgc_runs {
    uint64_t *recs;        // record buffer the regions are carved from (capacity >= all records of the pass)
    uint32_t *cnt;         // [P][W]: records of partition p left by producer w
    uint32_t *off;         // [P][W]: index in recs where they start
    uint32_t *tot;         // [P]: sum over w of cnt (zeroed before the producers start)
    uint32_t *cursor;      // bump allocator over recs (zeroed before the producers start)
    uint32_t W;            // number of producer workgroups (<= 1024): columns of the matrices
    uint32_t sub_bits;     // k_count_slices: log2 of the consuming pass's partitions per library slice when k_partition tagged
                           // them into the records (0..2), 0xFF = not tagged
    // partition function of the consuming pass: hash of the record's core bases, or RUN_DROP for a record whose
    // three windows are all dead (a read too short for the Centered window, src/counter.rs:158-166: it cannot match,
    // and all such records are identical, so they would pile up in one partition)
    uint32_t cs2, log2_p, sh, dead_all, cl;
    uint64_t cmask;
};

__device__ __forceinline__ uint32_t run_part(const sgc_runs &r, uint64_t rec) {
    if ((uint32_t)(rec >> r.sh) == r.dead_all) return RUN_DROP;
    return sgc_core_part(sgc_core_hash((uint32_t)((rec >> r.cs2) & r.cmask), r.cl), r.log2_p);
}

// Producer epilogue, part 1 (all threads of a 1024-lane workgroup; hn[p] = this workgroup's records per partition,
// complete and visible): takes the region, publishes the matrices' column w, leaves cur[p] = index in r.recs where
// the next record of partition p goes.  hn, cur: LDS arrays of RUN_MAXP entries; wtmp: LDS scratch of 17 words.  w may differ
// from thread to thread: thread t publishes row t.
__device__ __forceinline__ void run_reserve(const sgc_runs &r, uint32_t w, const uint32_t *hn, uint32_t *cur, uint32_t *wtmp,
                                            uint32_t *base_slot /* one LDS word */) {
    const uint32_t t = threadIdx.x, P = 1u << r.log2_p;
    const uint32_t c = t < P ? hn[t] : 0u;
    uint32_t M;
    const uint32_t lstart = wg_scan_1024(c, wtmp, &M);
    if (t == 0) *base_slot = M ? atomicAdd(r.cursor, M) : 0u;
    __syncthreads();
    const uint32_t base = *base_slot;
    if (t < P) {
        cur[t] = base + lstart;
        r.cnt[(size_t)t * r.W + w] = c;
        r.off[(size_t)t * r.W + w] = base + lstart;
        if (c) atomicAdd(&r.tot[t], c);
    }
    __syncthreads();
}

// part 2, per record (any subset of lanes): place it
__device__ __forceinline__ void run_place(const sgc_runs &r, uint32_t *cur, uint64_t rec) {
    const uint32_t p = run_part(r, rec);
    if (p != RUN_DROP) r.recs[atomicAdd(&cur[p], 1u)] = rec;
}
