// sgc_api.cpp — the C ABI declared in include/sgcount_hip.h (libsgcount_hip.so).
//
// Owns device memory, the stream, the library tables and per-sample count vectors.  Every entry point
// returns an error code; nothing throws across the boundary.  There is deliberately NO CPU fallback for
// the count path: without a HIP device sgc_init() fails and the caller must surface that.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sgcount_hip.h"
#include "sgc_bytes.h"
#include "sgc_format.h"
#include "sgc_kernels.h"
#include "sgc_runs.h"
#include "sgc_tables.h"

#if defined(SGC_STAMPS) && SGC_STAMPS
#define SGC_STAMPS_BUILD 1
#else
#define SGC_STAMPS_BUILD 0
#endif

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

enum { T_LOOKUP = 0, T_HIST = 1, T_PACK = 2, T_PART = 3, T_MISS = 4, T_H2D = 5, T_KINDS = 6 };

// Device memory of one library's tables, shared by a ctx and its clones (read-only once built)
struct table_owner {
    int device = 0;
    std::vector<void *> ptrs;
    ~table_owner() { (void)hipSetDevice(device); for (void *p : ptrs) (void)hipFree(p); }
};

struct sgc_ctx {
    int device = 0;
    std::shared_ptr<table_owner> tables;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side_stream = nullptr;          // k_generic runs beside k_resolve_miss (fork/join with events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // library
    bool has_lib = false, one_mm = false, rec16 = false;
    uint32_t n = 0, L = 0;
    uint64_t *d_lib_slots = nullptr, *d_perm_slots = nullptr, *d_lib_cuckoo = nullptr;
    bool dense = true;                 // k_count_slices writes its misses as dense runs (no per-group barrier) instead of in place
    bool direct = true;                // ... one run per partition of core pass A, consumed where it lies (needs dense + tag_sub)
    // Placement trials (DESIGN.md §6 "the two regimes of the pass"): where the pool falls in memory decides whether K1 and K2
    // run in their fast or their slow regime (~10 % apart), so the first large pass tries a few placements and keeps the fastest
    bool verbose = false;              // diagnostics on stderr (option "verbose")
    size_t vmm_chunk = 0;              // experiment: physical chunk size of the block pool (see vmm_alloc above); 0 = plain hipMalloc
    uint64_t vmm_shuffle = 0;          // map the pool's chunks in a seeded random order (0 = as created)
    bool vmm_runs = false;             // ... the run buffers of the core passes too (they are read more than written: plain memory reads faster)
    int place_trials = 1;              // allocations tried (1 = take what hipMalloc gives: the default; a host that re-counts resident
                                       // samples of >= 32M records may opt in — the search costs 10-30 ms and holds extra pools meanwhile)
    uint64_t place_info[4] = {0, 0, 0, 0};   // last search: candidates tried, peak transient bytes, duration in us, index of the one kept
    void *placed_pool = nullptr;       // the pool the trials chose (they run again if it was re-allocated since)
    bool six_byte = true;              // ... and the slice blocks hold six-byte records (needs direct, L <= 21)
    bool wide = true;                  // k_count_slices reads its five-byte blocks 256 consecutive records per wave (16-byte loads)
    bool five_byte = true;             // ... or five-byte records (needs direct and enough slices: 2 (L + 2) - slice bits <= 40)
    bool tag_sub = true;               // K1 tags the pass-A partition inside the slice, K2 counts misses by it (no histogram sweep)
    bool use_cuckoo = true;            // k_count_slices probes the two-choice image of the slices (no chain loop)
    uint32_t *d_lib_vals = nullptr, *d_perm_vals = nullptr;
    sgc_table_view v_lib{}, v_perm{};
    uint64_t *d_bloom_lib = nullptr, *d_bloom_perm = nullptr;
    sgc_bloom_view b_lib{}, b_perm{};
    uint64_t perm_entries = 0;
    // core indexes + ambiguity masks of the in-LDS single-mismatch resolver (variant 4; sgc_core.hip)
    bool has_core = false;
    uint64_t *d_core_ents[2] = {nullptr, nullptr}, *d_amb = nullptr;
    uint32_t *d_core_gids[2] = {nullptr, nullptr};
    uint16_t *d_core_starts[2] = {nullptr, nullptr};
    uint32_t *d_core_filt = nullptr;
    sgc_core_view v_core[2] = {};
    // generic byte-string path (sgc_bytes.h): libraries the 2-bit records cannot represent
    bool bytes_mode = false;
    // hybrid library (sgc_bytes.hip): the packed pass over the n_packed ACGT guides + the byte-string chain over all n guides for the
    // reads a guide with other bytes could influence; d_gid_map: packed guide number -> library index; b_shadow: Bloom filter of the
    // shadow keys (the ACGT windows one substitution away from a guide with exactly one byte outside ACGT)
    bool hybrid = false, allow_hybrid = true;
    bool host_routes = false;          // the host promises to push packed records only for reads no guide outside ACGT can influence
    uint32_t n_packed = 0;
    uint32_t *d_gid_map = nullptr;
    uint64_t *d_bloom_shadow = nullptr;
    sgc_bloom_view b_shadow{};
    void *d_flags = nullptr; size_t flags_cap = 0;      // per-read route flags of a hybrid push
    void *d_lines = nullptr; size_t lines_cap = 0;      // ... and the (start, end) of the reads' sequence lines
    bool force_bytes = false;          // serve every library through the byte-string path (next sgc_set_library; tests)
    uint8_t *d_bytes_seqs = nullptr;
    uint64_t *d_bytes_tags[2] = {nullptr, nullptr};
    uint32_t *d_bytes_vals[2] = {nullptr, nullptr};
    uint32_t *d_bytes_pl = nullptr;
    sgc_bytes_view v_bytes{};
    void *d_cbuf = nullptr; size_t cbuf_cap = 0;        // two record buffers of the core passes
    void *d_csmall = nullptr; size_t csmall_cap = 0;    // their histograms / partition starts / extents
    uint32_t *d_slice_tot = nullptr;                    // 2 x SGC_SLICE_TOT words: blocks per slice of this pass | zeroed for the next (balanced shares)
    int tot_parity = 0, balanced = 1;
    // scratch (grown on demand, stream-ordered reuse)
    void *d_stage = nullptr; size_t stage_cap = 0;      // host -> device staging of pushed buffers
    // FASTQ text pushed from host memory: uploads run on their own stream into two alternating device buffers, so that
    // the upload of part k+1 overlaps the ingest and count kernels of part k
    hipStream_t copy_stream = nullptr;
    void *d_text[2] = {nullptr, nullptr}; size_t text_cap[2] = {0, 0};
    hipEvent_t ev_use[2] = {nullptr, nullptr};          // the ingest kernels that read d_text[i] are done
    bool use_recorded[2] = {false, false};
    static constexpr int UP_RING = 8;
    hipEvent_t ev_up[UP_RING] = {};                     // upload k is complete: ev_up[k % UP_RING]
    uint64_t n_up = 0;                                  // uploads issued so far
    void *d_aux = nullptr; size_t aux_cap = 0;          // offsets / secondary staging
    uint64_t *d_recs = nullptr; size_t recs_cap = 0;    // records produced by the on-device packers
    void *d_gids = nullptr; size_t gids_cap = 0;        // per-read guide ids between the lookup and histogram kernels
    void *d_pool = nullptr; size_t pool_cap = 0;        // partitioned path: record blocks
    void *d_desc = nullptr; size_t desc_cap = 0;        // partitioned path: block descriptors
    // options
    int variant = 4;            // count path variant: 1 the generic kernels (gid array + LDS histogram), 3 partitioned +
                                // probing miss resolver, 4 partitioned + in-LDS core resolver (shipped)
    uint32_t dbg = 0;           // timing-only ablation flags / phase stamps (only in -DSGC_ABLATE=1 / -DSGC_STAMPS=1 builds; results are wrong when an ablation is on)
    uint32_t k1_wgs = 512;      // workgroups of the partition kernel: two per CU (more leave more half-empty blocks open, fewer expose its phases)
    uint64_t max_chunk = 1ull << 27;   // records per internal pass (bounds the scratch buffers)
    uint64_t batch_records = 1ull << 24;   // sgc_sample_push_packed_async: records per device-side batch (one count pass each)
    bool rest_filter = true;           // core pass A settles "no parent inside the core" with the rest filter (sgc_format.h)
    bool align_slices = true;          // build the library table with slices that follow the core hash (next sgc_set_library)
    int slice_log2 = 0;                // 0: sgc_choose_log2_slice(n); 12 / 13: slots per library slice of the next sgc_set_library (tuning)
    bool host_build = false;           // build the single-mismatch table on the host (sgc_tables.cpp) instead of the GPU
    uint32_t perm_bloom_bits = 8;      // Bloom bits per child of the single-mismatch filter
    // timing
    bool timing = false;
    struct span_ev { hipEvent_t a, b; int kind; bool owns_a; };
    std::vector<span_ev> pending;
    std::vector<hipEvent_t> free_events;
    sgc_timing acc{};
};

struct sgc_sample {
    sgc_ctx *ctx = nullptr;
    int reverse = 0; uint32_t offset = 0; int recursion = 1;
    uint32_t *d_c32 = nullptr;
    unsigned long long *d_c64 = nullptr;
    unsigned long long *d_matched = nullptr;
    unsigned long long *d_err = nullptr;   // [0] = ~0 - (first line whose marker byte is wrong), 0 = none; [1] = flags (FASTQ ingest)
    bool fastq_pushed = false;
    uint64_t total = 0;
    uint64_t since_fold = 0;     // reads counted into d_c32 since the last fold (u32 overflow guard)
    size_t state_bytes = 0;
    // sgc_sample_push_packed_async: two device batch buffers, filled by the upload stream, counted when full
    uint32_t *d_c32p = nullptr;        // hybrid library: u32 counts of the packed pass, numbered over the ACGT guides
    void *d_acc[2] = {nullptr, nullptr};
    hipEvent_t ev_acc_use[2] = {nullptr, nullptr};   // the count pass that read d_acc[i] is done
    bool acc_used[2] = {false, false};
    uint64_t acc_cap = 0, acc_fill = 0;              // records
    int acc_cur = 0;
    hipEvent_t acc_last_up = nullptr;                // the most recent upload into the current batch
};

// ---- helpers -------------------------------------------------------------------------------------

// Experiment hook (option "vmm_chunk_mb", default 0 = plain hipMalloc): the block pool built with the virtual-memory API — a
// reserved address range backed by physical chunks of `chunk` bytes (hipMemCreate / hipMemMap), optionally mapped in a shuffled
// order.  Why it exists: WHERE the pool lies in physical memory decides whether k_partition / k_count_slices run in their fast
// or their slow regime.  What round 3 learnt with it (tools/ubench/contig_bw.hip, tools/regime_sweep.py): a
// physically contiguous pool (hipDeviceMallocContiguous) is the SLOWEST placement (K1 0.39-0.42 ms against 0.30-0.35); plain
// streaming kernels write chunked memory faster than hipMalloc'ed memory (6.4 against 4.4 TB/s) and read it slower (6.3
// against 7.1); but for the pass neither the chunk size nor the mapping order removes the spread — the k-th allocation of a
// process lands in the same regime in the next process, whichever API made it: the regime is a property of the physical
// region, not of the mapping.
struct vmm_rec { size_t bytes; std::vector<hipMemGenericAllocationHandle_t> handles; };
static std::mutex g_vmm_mu;
static std::map<void *, vmm_rec> g_vmm;            // address -> its chunks (all contexts of the process)

static void dev_free(void *p) {
    if (!p) return;
    vmm_rec r;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm.find(p);
        if (it == g_vmm.end()) { (void)hipFree(p); return; }
        r = std::move(it->second);
        g_vmm.erase(it);
    }
    (void)hipMemUnmap(p, r.bytes);
    for (auto h : r.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(p, r.bytes);
}

static bool vmm_alloc(int device, void **out, size_t bytes, size_t chunk, uint64_t shuffle = 0) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { (void)hipGetLastError(); return false; }
    chunk = (std::max(chunk, gran) + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk, n = total / chunk;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    vmm_rec r; r.bytes = total;
    bool ok = true;
    size_t mapped = 0;
    for (size_t i = 0; i < n && ok; i++) {
        hipMemGenericAllocationHandle_t h;
        ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess;
        if (ok) r.handles.push_back(h);
    }
    if (ok && shuffle) {            // the chunks in a seeded random order (xorshift Fisher-Yates)
        uint64_t x = shuffle * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = n; i > 1; i--) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(r.handles[i - 1], r.handles[x % i]);
        }
    }
    for (size_t i = 0; i < n && ok; i++) {
        ok = hipMemMap((char *)va + i * chunk, chunk, 0, r.handles[i], 0) == hipSuccess;
        if (ok) mapped = (i + 1) * chunk;
    }
    hipMemAccessDesc ad = {};
    ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    ok = ok && hipMemSetAccess(va, total, &ad, 1) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (mapped) (void)hipMemUnmap(va, mapped);
        for (auto h : r.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, total);
        return false;
    }
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_vmm[va] = std::move(r);
    }
    *out = va;
    return true;
}

// vmm_chunk != 0: build the buffer from physical chunks of that size (falls back to hipMalloc if the device refuses)
static int ensure(void **p, size_t *cap, size_t need, int device = 0, size_t vmm_chunk = 0, uint64_t shuffle = 0) {
    if (need <= *cap) return SGC_OK;
    const size_t want = std::max(need, *cap * 2);
    if (*p) { dev_free(*p); *p = nullptr; *cap = 0; }
    if (vmm_chunk == 1024) {         // experiment ("vmm_chunk_kb" = 1): physically contiguous
        if (hipExtMallocWithFlags(p, want, hipDeviceMallocContiguous) == hipSuccess) { *cap = want; return SGC_OK; }
        (void)hipGetLastError(); *p = nullptr;
    } else if (vmm_chunk && vmm_alloc(device, p, want, vmm_chunk, shuffle)) { *cap = want; return SGC_OK; }
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return SGC_OK;
}

static hipEvent_t ev_get(sgc_ctx *c) {
    if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    // timing markers only: no system-scope fence (a default event makes the L2 write back its dirty lines)
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
    return e;
}

static void timing_drain(sgc_ctx *c) {
    for (auto &s : c->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            if (s.kind == T_LOOKUP) c->acc.lookup_ms += ms;
            else if (s.kind == T_HIST) c->acc.hist_ms += ms;
            else if (s.kind == T_PART) c->acc.part_ms += ms;
            else if (s.kind == T_MISS) c->acc.miss_ms += ms;
            else if (s.kind == T_H2D) c->acc.h2d_ms += ms;
            else c->acc.pack_ms += ms;
            if (s.kind != T_H2D) c->acc.launches++;
        }
        if (s.owns_a) c->free_events.push_back(s.a);
        c->free_events.push_back(s.b);
    }
    c->pending.clear();
}

// One timed span on the ctx stream.  chain = true: the span starts where the previous one ended (its end event
// is reused), which is only right when nothing was enqueued in between — it saves one event packet (~4 us of
// stream time) between back-to-back kernels.
struct timed {
    sgc_ctx *c; int kind; hipEvent_t a = nullptr, b = nullptr; bool owns_a = true; hipStream_t st;
    timed(sgc_ctx *c_, int k, bool chain = false, hipStream_t other = nullptr) : c(c_), kind(k), st(other ? other : c_->stream) {
        if (!c->timing) return;
        if (chain && !other && !c->pending.empty() && c->pending.back().kind != T_H2D) { a = c->pending.back().b; owns_a = false; }
        else { a = ev_get(c); if (a) hipEventRecord(a, st); }
        b = ev_get(c);
    }
    ~timed() {
        if (!c->timing || !a || !b) return;
        hipEventRecord(b, st);
        c->pending.push_back({a, b, kind, owns_a});
        if (c->pending.size() >= 256) timing_drain(c);
    }
};

// counts64 += counts32 (and, hybrid: the packed pass's counts through the guide map)
static void fold_counts(sgc_sample *s) {
    sgc_ctx *c = s->ctx;
    sgc_launch_fold(c->stream, s->d_c32, s->d_c64, c->n);
    if (c->hybrid) sgc_launch_fold_map(c->stream, s->d_c32p, c->d_gid_map, s->d_c64, c->n_packed);
    s->since_fold = 0;
}

static int count_records(sgc_sample *s, const uint64_t *d_recs, uint64_t n, bool add_total = true) {
    sgc_ctx *c = s->ctx;
    uint32_t *const c32 = c->hybrid ? s->d_c32p : s->d_c32;          // where the packed pass counts
    const uint32_t n_g = c->hybrid ? c->n_packed : c->n;              // ... and how many guides it numbers
    // u32 device counters: fold into the u64 vector before any counter could wrap
    if (s->since_fold + n > 0xFFFFFFF0ull) {
        fold_counts(s);
    }
    uint64_t done = 0;
    while (done < n) {
        const uint64_t chunk = std::min<uint64_t>(n - done, c->max_chunk);
        const uint64_t *p = d_recs + done * (c->rec16 ? 2 : 1);
        if (c->variant >= 3 && sgc_part_supported(c->v_lib, c->rec16)) {
            sgc_part_geometry g;
            sgc_part_plan(chunk, c->v_lib, c->k1_wgs, &g);
            int rc = ensure(&c->d_pool, &c->pool_cap, g.pool_bytes, c->device, c->vmm_chunk, c->vmm_shuffle);
            if (rc) return rc;
            rc = ensure(&c->d_desc, &c->desc_cap, g.desc_bytes);
            if (rc) return rc;
            uint64_t *pool = (uint64_t *)c->d_pool;
            uint32_t *desc = (uint32_t *)c->d_desc;
            // with core-hashed slices whose count divides core pass A's partitions by 1, 2 or 4, K1 tags every clean record with its
            // partition inside the slice and K2 counts its misses by it on the fly
            const bool core_path = c->variant >= 4 && c->has_core;
            const int sub = (int)c->v_core[0].log2_p - ((int)c->v_lib.log2_slots - (int)c->v_lib.log2_slice);
            const bool tag_sub = core_path && c->tag_sub && c->v_lib.core_cl == c->v_core[0].cl && c->v_lib.log2_slice < c->v_lib.log2_slots &&
                                 sub >= 0 && sub <= 2;
            // direct runs: the misses go straight to per-partition runs inside one allocation with pass A's other input
            // (runs region | miss runs | forward buffer); the run matrices get one more column per workgroup of a slice.
            // With them nothing but the probe loop reads the slice blocks, which then hold six-byte records if they fit.
            // (the run matrices have at most 1024 columns: one per workgroup for its share of the generic blocks + the direct runs' —
            // balanced shares: any workgroup may meet any slice; static shares: the workgroups of one slice)
            const bool direct = core_path && c->dense && c->direct && tag_sub &&
                                sgc_part_k2_grid(g) + sgc_part_k2_direct_cols(g, c->balanced != 0) <= 1024u;
            const bool bal = direct && c->balanced;
            if (bal && !c->d_slice_tot) {
                HIP_TRY(hipMalloc((void **)&c->d_slice_tot, 2 * SGC_SLICE_TOT * 4));
                HIP_TRY(hipMemsetAsync(c->d_slice_tot, 0, 2 * SGC_SLICE_TOT * 4, c->stream));
                c->tot_parity = 0;
            }
            uint32_t *const tot = bal ? c->d_slice_tot + SGC_SLICE_TOT * c->tot_parity : nullptr;
            uint32_t *const tot_next = bal ? c->d_slice_tot + SGC_SLICE_TOT * (c->tot_parity ^ 1) : nullptr;
            // what the slice blocks hold: 2 = five-byte records (the slice index is a prefix of the mixed core value, so a record keeps
            // 2 (L + 2) - slice bits <= 40 bits), 1 = six-byte records (span + sub-partition <= 48 bits), 0 = whole 8-byte records
            const uint32_t slice_bits = c->v_lib.log2_slots - c->v_lib.log2_slice;
            const int six = !direct ? 0 :
                            (c->five_byte && 2u * (c->L + 2u) - slice_bits <= 40u && slice_bits + (uint32_t)sub <= 2u * c->v_lib.core_cl) ? 2 :
                            (c->six_byte && 2u * (c->L + 2u) + 2u <= 48u) ? 1 : 0;
            // Every allocation of the pass BEFORE its first launch: k_partition adds its block counts into the per-slice totals, and a
            // pass that gave up after it (out of memory for the run buffers) would leave them there for the next pass to add to.
            sgc_core_geometry cg{};
            size_t mrun_bytes = 0;
            if (core_path) {
                sgc_core_plan(chunk, c->v_core[0], c->v_core[1], sgc_part_k2_grid(g) + (direct ? sgc_part_k2_direct_cols(g, bal) : 0u), &cg);
                mrun_bytes = c->dense ? (size_t)g.pool_bytes << (direct ? (uint32_t)sub : 0u) : 0;
                rc = ensure(&c->d_cbuf, &c->cbuf_cap, (size_t)(cg.runs_a_bytes + mrun_bytes + cg.fwd_bytes), c->device, c->vmm_runs ? c->vmm_chunk : 0);
                if (rc) return rc;
                rc = ensure(&c->d_csmall, &c->csmall_cap, cg.small_bytes);
                if (rc) return rc;
            }
            { timed t(c, T_PART); sgc_launch_part_k1(c->stream, s->d_err + 1, p, chunk, c->L, c->v_lib, tag_sub ? (uint32_t)sub : 0u, g, pool, desc, six, c->dbg, tot); }
            if (core_path) {
                // everything the slice probe does not settle (its misses + the generic partition) is resolved in LDS by the
                // two core passes; k_count_slices itself lays those records out as pass A's runs
                if (direct && c->place_trials > 1 && chunk >= (1ull << 25) && c->placed_pool != c->d_pool) {
                    // Placement trials.  K1 and K2 speed up and slow down TOGETHER from candidate to candidate (K1 0.30 ... 0.36 ms,
                    // K2 0.24 ... 0.31), and K1 touches nothing of ours but the pool: it is the pool's place in memory that decides.
                    // So: up to place_trials pool allocations — all held during the search, so that each falls somewhere else —,
                    // K1 of this very chunk twice on each, the second run timed; the fastest is kept.
                    struct cand { void *pool; float ms; };
                    std::vector<cand> cands;
                    cands.push_back({c->d_pool, 0.f});
                    const auto search_t0 = std::chrono::steady_clock::now();
                    // the candidates are all held at once: never more than a quarter of the memory that is free right now
                    size_t mem_free = 0, mem_total = 0;
                    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_free = 0; }
                    const int max_extra = (int)std::min<size_t>((size_t)c->place_trials - 1, (mem_free / 4) / std::max<size_t>(c->pool_cap, 1));
                    hipEvent_t e0 = nullptr, e1 = nullptr;
                    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
                    float worst = 0.f;
                    for (int k = 0; ok && k <= max_extra; k++) {
                        if (k) {
                            void *P = nullptr;
                            size_t cap2 = 0;
                            if (ensure(&P, &cap2, c->pool_cap, c->device, c->vmm_chunk) != SGC_OK) { (void)hipGetLastError(); break; }    // out of memory: settle for what there is
                            cands.push_back({P, 0.f});
                        }
                        cand &cd = cands.back();
                        for (int rep = 0; rep < 2; rep++) {
                            if (rep) ok = ok && hipEventRecord(e0, c->stream) == hipSuccess;
                            sgc_launch_part_k1(c->stream, s->d_err + 1, p, chunk, c->L, c->v_lib, (uint32_t)sub, g, (uint64_t *)cd.pool, desc, six, c->dbg, nullptr);
                        }
                        ok = ok && hipEventRecord(e1, c->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                             hipEventElapsedTime(&cd.ms, e0, e1) == hipSuccess;
                        if (c->verbose) fprintf(stderr, "placement trial %d: pool %p: K1 %.3f ms\n", k, cd.pool, cd.ms);
                        worst = std::max(worst, cd.ms);
                        // the regimes are >= 10 % apart: a candidate that far ahead of the slowest one seen is in the fast one
                        if (ok && k >= 3 && cd.ms < 0.89f * worst) break;
                    }
                    if (e0) hipEventDestroy(e0);
                    if (e1) hipEventDestroy(e1);
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    size_t best = 0;
                    for (size_t k = 1; k < cands.size(); k++)
                        if (ok && cands[k].ms > 0.f && cands[k].ms < cands[best].ms) best = k;
                    for (size_t k = 0; k < cands.size(); k++)
                        if (k != best) dev_free(cands[k].pool);
                    c->d_pool = cands[best].pool;
                    c->placed_pool = c->d_pool;
                    c->place_info[0] = cands.size(); c->place_info[1] = (uint64_t)(cands.size() - 1) * c->pool_cap;
                    c->place_info[2] = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - search_t0).count();
                    c->place_info[3] = best;
                    if (c->verbose) fprintf(stderr, "placement trials: %zu candidates (%.2f GB held meanwhile), kept #%zu, %.1f ms\n", cands.size(),
                                            (double)c->place_info[1] / 1e9, best, (double)c->place_info[2] / 1e3);
                    pool = (uint64_t *)c->d_pool;
                    HIP_TRY(hipGetLastError());
                    // the real pass starts over on the chosen pool
                    if (tot) HIP_TRY(hipMemsetAsync(tot, 0, SGC_SLICE_TOT * 4, c->stream));
                    { timed t(c, T_PART); sgc_launch_part_k1(c->stream, s->d_err + 1, p, chunk, c->L, c->v_lib, tag_sub ? (uint32_t)sub : 0u, g, pool, desc, six, c->dbg, tot); }
                }
                uint64_t *buf0 = (uint64_t *)c->d_cbuf, *mrun = c->dense ? (uint64_t *)((char *)c->d_cbuf + cg.runs_a_bytes) : nullptr;
                uint64_t *buf1 = (uint64_t *)((char *)c->d_cbuf + cg.runs_a_bytes + mrun_bytes);
                void *zeroed = (char *)c->d_desc + g.desc_tail_off;
                sgc_runs ra = sgc_core_runs_a(cg, c->v_core[0], c->L, buf0, zeroed, c->d_csmall);
                if (tag_sub) ra.sub_bits = (uint32_t)sub;
                uint32_t *mcur = (uint32_t *)zeroed + 2 * RUN_MAXP + 2;        // behind totals A | totals B | region cursors (zeroed by K1)
                { timed t(c, T_LOOKUP, true); sgc_launch_part_k2(c->stream, c->L, c->v_lib, g, pool, desc, c32, s->d_matched, c->dbg, &ra,
                                                                 c->use_cuckoo ? c->d_lib_cuckoo : nullptr, mrun, mcur, direct, six, tot, tot_next, c->wide); }
                if (bal) c->tot_parity ^= 1;
                // timing: miss_ms = core pass A (+ its epilogue), hist_ms = core pass B
                if (!c->one_mm) {
                    timed t(c, T_MISS, true);
                    sgc_launch_core(c->stream, 2, c->L, c->v_lib, c->v_perm, c->v_core[0], c->v_core[1], c->d_amb, cg, buf0, buf1, pool,
                                    zeroed, c->d_csmall, c32, s->d_matched, c->dbg);
                } else {
                { timed t(c, T_MISS, true); sgc_launch_core(c->stream, 0, c->L, c->v_lib, c->v_perm, c->v_core[0], c->v_core[1], c->d_amb, cg, buf0, buf1, pool,
                                                            zeroed, c->d_csmall, c32, s->d_matched, c->dbg); }
                { timed t(c, T_HIST, true); sgc_launch_core(c->stream, 1, c->L, c->v_lib, c->v_perm, c->v_core[0], c->v_core[1], c->d_amb, cg, buf0, buf1, pool,
                                                            zeroed, c->d_csmall, c32, s->d_matched, c->dbg); }
                }
                HIP_TRY(hipGetLastError());
                done += chunk;
                s->since_fold += chunk;
                if (done < n) fold_counts(s);
                continue;
            }
            { timed t(c, T_LOOKUP, true); sgc_launch_part_k2(c->stream, c->L, c->v_lib, g, pool, desc, c32, s->d_matched, c->dbg, nullptr, c->use_cuckoo ? c->d_lib_cuckoo : nullptr, nullptr, nullptr, false, 0, nullptr, nullptr); }
            rc = ensure(&c->d_gids, &c->gids_cap, g.gids_bytes);                // one slot per pool record: every read may miss
            if (rc) return rc;
            rc = ensure(&c->d_aux, &c->aux_cap, ((size_t)g.n_segs + 1) * 4);
            if (rc) return rc;
            // k_generic (serial chains, no LDS) runs beside k_resolve_miss (LDS-heavy, one workgroup per CU):
            // fork to the side stream here, join before the histogram
            HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
            sgc_launch_part_generic(c->side_stream, c->L, c->v_lib, c->v_perm, c->one_mm, g, pool, desc, c32, s->d_matched);
            HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
            { timed t(c, T_MISS); sgc_launch_part_k3(c->stream, c->L, c->v_lib, c->v_perm, c->one_mm, c->b_lib, c->b_perm, g, pool, desc,
                                                     (uint32_t *)c->d_aux, (uint32_t *)c->d_gids, c->dbg); }
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            { timed t(c, T_HIST); sgc_launch_part_k4(c->stream, n_g, g, (const uint32_t *)c->d_gids,
                                                     (const uint32_t *)c->d_aux, c32, s->d_matched); }
        } else {
            // the generic kernels (variant 1): any record layout, any table layout — one gid per read, then an LDS histogram
            int rc = ensure(&c->d_gids, &c->gids_cap, (size_t)chunk * 4);
            if (rc) return rc;
            {
                timed t(c, T_LOOKUP);
                sgc_launch_lookup_gids(c->stream, p, chunk, c->L, c->rec16, c->v_lib, c->v_perm, c->one_mm,
                                       (uint32_t *)c->d_gids, s->d_matched);
            }
            {
                timed t(c, T_HIST);
                sgc_launch_hist_slices(c->stream, (const uint32_t *)c->d_gids, chunk, n_g, c32);
            }
        }
        HIP_TRY(hipGetLastError());
        done += chunk;
        s->since_fold += chunk;
        if (done < n) fold_counts(s);
    }
    if (add_total) s->total += n;
    return SGC_OK;
}

// the byte-string path (sgc_bytes.hip): read i = text[starts[i], ends[i])
// flags (hybrid library): only the reads flagged for this chain take part, and the reads were already counted into the total
static int count_bytes(sgc_sample *s, const uint8_t *d_text, const uint64_t *d_starts, const uint64_t *d_ends, uint64_t n, const uint8_t *flags = nullptr) {
    sgc_ctx *c = s->ctx;
    uint64_t done = 0;
    while (done < n) {
        const uint64_t chunk = std::min<uint64_t>(n - done, 1ull << 30);
        if (s->since_fold + chunk > 0xFFFFFFF0ull) fold_counts(s);
        { timed t(c, T_LOOKUP); sgc_launch_bytes_count(c->stream, c->v_bytes, d_text, d_starts + done, d_ends + done, chunk, s->reverse, s->offset,
                                                       s->recursion, c->one_mm, s->d_c32, s->d_matched, flags ? flags + done : nullptr); }
        HIP_TRY(hipGetLastError());
        done += chunk; s->since_fold += chunk;
    }
    if (!flags) s->total += n;
    return SGC_OK;
}

// hybrid library: records + the reads' bytes -> route flags (flagged records die), the packed pass, the byte-string chain of the flagged
static int count_hybrid(sgc_sample *s, uint64_t *d_recs, const uint8_t *d_text, const uint64_t *d_starts, const uint64_t *d_ends, uint64_t n) {
    sgc_ctx *c = s->ctx;
    int rc = ensure(&c->d_flags, &c->flags_cap, n ? n : 1);
    if (rc) return rc;
    { timed t(c, T_PACK); sgc_launch_bytes_route(c->stream, d_text, d_starts, d_ends, n, c->L, c->rec16, s->reverse, s->offset, s->recursion, d_recs,
                                                 c->b_shadow, (uint8_t *)c->d_flags); }
    HIP_TRY(hipGetLastError());
    rc = count_records(s, d_recs, n);
    if (rc) return rc;
    return count_bytes(s, d_text, d_starts, d_ends, n, (const uint8_t *)c->d_flags);
}

// ---- ABI -------------------------------------------------------------------------------------------

extern "C" {

const char *sgc_last_error(void) { return g_err.c_str(); }
const char *sgc_version(void) { return SGC_CHECK ? "sgcount_hip 0.1.0 (gfx950, bounds-checked build)" : "sgcount_hip 0.1.0 (gfx950)"; }

uint32_t sgc_record_bytes(uint32_t L) {
    if (L == 0 || L > SGC_MAXL) return 0;
    return L <= SGC_REC8_MAXL ? 8 : 16;
}

int sgc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sgc_init(int device, sgc_ctx **out) {
    if (!out) return fail(SGC_E_ARG, "sgc_init: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(SGC_E_HIP, std::string("sgc_init: no HIP device available (") + hipGetErrorString(e) +
                                   "); the count path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(SGC_E_ARG, "sgc_init: device index out of range");
    HIP_TRY(hipSetDevice(device));
    sgc_ctx *c = new (std::nothrow) sgc_ctx();
    if (!c) return fail(SGC_E_OOM, "sgc_init: out of host memory");
    c->device = device;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(SGC_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    c->stream = c->own_stream;
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        sgc_free(c);
        return fail(SGC_E_HIP, "sgc_init: cannot create the side stream");
    }
    bool ok = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++) ok = hipEventCreateWithFlags(&c->ev_use[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < sgc_ctx::UP_RING && ok; i++) ok = hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { sgc_free(c); return fail(SGC_E_HIP, "sgc_init: cannot create the upload stream"); }
    *out = c;
    return SGC_OK;
}

// every device allocation that belongs to the library tables of a ctx
static std::vector<void *> table_ptrs(const sgc_ctx *c) {
    std::vector<void *> v = {c->d_lib_slots, c->d_lib_cuckoo, c->d_perm_slots, c->d_lib_vals, c->d_perm_vals, c->d_bloom_lib, c->d_bloom_perm,
                             c->d_amb, c->d_core_filt, c->d_bytes_seqs, c->d_bytes_pl, c->d_gid_map, c->d_bloom_shadow};
    for (int k = 0; k < 2; k++) {
        v.push_back(c->d_core_ents[k]); v.push_back(c->d_core_gids[k]); v.push_back(c->d_core_starts[k]);
        v.push_back(c->d_bytes_tags[k]); v.push_back(c->d_bytes_vals[k]);
    }
    v.erase(std::remove(v.begin(), v.end(), nullptr), v.end());
    return v;
}

static void free_tables(sgc_ctx *c) {
    // a finished library belongs to its owner token, shared with the clones of the ctx (sgc_ctx_clone): the last one frees it;
    // a half-built one (error paths of sgc_set_library) is freed here
    if (c->tables) c->tables.reset();
    else for (void *p : table_ptrs(c)) hipFree(p);
    c->d_lib_cuckoo = nullptr;
    c->d_bloom_lib = c->d_bloom_perm = nullptr;
    for (int k = 0; k < 2; k++) {
        c->d_core_ents[k] = nullptr; c->d_core_gids[k] = nullptr; c->d_core_starts[k] = nullptr; c->v_core[k] = sgc_core_view{};
    }
    c->d_amb = nullptr; c->d_core_filt = nullptr; c->has_core = false;
    for (int k = 0; k < 2; k++) { c->d_bytes_tags[k] = nullptr; c->d_bytes_vals[k] = nullptr; }
    c->d_bytes_seqs = nullptr; c->d_bytes_pl = nullptr; c->bytes_mode = false; c->v_bytes = sgc_bytes_view{};
    c->b_lib = sgc_bloom_view{}; c->b_perm = sgc_bloom_view{};
    c->d_lib_slots = c->d_perm_slots = nullptr;
    c->d_lib_vals = c->d_perm_vals = nullptr;
    c->d_gid_map = nullptr; c->d_bloom_shadow = nullptr; c->b_shadow = sgc_bloom_view{}; c->hybrid = false; c->n_packed = 0;
    c->has_lib = false;
}

// the library of the ctx is complete: hand its allocations to an owner token
static void adopt_tables(sgc_ctx *c) {
    auto t = std::make_shared<table_owner>();
    t->device = c->device; t->ptrs = table_ptrs(c);
    c->tables = std::move(t);
}

void sgc_free(sgc_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    timing_drain(c);
    for (auto e : c->free_events) hipEventDestroy(e);
    free_tables(c);
    dev_free(c->d_flags); dev_free(c->d_lines);
    dev_free(c->d_stage);
    dev_free(c->d_aux);
    dev_free(c->d_recs);
    dev_free(c->d_gids);
    dev_free(c->d_pool);
    dev_free(c->d_desc);
    dev_free(c->d_cbuf);
    dev_free(c->d_csmall);
    if (c->d_slice_tot) (void)hipFree(c->d_slice_tot);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 2; i++) { dev_free(c->d_text[i]); if (c->ev_use[i]) hipEventDestroy(c->ev_use[i]); }
    for (int i = 0; i < sgc_ctx::UP_RING; i++) if (c->ev_up[i]) hipEventDestroy(c->ev_up[i]);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

int sgc_ctx_clone(sgc_ctx *src, sgc_ctx **out) {
    if (!src || !out) return fail(SGC_E_ARG, "sgc_ctx_clone: NULL argument");
    *out = nullptr;
    if (!src->has_lib || !src->tables) return fail(SGC_E_STATE, "sgc_ctx_clone: the source ctx has no library");
    sgc_ctx *c = nullptr;
    const int rc = sgc_init(src->device, &c);
    if (rc) return rc;
    // the tables (shared, read-only) and everything that describes them; streams, scratch and samples are the clone's own
    c->tables = src->tables;
    c->has_lib = src->has_lib; c->one_mm = src->one_mm; c->rec16 = src->rec16; c->n = src->n; c->L = src->L;
    c->d_lib_slots = src->d_lib_slots; c->d_perm_slots = src->d_perm_slots; c->d_lib_cuckoo = src->d_lib_cuckoo;
    c->d_lib_vals = src->d_lib_vals; c->d_perm_vals = src->d_perm_vals;
    c->v_lib = src->v_lib; c->v_perm = src->v_perm;
    c->d_bloom_lib = src->d_bloom_lib; c->d_bloom_perm = src->d_bloom_perm; c->b_lib = src->b_lib; c->b_perm = src->b_perm;
    c->perm_entries = src->perm_entries;
    c->has_core = src->has_core; c->d_amb = src->d_amb; c->d_core_filt = src->d_core_filt;
    for (int k = 0; k < 2; k++) {
        c->d_core_ents[k] = src->d_core_ents[k]; c->d_core_gids[k] = src->d_core_gids[k]; c->d_core_starts[k] = src->d_core_starts[k];
        c->v_core[k] = src->v_core[k];
        c->d_bytes_tags[k] = src->d_bytes_tags[k]; c->d_bytes_vals[k] = src->d_bytes_vals[k];
    }
    c->hybrid = src->hybrid; c->n_packed = src->n_packed; c->d_gid_map = src->d_gid_map; c->d_bloom_shadow = src->d_bloom_shadow; c->b_shadow = src->b_shadow;
    c->bytes_mode = src->bytes_mode; c->d_bytes_seqs = src->d_bytes_seqs; c->d_bytes_pl = src->d_bytes_pl; c->v_bytes = src->v_bytes;
    // and the options that shape the passes
    c->variant = src->variant; c->k1_wgs = src->k1_wgs; c->max_chunk = src->max_chunk;
    c->batch_records = src->batch_records; c->dense = src->dense; c->direct = src->direct; c->six_byte = src->six_byte;
    c->five_byte = src->five_byte; c->wide = src->wide; c->balanced = src->balanced; c->tag_sub = src->tag_sub; c->use_cuckoo = src->use_cuckoo; c->place_trials = src->place_trials;
    c->verbose = src->verbose; c->host_routes = src->host_routes;
    *out = c;
    return SGC_OK;
}

int sgc_set_stream(sgc_ctx *c, void *hip_stream) {
    if (!c) return fail(SGC_E_ARG, "sgc_set_stream: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream == SGC_STREAM_OWN ? c->own_stream : (hipStream_t)hip_stream;
    return SGC_OK;
}

void *sgc_get_stream(sgc_ctx *c) { return c ? (void *)c->stream : nullptr; }

void *sgc_alloc_pinned(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void sgc_free_pinned(void *p) { if (p) hipHostFree(p); }

int sgc_set_option(sgc_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return fail(SGC_E_ARG, "sgc_set_option: NULL argument");
    if (!strcmp(key, "variant")) {
        // 4: the shipped pass; 3: the probing resolver (what a library without a core index gets); 1: the generic kernels (what a
        // library without one-word records or packed slots gets).  Rounds 1-3 also kept 0 (direct atomics) and 2 (a lookup kernel
        // that variant 3 superseded): they served no library shape any more and are gone.
        if (value != 1 && value != 3 && value != 4) return fail(SGC_E_ARG, "sgc_set_option: variant must be 1, 3 or 4");
        c->variant = (int)value; return SGC_OK;
    }
    if (!strcmp(key, "dbg")) {
        // timing-only ablation flags / phase stamps of the kernels: they exist only in a library built for it (sgc_kernels.h)
        if (value && !(SGC_ABLATE || SGC_STAMPS_BUILD))
            return fail(SGC_E_ARG, "sgc_set_option: dbg flags need a library built with -DSGC_ABLATE=1 or -DSGC_STAMPS=1 (SGC_HIPCC_FLAGS; tools/tune.py --ablate)");
        c->dbg = (uint32_t)value; return SGC_OK;
    }
    if (!strcmp(key, "balanced")) { c->balanced = value != 0; return SGC_OK; }      // k_count_slices: equal shares of all slice blocks (1) or of each slice's (0)
    if (!strcmp(key, "timeline_dump")) {
        // -DSGC_STAMPS=1 builds: print (and clear) the workgroup timelines the kernels of the LAST pass left with dbg 1048576 (sgc_device.h)
        if (!SGC_STAMPS_BUILD) return fail(SGC_E_ARG, "sgc_set_option: timeline_dump needs a library built with -DSGC_STAMPS=1");
        sgc_part_timeline_dump(); sgc_core_timeline_dump(); fflush(stdout);
        return SGC_OK;
    }
    if (!strcmp(key, "k1_wgs")) { c->k1_wgs = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 65536)); return SGC_OK; }
    if (!strcmp(key, "max_chunk")) {
        if (value < 1 || value > (int64_t)(1ll << 28)) return fail(SGC_E_ARG, "max_chunk out of range (1 .. 2^28: the partition kernel addresses its pool with 32-bit byte offsets)");
        c->max_chunk = (uint64_t)value; return SGC_OK;
    }
    if (!strcmp(key, "batch_records")) {
        if (value < 1 || value > (int64_t)(1ll << 28)) return fail(SGC_E_ARG, "batch_records out of range (1 .. 2^28)");
        c->batch_records = (uint64_t)value; return SGC_OK;          // takes effect for samples that have not pushed asynchronously yet
    }
    if (!strcmp(key, "align_slices")) { c->align_slices = value != 0; return SGC_OK; }       // takes effect at the next sgc_set_library
    if (!strcmp(key, "slice_log2")) {
        if (value != 0 && value != SGC_LDS_LOG2_SLICE && value != SGC_LDS_LOG2_SLICE_BIG) return fail(SGC_E_ARG, "sgc_set_option: slice_log2 must be 0, 12 or 13");
        c->slice_log2 = (int)value; return SGC_OK;
    }
    if (!strcmp(key, "force_bytes")) { c->force_bytes = value != 0; return SGC_OK; }         // takes effect at the next sgc_set_library
    if (!strcmp(key, "host_routes")) { c->host_routes = value != 0; return SGC_OK; }
    if (!strcmp(key, "hybrid")) { c->allow_hybrid = value != 0; return SGC_OK; }             // 0: a library with any byte outside ACGT is served by the byte-string path alone (next sgc_set_library)
    if (!strcmp(key, "dense")) { c->dense = value != 0; return SGC_OK; }
    if (!strcmp(key, "direct")) { c->direct = value != 0; return SGC_OK; }
    if (!strcmp(key, "six_byte")) { c->six_byte = value != 0; return SGC_OK; }
    if (!strcmp(key, "five_byte")) { c->five_byte = value != 0; return SGC_OK; }
    if (!strcmp(key, "wide")) { c->wide = value != 0; return SGC_OK; }
    if (!strcmp(key, "tag_sub")) { c->tag_sub = value != 0; return SGC_OK; }
    if (!strcmp(key, "cuckoo")) { c->use_cuckoo = value != 0; return SGC_OK; }
    if (!strcmp(key, "rest_filter")) { c->rest_filter = value != 0; return SGC_OK; }         // takes effect at the next sgc_set_library
    if (!strcmp(key, "host_build")) { c->host_build = value != 0; return SGC_OK; }          // takes effect at the next sgc_set_library
    if (!strcmp(key, "perm_bloom_bits")) {
        if (value < 1 || value > 64) return fail(SGC_E_ARG, "perm_bloom_bits must be 1..64");
        c->perm_bloom_bits = (uint32_t)value; return SGC_OK;
    }
    if (!strcmp(key, "verbose")) { c->verbose = value != 0; return SGC_OK; }
    if (!strcmp(key, "vmm_chunk_mb")) {          // takes effect when the buffers are next (re)allocated
        if (value < 0 || value > 1024) return fail(SGC_E_ARG, "vmm_chunk_mb must be 0 (plain hipMalloc) .. 1024");
        c->vmm_chunk = (size_t)value << 20; return SGC_OK;
    }
    if (!strcmp(key, "vmm_runs")) { c->vmm_runs = value != 0; return SGC_OK; }
    if (!strcmp(key, "vmm_shuffle")) { c->vmm_shuffle = (uint64_t)value; return SGC_OK; }
    if (!strcmp(key, "vmm_chunk_kb")) { c->vmm_chunk = (size_t)value << 10; return SGC_OK; }       // experiments (1 KB = physically contiguous)
    if (!strcmp(key, "drop_scratch")) {          // frees the pass scratch: the next pass allocates it afresh (with the options of that moment)
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        dev_free(c->d_pool); c->d_pool = nullptr; c->pool_cap = 0; c->placed_pool = nullptr;
        dev_free(c->d_cbuf); c->d_cbuf = nullptr; c->cbuf_cap = 0;
        return SGC_OK;
    }
    if (!strcmp(key, "place_trials")) {
        if (value < 1 || value > 64) return fail(SGC_E_ARG, "place_trials must be 1..64");
        c->place_trials = (int)value; c->placed_pool = nullptr;
        return SGC_OK;
    }
    if (!strcmp(key, "print_occupancy")) {
        sgc_core_print_occupancy();
        fprintf(stderr, "scratch: pool %p (%zu MB) runs %p (%zu MB) desc %p\n", c->d_pool, c->pool_cap >> 20, c->d_cbuf, c->cbuf_cap >> 20, c->d_desc);
        return SGC_OK;
    }
    return fail(SGC_E_ARG, std::string("sgc_set_option: unknown key ") + key);
}

static int upload_bloom(const std::vector<uint64_t> &keys, uint32_t log2_words, uint64_t **d_words, sgc_bloom_view *v,
                        hipStream_t st) {
    std::vector<uint64_t> words;
    sgc_build_bloom(keys, log2_words, words);
    HIP_TRY(hipMalloc((void **)d_words, words.size() * 8));
    HIP_TRY(hipMemcpyAsync(*d_words, words.data(), words.size() * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    v->words = *d_words; v->log2_words = log2_words; v->pad_ = 0;
    return SGC_OK;
}

static int upload_table(const sgc_host_table &h, uint64_t **d_slots, uint32_t **d_vals, sgc_table_view *v,
                        hipStream_t st) {
    const size_t nslots = h.slots.size();
    HIP_TRY(hipMalloc((void **)d_slots, nslots * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(*d_slots, h.slots.data(), nslots * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    if (h.gid_bits == 0) {
        HIP_TRY(hipMalloc((void **)d_vals, nslots * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(*d_vals, h.vals.data(), nslots * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    v->slots = *d_slots; v->vals = *d_vals; v->log2_slots = h.log2_slots; v->gid_bits = h.gid_bits;
    v->log2_slice = h.log2_slice; v->core_cl = h.core_cl;
    return SGC_OK;
}

// Libraries with bytes outside ACGT or longer than SGC_MAX_GUIDE_LEN: hashed byte-string tables (sgc_bytes.h), built on the host
static int set_library_bytes(sgc_ctx *c, const uint8_t *seqs, uint32_t n, uint32_t L, bool one_mm) {
    sgc_host_bytes hb;
    std::string err;
    // the children table (the 'Permuter' of the byte-string path) is built on the device where the device builder covers the
    // library (sgc_build.hip: under 4M guides of under 128 bytes) — the host builder needs half a second for 100k guides of 20
    const bool dev_children = one_mm && !c->host_build && sgc_device_bytes_children_supported(n, L);
    const int rc = sgc_build_bytes_tables(seqs, n, L, one_mm && !dev_children, hb, err);
    if (rc != SGC_OK) return fail(rc, "sgc_set_library: " + err);
    hipError_t e = hipMalloc((void **)&c->d_bytes_seqs, (size_t)n * L);
    auto up = [&](void **d, const void *h, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(d, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, c->stream);
    };
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_bytes_seqs, seqs, (size_t)n * L, hipMemcpyHostToDevice, c->stream);
    up((void **)&c->d_bytes_tags[0], hb.lib_tag.data(), hb.lib_tag.size() * 8);
    up((void **)&c->d_bytes_vals[0], hb.lib_val.data(), hb.lib_val.size() * 4);
    uint32_t perm_log2 = hb.perm_log2;
    void *d_scr = nullptr; unsigned long long *d_ent = nullptr;
    if (one_mm && !dev_children) {
        up((void **)&c->d_bytes_tags[1], hb.perm_tag.data(), hb.perm_tag.size() * 8);
        up((void **)&c->d_bytes_vals[1], hb.perm_val.data(), hb.perm_val.size() * 4);
        up((void **)&c->d_bytes_pl, hb.perm_pl.data(), hb.perm_pl.size() * 4);
    } else if (dev_children) {
        // room for every child (n L 4 of them at most) at load <= 0.5, as the host builder sizes it for the children it keeps
        perm_log2 = 4;
        while ((1ull << perm_log2) < (uint64_t)n * L * 4u * 2u + 1u) perm_log2++;
        const size_t slots = (size_t)1 << perm_log2;
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_bytes_tags[1], slots * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_bytes_vals[1], slots * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_bytes_pl, slots * 4);
        if (e == hipSuccess) e = hipMalloc(&d_scr, sgc_device_bytes_children_scratch(n, L));
        if (e == hipSuccess) e = hipMalloc((void **)&d_ent, 8);
        if (e == hipSuccess) e = hipMemsetAsync(c->d_bytes_tags[1], 0xFF, slots * 8, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_ent, 0, 8, c->stream);
        if (e == hipSuccess) {
            const sgc_bytes_view v0{c->d_bytes_seqs, c->d_bytes_tags[0], c->d_bytes_vals[0], nullptr, nullptr, nullptr, n, L, hb.lib_log2, perm_log2};
            if (sgc_device_bytes_children(c->stream, v0, c->d_bytes_tags[1], c->d_bytes_vals[1], c->d_bytes_pl, perm_log2, d_ent, d_scr) != 0) e = hipErrorUnknown;
        }
        unsigned long long ent = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&ent, d_ent, 8, hipMemcpyDeviceToHost, c->stream);
        const hipError_t e3 = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = e3;
        hb.perm_entries = ent;
        if (d_scr) hipFree(d_scr);
        if (d_ent) hipFree(d_ent);
    }
    const hipError_t e2 = hipStreamSynchronize(c->stream);          // the host vectors must outlive the copies
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) {
        free_tables(c);
        return fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_set_library: ") + hipGetErrorString(e));
    }
    c->v_bytes = sgc_bytes_view{c->d_bytes_seqs, c->d_bytes_tags[0], c->d_bytes_vals[0], c->d_bytes_tags[1], c->d_bytes_vals[1], c->d_bytes_pl,
                                n, L, hb.lib_log2, perm_log2};
    c->perm_entries = hb.perm_entries;
    c->bytes_mode = true;
    c->n = n; c->L = L; c->one_mm = one_mm; c->rec16 = false;
    return SGC_OK;
}

// the packed tables of an all-ACGT set of guides (every exit with an error leaves a half-built ctx for the caller's free_tables)
static int set_library_packed(sgc_ctx *c, const uint8_t *seqs, uint32_t n, uint32_t L, int enable_1mm) {
    std::vector<uint64_t> keys;
    sgc_host_table h_lib, h_perm;
    std::string err;
    // option "verbose": where the time of this call goes, phase by phase (stderr)
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!c->verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "sgc_set_library: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
        lap_t = now;
    };
    const uint32_t want_cl = (c->align_slices && L >= 4 && L <= SGC_REC8_MAXL) ? (L - 2) / 2 : 0;
    int rc = sgc_build_library_table(seqs, n, L, c->slice_log2 ? (uint32_t)c->slice_log2 : sgc_choose_log2_slice(n), want_cl, keys, h_lib, err);
    if (rc != SGC_OK) return fail(rc, "sgc_set_library: " + err);
    lap("library table (host)");
    rc = upload_table(h_lib, &c->d_lib_slots, &c->d_lib_vals, &c->v_lib, c->stream);
    if (rc != SGC_OK) { free_tables(c); return rc; }
    rc = upload_bloom(keys, SGC_LIB_BLOOM_LOG2_WORDS, &c->d_bloom_lib, &c->b_lib, c->stream);
    if (rc != SGC_OK) { free_tables(c); return rc; }
    lap("upload + library filter");
    if (sgc_part_supported(c->v_lib, L > SGC_REC8_MAXL)) {
        // the partitioned path probes a two-choice image of every slice (k_count_slices<CUCKOO>)
        std::vector<uint64_t> ck;
        if (sgc_build_slice_cuckoo(h_lib, ck)) {
            hipError_t e = hipMalloc((void **)&c->d_lib_cuckoo, ck.size() * 8);
            if (e == hipSuccess) e = hipMemcpyAsync(c->d_lib_cuckoo, ck.data(), ck.size() * 8, hipMemcpyHostToDevice, c->stream);
            const hipError_t e2 = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = e2;
            if (e != hipSuccess) { free_tables(c); return fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_set_library: ") + hipGetErrorString(e)); }
        }
    }
    lap("two-choice slice image");
    c->v_perm = sgc_table_view{nullptr, nullptr, 0, h_lib.gid_bits, 0, 0};
    c->perm_entries = 0;
    std::vector<uint64_t> amb;
    bool device_build = false;
    if (enable_1mm) {
        const uint32_t bpk = c->perm_bloom_bits;   // Bloom bits per child, rounded up to a power-of-two word count: 6.0 M children -> 8 MiB
        const uint64_t n_children = (uint64_t)n * 3 * L;
        const uint32_t bloom_log2 = sgc_bloom_log2_words(n_children, bpk, 10, 24);
        device_build = h_lib.gid_bits != 0 && !c->host_build;
        if (device_build) {
            // children, their table, its filter and the ambiguity masks are built on the GPU (sgc_build.hip)
            const uint32_t pl2 = sgc_permute_log2_slots(n_children);
            uint64_t *d_keys = nullptr; void *d_scr = nullptr; unsigned long long *d_ent = nullptr;
            auto drop = [&]() { if (d_keys) hipFree(d_keys); if (d_scr) hipFree(d_scr); if (d_ent) hipFree(d_ent); };
            hipError_t e = hipMalloc((void **)&d_keys, (size_t)n * 8);
            if (e == hipSuccess) e = hipMalloc(&d_scr, sgc_device_build_scratch_bytes(n, L));
            if (e == hipSuccess) e = hipMalloc((void **)&d_ent, 8);
            if (e == hipSuccess) e = hipMalloc((void **)&c->d_perm_slots, sizeof(uint64_t) << pl2);
            if (e == hipSuccess) e = hipMalloc((void **)&c->d_bloom_perm, sizeof(uint64_t) << bloom_log2);
            if (e == hipSuccess) e = hipMalloc((void **)&c->d_amb, (size_t)n * 16);
            if (e == hipSuccess) e = hipMemcpyAsync(d_keys, keys.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(c->d_perm_slots, 0xFF, sizeof(uint64_t) << pl2, c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(c->d_bloom_perm, 0, sizeof(uint64_t) << bloom_log2, c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(c->d_amb, 0, (size_t)n * 16, c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_ent, 0, 8, c->stream);
            if (e != hipSuccess) { drop(); free_tables(c); return fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_set_library: ") + hipGetErrorString(e)); }
            if (sgc_device_build_permute(c->stream, d_keys, n, L, c->v_lib, c->d_perm_slots, pl2, h_lib.gid_bits, c->d_bloom_perm,
                                         bloom_log2, c->d_amb, d_ent, d_scr) != 0) {
                drop(); free_tables(c);
                return fail(SGC_E_HIP, "sgc_set_library: device build of the single-mismatch table failed");
            }
            unsigned long long ent = 0;
            e = hipMemcpyAsync(&ent, d_ent, 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            drop();
            if (e != hipSuccess) { free_tables(c); return fail(SGC_E_HIP, std::string("sgc_set_library: ") + hipGetErrorString(e)); }
            c->perm_entries = ent;
            c->v_perm = sgc_table_view{c->d_perm_slots, nullptr, pl2, h_lib.gid_bits, pl2, 0};
            c->b_perm.words = c->d_bloom_perm; c->b_perm.log2_words = bloom_log2; c->b_perm.pad_ = 0;
        } else {
            std::vector<uint64_t> child_keys;
            sgc_build_permute_table(keys, L, h_lib, h_perm, &child_keys, &amb);
            rc = upload_table(h_perm, &c->d_perm_slots, &c->d_perm_vals, &c->v_perm, c->stream);
            if (rc != SGC_OK) { free_tables(c); return rc; }
            c->perm_entries = h_perm.entries;
            rc = upload_bloom(child_keys, bloom_log2, &c->d_bloom_perm, &c->b_perm, c->stream);
            if (rc != SGC_OK) { free_tables(c); return rc; }
        }
    }
    lap("single-mismatch table");
    {
        // core indexes (variant 4): span bases [2, L) cut in two; absent => the probing resolver stays in charge.  Built for -x
        // too: there one exact-only pass over core A takes the place of the probing resolver (k_core<EXACT>)
        if (L >= 4 && L <= SGC_REC8_MAXL && h_lib.gid_bits != 0) {
            const uint32_t ca = (L - 2) / 2;
            sgc_host_core hc[2];
            if (sgc_build_core_index(keys, L, 2, ca, hc[0]) && sgc_build_core_index(keys, L, 2 + ca, L - 2 - ca, hc[1])) {
                hipError_t e = hipSuccess;
                std::vector<uint32_t> filt;
                lap("core indexes (host)");
                const uint32_t fl2 = sgc_rest_filter_log2(n);
                if (c->rest_filter && enable_1mm) sgc_build_rest_filter(keys, hc[0].cs, hc[0].cl, fl2, filt);
                lap("rest filter (host)");
                for (int k = 0; k < 2 && e == hipSuccess; k++) {
                    e = hipMalloc((void **)&c->d_core_ents[k], hc[k].ents.size() * 8);
                    if (e == hipSuccess) e = hipMalloc((void **)&c->d_core_gids[k], hc[k].gids.size() * 4);
                    if (e == hipSuccess) e = hipMalloc((void **)&c->d_core_starts[k], hc[k].starts.size() * 2);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_core_ents[k], hc[k].ents.data(), hc[k].ents.size() * 8, hipMemcpyHostToDevice, c->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_core_gids[k], hc[k].gids.data(), hc[k].gids.size() * 4, hipMemcpyHostToDevice, c->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_core_starts[k], hc[k].starts.data(), hc[k].starts.size() * 2, hipMemcpyHostToDevice, c->stream);
                    c->v_core[k] = sgc_core_view{c->d_core_ents[k], c->d_core_gids[k], c->d_core_starts[k], hc[k].log2_p, hc[k].cs, hc[k].cl, 0, nullptr};
                }
                if (e == hipSuccess && !filt.empty()) {
                    e = hipMalloc((void **)&c->d_core_filt, filt.size() * 4);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_core_filt, filt.data(), filt.size() * 4, hipMemcpyHostToDevice, c->stream);
                    c->v_core[0].filt = c->d_core_filt; c->v_core[0].filt_log2 = fl2;
                }
                if (e == hipSuccess && enable_1mm && !device_build) {
                    e = hipMalloc((void **)&c->d_amb, amb.size() * 8);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_amb, amb.data(), amb.size() * 8, hipMemcpyHostToDevice, c->stream);
                }
                if (e == hipSuccess && enable_1mm && c->d_amb)
                    for (int k = 0; k < 2; k++) sgc_flag_ambiguous(c->stream, c->d_core_gids[k], hc[k].gids.size(), c->d_amb);
                // the host vectors above must outlive the copies: synchronise before they go out of scope, error or not
                const hipError_t e2 = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = e2;
                if (e != hipSuccess) {
                    free_tables(c);                       // nothing of a half-built library stays behind
                    return fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_set_library: core index upload: ") + hipGetErrorString(e));
                }
                c->has_core = true;
                lap("core index upload");
            }
        }
    }
    c->n = n; c->L = L; c->one_mm = enable_1mm != 0; c->rec16 = L > SGC_REC8_MAXL;
    return SGC_OK;
}

int sgc_set_library(sgc_ctx *c, const uint8_t *seqs, uint32_t n, uint32_t L, int enable_1mm) {
    if (!c || !seqs) return fail(SGC_E_ARG, "sgc_set_library: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_tables(c);
    if (L == 0 || n == 0) return fail(n == 0 ? SGC_E_ARG : SGC_E_UNSUPPORTED, n == 0 ? "sgc_set_library: empty library" : "sgc_set_library: guide length 0");
    // which guides can the packed records carry (ACGT only, at most SGC_MAXL bases)?
    std::vector<uint32_t> acgt;                      // library indexes of the all-ACGT guides
    if (L <= SGC_MAXL && !c->force_bytes) {
        acgt.reserve(n);
        for (uint32_t g = 0; g < n; g++) {
            const uint8_t *q = seqs + (size_t)g * L;
            bool ok = true;
            for (uint32_t k = 0; k < L && ok; k++) ok = q[k] == 'A' || q[k] == 'C' || q[k] == 'G' || q[k] == 'T';
            if (ok) acgt.push_back(g);
        }
    }
    int rc;
    if (acgt.size() == n) {
        rc = set_library_packed(c, seqs, n, L, enable_1mm);
    } else if (c->allow_hybrid && !acgt.empty() && (n - acgt.size()) * 2 <= (size_t)n) {
        // hybrid (sgc_bytes.hip): byte-string tables over ALL guides, packed tables over the ACGT ones (numbered 0 .. n_packed - 1),
        // the guide map, and the Bloom filter of the shadow keys
        rc = set_library_bytes(c, seqs, n, L, enable_1mm != 0);
        if (rc == SGC_OK) {
            const uint64_t bytes_perm_entries = c->perm_entries;
            std::vector<uint8_t> compact((size_t)acgt.size() * L);
            for (size_t i = 0; i < acgt.size(); i++) memcpy(&compact[i * L], seqs + (size_t)acgt[i] * L, L);
            rc = set_library_packed(c, compact.data(), (uint32_t)acgt.size(), L, enable_1mm);
            if (rc == SGC_OK) {
                std::vector<uint64_t> shadow;
                std::vector<bool> is_acgt(n, false);
                for (uint32_t g : acgt) is_acgt[g] = true;
                for (uint32_t g = 0; g < n; g++) {
                    if (is_acgt[g]) continue;
                    const uint8_t *q = seqs + (size_t)g * L;
                    uint32_t bad = 0, at = 0;
                    for (uint32_t k = 0; k < L; k++) if (!(q[k] == 'A' || q[k] == 'C' || q[k] == 'G' || q[k] == 'T')) { bad++; at = k; }
                    if (bad != 1) continue;              // two or more such bytes: no all-ACGT window is within one substitution
                    uint64_t key = 0;
                    for (uint32_t k = 0; k < L; k++) if (k != at) key |= (uint64_t)sgc_base_code(q[k]) << (2 * k);
                    for (uint64_t b = 0; b < 4; b++) shadow.push_back(key | (b << (2 * at)));
                }
                uint32_t lw = 6;
                while (lw < 20 && (1ull << lw) * 8 < shadow.size() * 2 + 64) lw++;       // >= 32 bits per key: almost no false positives
                rc = upload_bloom(shadow, lw, &c->d_bloom_shadow, &c->b_shadow, c->stream);
                if (rc == SGC_OK) {
                    hipError_t e = hipMalloc((void **)&c->d_gid_map, acgt.size() * 4);
                    if (e == hipSuccess) e = hipMemcpyAsync(c->d_gid_map, acgt.data(), acgt.size() * 4, hipMemcpyHostToDevice, c->stream);
                    const hipError_t e2 = hipStreamSynchronize(c->stream);
                    if (e == hipSuccess) e = e2;
                    if (e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_set_library: ") + hipGetErrorString(e));
                }
                c->n_packed = (uint32_t)acgt.size();
                c->n = n; c->hybrid = true; c->bytes_mode = false;
                c->perm_entries = bytes_perm_entries;            // of the whole library, 'N' children included (src/permutes.rs map.len())
            }
        }
    } else if (L <= SGC_BYTES_MAXL) {
        rc = set_library_bytes(c, seqs, n, L, enable_1mm != 0);
    } else {
        rc = fail(SGC_E_UNSUPPORTED, "sgc_set_library: guides longer than 65535 bytes");
    }
    if (rc != SGC_OK) { const std::string keep = g_err; free_tables(c); g_err = keep; return rc; }
    c->has_lib = true;
    adopt_tables(c);
    return SGC_OK;
}

int sgc_library_info(sgc_ctx *c, sgc_lib_info *out) {
    if (!c || !out) return fail(SGC_E_ARG, "sgc_library_info: NULL argument");
    if (!c->has_lib) return fail(SGC_E_STATE, "sgc_library_info: no library set");
    memset(out, 0, sizeof(*out));
    out->n_guides = c->n; out->guide_len = c->L; out->record_bytes = c->rec16 ? 16 : 8;
    out->one_mismatch = c->one_mm;
    if (c->hybrid) {
        // no packed records from outside (a host cannot know which reads need the byte-string chain): reads and FASTQ text only
        out->record_bytes = 0; out->path = 2;
        out->lib_slots = 1ull << c->v_bytes.lib_log2; out->perm_slots = c->one_mm ? 1ull << c->v_bytes.perm_log2 : 0;
        out->perm_entries = c->perm_entries; out->core_partitions = c->has_core ? 1ull << c->v_core[0].log2_p : 0;
        out->slices = sgc_part_supported(c->v_lib, c->rec16) ? 1u << (c->v_lib.log2_slots - c->v_lib.log2_slice) : 0;
        out->table_bytes = (uint64_t)c->n * c->L + out->lib_slots * 12 + out->perm_slots * 16 + ((1ull << c->v_lib.log2_slots) * 16);
        out->reserved_ = c->n_packed;          // guides the packed pass serves
        return SGC_OK;
    }
    if (c->bytes_mode) {
        out->record_bytes = 0;
        out->lib_slots = 1ull << c->v_bytes.lib_log2;
        out->perm_slots = c->one_mm ? 1ull << c->v_bytes.perm_log2 : 0;
        out->perm_entries = c->perm_entries;
        out->table_bytes = (uint64_t)c->n * c->L + out->lib_slots * 12 + out->perm_slots * 16;
        return SGC_OK;
    }
    out->lib_slots = 1ull << c->v_lib.log2_slots;
    out->perm_slots = c->one_mm ? 1ull << c->v_perm.log2_slots : 0;
    out->perm_entries = c->perm_entries;
    if (sgc_part_supported(c->v_lib, c->rec16)) {
        out->path = c->has_core ? 4 : 3;
        out->slices = 1u << (c->v_lib.log2_slots - c->v_lib.log2_slice);
        // as count_records decides (default options): five-byte, six-byte or whole records in the slice blocks
        const int sub = c->has_core ? (int)c->v_core[0].log2_p - (int)(c->v_lib.log2_slots - c->v_lib.log2_slice) : -1;
        const bool direct = c->has_core && c->v_lib.core_cl == c->v_core[0].cl && c->v_lib.log2_slice < c->v_lib.log2_slots && sub >= 0 && sub <= 2;
        const uint32_t sb = c->v_lib.log2_slots - c->v_lib.log2_slice;
        out->slice_record_bytes = !direct ? 8 : (2u * (c->L + 2u) - sb <= 40u && sb + (uint32_t)sub <= 2u * c->v_lib.core_cl) ? 5 : (2u * (c->L + 2u) + 2u <= 48u) ? 6 : 8;
    } else {
        out->path = 1;
    }
    const uint64_t per = c->v_lib.gid_bits ? 8 : 12;
    out->table_bytes = (out->lib_slots + out->perm_slots) * per + (c->d_lib_cuckoo ? out->lib_slots * 8 : 0);
    if (c->has_core) {
        out->core_partitions = 1ull << c->v_core[0].log2_p;
        for (int k = 0; k < 2; k++)
            out->table_bytes += ((uint64_t)SGC_CORE_EMAX * 12 + SGC_CORE_STARTS * 2) << c->v_core[k].log2_p;
        if (c->d_core_filt) out->table_bytes += (uint64_t)3 << (c->v_core[0].filt_log2 - 3);
        out->table_bytes += (uint64_t)c->n * 16;
    }
    return SGC_OK;
}

int sgc_lookup(sgc_ctx *c, const uint8_t *tokens, uint64_t n, int which, int32_t *gid_out) {
    if (!c || (!tokens && n) || (!gid_out && n)) return fail(SGC_E_ARG, "sgc_lookup: NULL argument");
    if (!c->has_lib) return fail(SGC_E_STATE, "sgc_lookup: no library set");
    if (which < 0 || which > 2) return fail(SGC_E_ARG, "sgc_lookup: which must be 0, 1 or 2");
    if (n == 0) return SGC_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (c->bytes_mode || c->hybrid) {
        int rc = ensure(&c->d_stage, &c->stage_cap, n * c->L);
        if (rc) return rc;
        rc = ensure(&c->d_aux, &c->aux_cap, n * 4);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_stage, tokens, n * c->L, hipMemcpyHostToDevice, c->stream));
        sgc_launch_bytes_lookup(c->stream, c->v_bytes, (const uint8_t *)c->d_stage, n, which, c->one_mm, (int32_t *)c->d_aux);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(gid_out, c->d_aux, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return SGC_OK;
    }
    std::vector<uint64_t> keys(n);
    for (uint64_t i = 0; i < n; i++)
        if (!sgc_pack_key(tokens + i * c->L, c->L, keys[i])) keys[i] = SGC_EMPTY;   // non-ACGT token: no match
    int rc = ensure(&c->d_stage, &c->stage_cap, n * 8);
    if (rc) return rc;
    rc = ensure(&c->d_aux, &c->aux_cap, n * 4);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage, keys.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    sgc_launch_lookup(c->stream, (const uint64_t *)c->d_stage, n, c->v_lib, c->v_perm, which, c->one_mm,
                      (int32_t *)c->d_aux);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(gid_out, c->d_aux, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SGC_OK;
}

int sgc_pack_reads_host(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L, int reverse,
                        uint32_t offset, int position_recursion, void *records_out) {
    if ((!seqs && n) || !offsets || (!records_out && n)) return fail(SGC_E_ARG, "sgc_pack_reads_host: NULL argument");
    const uint32_t rb = sgc_record_bytes(L);
    if (!rb) return fail(SGC_E_UNSUPPORTED, "sgc_pack_reads_host: guide length outside 1..30");
    uint64_t *out = (uint64_t *)records_out;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t span, status;
        sgc_pack_one(seqs + offsets[i], offsets[i + 1] - offsets[i], L, reverse, offset, position_recursion, span,
                     status);
        if (rb == 16) { out[2 * i] = span; out[2 * i + 1] = status; }
        else out[i] = span | (status << (2 * (L + 2)));
    }
    return SGC_OK;
}

int sgc_pack_reads_device(sgc_ctx *c, const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, int reverse,
                          uint32_t offset, int position_recursion, void *d_records_out) {
    if (!c || !d_offsets || (!d_records_out && n)) return fail(SGC_E_ARG, "sgc_pack_reads_device: NULL argument");
    if (!c->has_lib) return fail(SGC_E_STATE, "sgc_pack_reads_device: no library set");
    if (c->bytes_mode || c->hybrid) return fail(SGC_E_STATE, "sgc_pack_reads_device: this library has no packed record format (sgc_library_info: record_bytes == 0)");
    if (n == 0) return SGC_OK;
    HIP_TRY(hipSetDevice(c->device));
    {
        timed t(c, T_PACK);
        sgc_launch_pack_reads_lds(c->stream, d_seqs, d_offsets, n, c->L, c->rec16, reverse != 0, offset,
                                  position_recursion != 0, (uint64_t *)d_records_out);
    }
    HIP_TRY(hipGetLastError());
    return SGC_OK;
}

int sgc_sample_begin(sgc_ctx *c, sgc_sample **out, int reverse, uint32_t offset, int position_recursion) {
    if (!c || !out) return fail(SGC_E_ARG, "sgc_sample_begin: NULL argument");
    *out = nullptr;
    if (!c->has_lib) return fail(SGC_E_STATE, "sgc_sample_begin: no library set");
    HIP_TRY(hipSetDevice(c->device));
    sgc_sample *s = new (std::nothrow) sgc_sample();
    if (!s) return fail(SGC_E_OOM, "sgc_sample_begin: out of host memory");
    s->ctx = c; s->reverse = reverse != 0; s->offset = offset; s->recursion = position_recursion != 0;
    // one allocation: u64 counts[n] | u64 matched | u64 spare | u64 err[2] | u32 counts[n]  (one memset resets a sample)
    s->state_bytes = (size_t)c->n * 8 + 32 + (size_t)c->n * 4 + (c->hybrid ? (size_t)c->n_packed * 4 : 0);
    hipError_t e = hipMalloc((void **)&s->d_c64, s->state_bytes);
    if (e != hipSuccess) { s->d_c64 = nullptr; sgc_sample_free(s); return fail(SGC_E_OOM, std::string("sgc_sample_begin: ") + hipGetErrorString(e)); }
    s->d_matched = s->d_c64 + c->n;
    s->d_err = s->d_c64 + c->n + 2;
    s->d_c32 = (uint32_t *)(s->d_c64 + c->n + 4);
    s->d_c32p = c->hybrid ? s->d_c32 + c->n : nullptr;
    int rc = sgc_sample_reset(s);
    if (rc) { sgc_sample_free(s); return rc; }
    *out = s;
    return SGC_OK;
}

int sgc_sample_reset(sgc_sample *s) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_reset: NULL");
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(s->d_c64, 0, s->state_bytes, c->stream));
    s->total = 0; s->since_fold = 0; s->fastq_pushed = false;
    s->acc_fill = 0; s->acc_last_up = nullptr;        // records uploaded but not yet counted are dropped with the counts
    return SGC_OK;
}

void sgc_sample_free(sgc_sample *s) {
    if (!s) return;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    if (s->d_acc[0] || s->d_acc[1]) hipStreamSynchronize(s->ctx->copy_stream);
    for (int i = 0; i < 2; i++) { if (s->d_acc[i]) hipFree(s->d_acc[i]); if (s->ev_acc_use[i]) hipEventDestroy(s->ev_acc_use[i]); }
    if (s->d_c64) hipFree(s->d_c64);
    delete s;
}

int sgc_sample_push_packed(sgc_sample *s, const void *records, uint64_t n, int where) {
    if (!s || (!records && n)) return fail(SGC_E_ARG, "sgc_sample_push_packed: NULL argument");
    if (n == 0) return SGC_OK;
    sgc_ctx *c = s->ctx;
    if (c->bytes_mode || (c->hybrid && !c->host_routes))
        return fail(SGC_E_STATE, "sgc_sample_push_packed: this library has no packed record format (sgc_library_info: record_bytes == 0); "
                                 "push reads or FASTQ text");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * (c->rec16 ? 16 : 8);
    const uint64_t *d = (const uint64_t *)records;
    if (where == SGC_MEM_HOST) {
        int rc = ensure(&c->d_stage, &c->stage_cap, bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_stage, records, bytes, hipMemcpyHostToDevice, c->stream));
        d = (const uint64_t *)c->d_stage;
    } else if (where != SGC_MEM_DEVICE) {
        return fail(SGC_E_ARG, "sgc_sample_push_packed: where must be SGC_MEM_HOST or SGC_MEM_DEVICE");
    }
    return count_records(s, d, n);
}

// the records uploaded into the current batch buffer so far -> one count pass (after their uploads, on the ctx stream)
static int flush_batch(sgc_sample *s) {
    if (s->acc_fill == 0) return SGC_OK;
    sgc_ctx *c = s->ctx;
    const int cur = s->acc_cur;
    if (s->acc_last_up) HIP_TRY(hipStreamWaitEvent(c->stream, s->acc_last_up, 0));
    const uint64_t n = s->acc_fill;
    s->acc_fill = 0; s->acc_cur = cur ^ 1; s->acc_last_up = nullptr;
    const int rc = count_records(s, (const uint64_t *)s->d_acc[cur], n);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev_acc_use[cur], c->stream));
    s->acc_used[cur] = true;
    return SGC_OK;
}

int sgc_sample_push_packed_async(sgc_sample *s, const void *records, uint64_t n) {
    if (!s || (!records && n)) return fail(SGC_E_ARG, "sgc_sample_push_packed_async: NULL argument");
    if (n == 0) return SGC_OK;
    sgc_ctx *c = s->ctx;
    if (c->bytes_mode || (c->hybrid && !c->host_routes))
        return fail(SGC_E_STATE, "sgc_sample_push_packed_async: this library has no packed record format (sgc_library_info: record_bytes == 0)");
    HIP_TRY(hipSetDevice(c->device));
    const size_t rb = c->rec16 ? 16 : 8;
    if (!s->d_acc[0]) {
        s->acc_cap = c->batch_records;
        for (int i = 0; i < 2; i++) {
            hipError_t e = hipMalloc(&s->d_acc[i], (size_t)s->acc_cap * rb);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_acc_use[i], hipEventDisableTiming);
            if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? SGC_E_OOM : SGC_E_HIP, std::string("sgc_sample_push_packed_async: ") + hipGetErrorString(e));
        }
    }
    const uint8_t *src = (const uint8_t *)records;
    while (n) {
        const int cur = s->acc_cur;
        const uint64_t m = std::min<uint64_t>(n, s->acc_cap - s->acc_fill);
        // a batch buffer is overwritten only after the count pass that read it
        if (s->acc_fill == 0 && s->acc_used[cur]) HIP_TRY(hipStreamWaitEvent(c->copy_stream, s->ev_acc_use[cur], 0));
        hipEvent_t up = c->ev_up[c->n_up % sgc_ctx::UP_RING];
        if (c->n_up >= (uint64_t)sgc_ctx::UP_RING) HIP_TRY(hipEventSynchronize(up));      // at most UP_RING uploads in flight
        {
            timed t(c, T_H2D, false, c->copy_stream);
            HIP_TRY(hipMemcpyAsync((char *)s->d_acc[cur] + (size_t)s->acc_fill * rb, src, (size_t)m * rb, hipMemcpyHostToDevice, c->copy_stream));
        }
        HIP_TRY(hipEventRecord(up, c->copy_stream));
        c->n_up++;
        s->acc_last_up = up;
        s->acc_fill += m;
        src += (size_t)m * rb; n -= m;
        if (s->acc_fill == s->acc_cap) { const int rc = flush_batch(s); if (rc) return rc; }
    }
    return SGC_OK;
}

int sgc_sample_push_reads(sgc_sample *s, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, int where) {
    if (!s || !offsets) return fail(SGC_E_ARG, "sgc_sample_push_reads: NULL argument");
    if (n == 0) return SGC_OK;
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint8_t *d_seqs = seqs; const uint64_t *d_off = offsets;
    if (where == SGC_MEM_HOST) {
        const uint64_t nbytes = offsets[n];
        int rc = ensure(&c->d_stage, &c->stage_cap, nbytes ? nbytes : 1);
        if (rc) return rc;
        // (the offsets live in a buffer of their own: a hybrid ctx reads them again AFTER its packed pass — the byte-string chain of
        // the flagged reads —, and the probing resolver of that pass, variant 3, keeps its segment counts in d_aux)
        rc = ensure(&c->d_lines, &c->lines_cap, (n + 1) * 8);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_stage, seqs, nbytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_lines, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        d_seqs = (const uint8_t *)c->d_stage; d_off = (const uint64_t *)c->d_lines;
    } else if (where != SGC_MEM_DEVICE) {
        return fail(SGC_E_ARG, "sgc_sample_push_reads: where must be SGC_MEM_HOST or SGC_MEM_DEVICE");
    }
    if (c->bytes_mode) return count_bytes(s, d_seqs, d_off, d_off + 1, n);
    void *p = c->d_recs; size_t cap = c->recs_cap;
    int rc = ensure(&p, &cap, (size_t)n * (c->rec16 ? 16 : 8));
    c->d_recs = (uint64_t *)p; c->recs_cap = cap;
    if (rc) return rc;
    {
        timed t(c, T_PACK);
        sgc_launch_pack_reads_lds(c->stream, d_seqs, d_off, n, c->L, c->rec16, s->reverse, s->offset, s->recursion,
                                  c->d_recs);
    }
    HIP_TRY(hipGetLastError());
    if (c->hybrid) return count_hybrid(s, c->d_recs, d_seqs, d_off, d_off + 1, n);
    return count_records(s, c->d_recs, n);
}

int sgc_sample_push_windows(sgc_sample *s, const uint8_t *pieces, const uint64_t *offsets, uint64_t n, int where, uint32_t window_offset) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_push_windows: NULL argument");
    if (window_offset != (s->offset >= 1 ? 1u : 0u)) return fail(SGC_E_ARG, "sgc_sample_push_windows: window_offset is 1, or 0 for a sample with offset 0");
    // (every launch below takes the offset by value: the sample's own is back before anything else can look at it)
    const auto keep = s->offset;
    s->offset = window_offset;
    const int rc = sgc_sample_push_reads(s, pieces, offsets, n, where);
    s->offset = keep;
    return rc;
}

// One part of a FASTQ stream: whole lines, starting at global line number first_line (any phase of the 4-line
// cycle).  n_newlines == UINT64_MAX: unknown — the device counts and the call waits for the count.
static int push_fastq_part(sgc_sample *s, const uint8_t *text, uint64_t n_bytes, int where, uint64_t first_line,
                           uint64_t n_newlines, uint64_t n_lines, uint64_t *n_records_out, uint64_t *n_lines_out) {
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t tiles = sgc_fastq_tiles(n_bytes);
    int rc = ensure(&c->d_aux, &c->aux_cap, ((size_t)tiles + 1) * 4);
    if (rc) return rc;
    const uint8_t *d_text = text;
    int slot = -1;
    if (where == SGC_MEM_HOST) {
        // upload on the copy stream into the buffer the part before last used (its ingest kernels must be done)
        slot = (int)(c->n_up & 1u);
        if (n_bytes > c->text_cap[slot]) {
            if (c->use_recorded[slot]) HIP_TRY(hipEventSynchronize(c->ev_use[slot]));
            rc = ensure(&c->d_text[slot], &c->text_cap[slot], n_bytes);
            if (rc) return rc;
        }
        hipEvent_t up = c->ev_up[c->n_up % sgc_ctx::UP_RING];
        if (c->n_up >= (uint64_t)sgc_ctx::UP_RING) HIP_TRY(hipEventSynchronize(up));      // at most UP_RING uploads in flight
        if (c->use_recorded[slot]) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_use[slot], 0));
        {
            timed t(c, T_H2D, false, c->copy_stream);
            HIP_TRY(hipMemcpyAsync(c->d_text[slot], text, n_bytes, hipMemcpyHostToDevice, c->copy_stream));
        }
        HIP_TRY(hipEventRecord(up, c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, up, 0));
        c->n_up++;
        d_text = (const uint8_t *)c->d_text[slot];
    } else if (where != SGC_MEM_DEVICE) {
        return fail(SGC_E_ARG, "sgc_sample_push_fastq: where must be SGC_MEM_HOST or SGC_MEM_DEVICE");
    }
    uint32_t *tile_scratch = (uint32_t *)c->d_aux;
    uint64_t lines = n_lines;
    { timed t(c, T_PACK); sgc_launch_fastq_count(c->stream, d_text, n_bytes, tile_scratch); }
    if (n_newlines == UINT64_MAX) {
        uint32_t nl = 0; uint8_t last = 0;
        HIP_TRY(hipMemcpyAsync(&nl, tile_scratch + tiles, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        n_newlines = nl;
        lines = (uint64_t)nl + (last == '\n' ? 0 : 1);
    }
    if (n_lines_out) *n_lines_out = lines;
    // records = sequence lines (line % 4 == 1) of the part.  A last line without its '\n' can only be a quality line
    // in a well-formed stream, so the announced newline count decides (a truncated record is the caller's to report).
    const uint64_t n_records = sgc_fastq_records(first_line, lines);
    void *p = c->d_recs; size_t cap = c->recs_cap;
    rc = ensure(&p, &cap, (size_t)(n_records ? n_records : 1) * ((c->rec16 || c->bytes_mode) ? 16 : 8));
    c->d_recs = (uint64_t *)p; c->recs_cap = cap;
    if (rc) return rc;
    if (c->bytes_mode) {
        // no packed records: the (start, end) of every sequence line, then the byte-string chain on the text itself
        uint64_t *starts = c->d_recs, *ends = c->d_recs + n_records;
        { timed t(c, T_PACK, true); sgc_launch_fastq_lines(c->stream, d_text, n_bytes, tile_scratch, first_line, (uint32_t)n_newlines, (uint32_t)lines,
                                                           starts, ends, s->d_err); }
        HIP_TRY(hipGetLastError());
        s->fastq_pushed = true;
        if (n_records_out) *n_records_out = n_records;
        rc = count_bytes(s, d_text, starts, ends, n_records);
        if (slot >= 0) { HIP_TRY(hipEventRecord(c->ev_use[slot], c->stream)); c->use_recorded[slot] = true; }
        return rc;
    }
    {
        timed t(c, T_PACK, true);
        sgc_launch_fastq_pack(c->stream, d_text, n_bytes, tile_scratch, first_line, (uint32_t)n_newlines, (uint32_t)lines, c->L, c->rec16,
                              s->reverse, s->offset, s->recursion, c->d_recs, s->d_err, (c->dbg >> 24) & 3u);
    }
    HIP_TRY(hipGetLastError());
    if (c->hybrid) {
        // the reads' bytes are needed too (route flags; the byte-string chain of the flagged): the (start, end) of every sequence line
        rc = ensure(&c->d_lines, &c->lines_cap, (size_t)(n_records ? n_records : 1) * 16);
        if (rc) return rc;
        uint64_t *starts = (uint64_t *)c->d_lines, *ends = starts + n_records;
        { timed t(c, T_PACK, true); sgc_launch_fastq_lines(c->stream, d_text, n_bytes, tile_scratch, first_line, (uint32_t)n_newlines, (uint32_t)lines,
                                                           starts, ends, s->d_err); }
        HIP_TRY(hipGetLastError());
        s->fastq_pushed = true;
        if (n_records_out) *n_records_out = n_records;
        rc = count_hybrid(s, c->d_recs, d_text, starts, ends, n_records);
        if (slot >= 0) { HIP_TRY(hipEventRecord(c->ev_use[slot], c->stream)); c->use_recorded[slot] = true; }      // the text is read to the end
        return rc;
    }
    if (slot >= 0) { HIP_TRY(hipEventRecord(c->ev_use[slot], c->stream)); c->use_recorded[slot] = true; }
    s->fastq_pushed = true;
    if (n_records_out) *n_records_out = n_records;
    return count_records(s, c->d_recs, n_records);
}

int sgc_sample_push_fastq(sgc_sample *s, const uint8_t *text, uint64_t n_bytes, int where, uint64_t *n_records_out) {
    if (!s || (!text && n_bytes)) return fail(SGC_E_ARG, "sgc_sample_push_fastq: NULL argument");
    if (n_records_out) *n_records_out = 0;
    if (n_bytes == 0) return SGC_OK;
    if (n_bytes > 0xFFF00000ull) return fail(SGC_E_ARG, "sgc_sample_push_fastq: chunk larger than 4 GiB");
    uint64_t lines = 0;
    const int rc = push_fastq_part(s, text, n_bytes, where, 0, UINT64_MAX, UINT64_MAX, n_records_out, &lines);
    if (rc) return rc;
    // (3 mod 4: the text ends behind a separator line — its last record has an empty quality line; DESIGN.md §2, reader decision #3)
    if (lines % 4 != 0 && lines % 4 != 3)
        return fail(SGC_E_ARG, "sgc_sample_push_fastq: the chunk does not hold whole 4-line records (" +
                                   std::to_string(lines) + " lines)");
    return SGC_OK;
}

int sgc_sample_push_fastq_part(sgc_sample *s, const uint8_t *text, uint64_t n_bytes, int where, uint64_t first_line,
                               uint64_t n_newlines, uint64_t *n_records_out) {
    if (!s || (!text && n_bytes)) return fail(SGC_E_ARG, "sgc_sample_push_fastq_part: NULL argument");
    if (n_records_out) *n_records_out = 0;
    if (n_bytes == 0) return SGC_OK;
    if (n_bytes > 0xFFF00000ull) return fail(SGC_E_ARG, "sgc_sample_push_fastq_part: part larger than 4 GiB");
    if (n_newlines != UINT64_MAX && n_newlines > n_bytes) return fail(SGC_E_ARG, "sgc_sample_push_fastq_part: more newlines than bytes");
    uint64_t lines = n_newlines;
    if (n_newlines != UINT64_MAX && where == SGC_MEM_HOST && text[n_bytes - 1] != '\n') lines++;      // the stream's last line
    return push_fastq_part(s, text, n_bytes, where, first_line, n_newlines, lines, n_records_out, nullptr);
}

int sgc_sample_wait_uploads(sgc_sample *s, uint32_t max_pending) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_wait_uploads: NULL");
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (max_pending >= (uint32_t)sgc_ctx::UP_RING) return SGC_OK;     // never more than UP_RING in flight anyway
    if (c->n_up > max_pending) HIP_TRY(hipEventSynchronize(c->ev_up[(c->n_up - 1 - max_pending) % sgc_ctx::UP_RING]));
    return SGC_OK;
}

// marker bytes / newline count reported by the ingest kernels (after a synchronisation of the stream)
static int check_fastq_errors(sgc_sample *s) {
    if (!s->fastq_pushed && !SGC_CHECK) return SGC_OK;
    sgc_ctx *c = s->ctx;
    unsigned long long e[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(e, s->d_err, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (SGC_CHECK && (e[1] >> 8))
        return fail(SGC_E_STATE, "internal bounds check failed in the partitioned pass (flags " + std::to_string(e[1] >> 8) +
                                     ": 1 pool write, 2 block id, 4 block fill, 8 miss-run slot, 16 slice totals)");
    if (e[1] & 1ull) return fail(SGC_E_FORMAT, "FASTQ text: the newline count announced for a part differs from its contents");
    if (e[0]) return fail(SGC_E_FORMAT, "malformed FASTQ record: line " + std::to_string(~0ull - e[0]) +
                                           " does not start with its marker byte ('@' header / '+' separator)");
    return SGC_OK;
}

int sgc_sample_sync(sgc_sample *s) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_sync: NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    { const int rc = flush_batch(s); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    return check_fastq_errors(s);
}

int sgc_sample_flush(sgc_sample *s) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_flush: NULL");
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_batch(s); if (rc) return rc; }
    fold_counts(s);
    HIP_TRY(hipGetLastError());
    return SGC_OK;
}

void *sgc_sample_device_counts(sgc_sample *s) { return s ? (void *)s->d_c64 : nullptr; }

int sgc_sample_export_device(sgc_sample *s, uint64_t *d_out) {
    if (!s || !d_out) return fail(SGC_E_ARG, "sgc_sample_export_device: NULL");
    sgc_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_batch(s); if (rc) return rc; }
    // fold + export in one launch: d_out = counts64 (+= counts32) | total | matched
    if (c->hybrid) sgc_launch_fold_map(c->stream, s->d_c32p, c->d_gid_map, s->d_c64, c->n_packed);
    sgc_launch_export(c->stream, s->d_c32, s->d_c64, s->d_matched, s->total, c->n, (unsigned long long *)d_out);
    HIP_TRY(hipGetLastError());
    s->since_fold = 0;
    return SGC_OK;
}

int sgc_sample_finish(sgc_sample *s, uint64_t *counts, uint64_t *total_reads, uint64_t *matched_reads) {
    if (!s) return fail(SGC_E_ARG, "sgc_sample_finish: NULL");
    sgc_ctx *c = s->ctx;
    int rc = sgc_sample_flush(s);
    if (rc) return rc;
    unsigned long long m = 0;
    if (counts) HIP_TRY(hipMemcpyAsync(counts, s->d_c64, (size_t)c->n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&m, s->d_matched, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (total_reads) *total_reads = s->total;
    if (matched_reads) *matched_reads = m;
    return check_fastq_errors(s);
}

// Host-side self-check of the table builders (no device needed): every guide must be found where the kernels will look.
int sgc_check_host_tables(const uint8_t *seqs, uint32_t n, uint32_t L, int enable_1mm, uint64_t *stats) {
    if (!seqs || !stats) return fail(SGC_E_ARG, "sgc_check_host_tables: NULL argument");
    memset(stats, 0, 4 * sizeof(uint64_t));
    std::string err;
    std::vector<uint64_t> keys;
    sgc_host_table h;
    const uint32_t want_cl = (L >= 4 && L <= SGC_REC8_MAXL) ? (L - 2) / 2 : 0;
    int rc = sgc_build_library_table(seqs, n, L, sgc_choose_log2_slice(n), want_cl, keys, h, err);
    if (rc == SGC_E_UNSUPPORTED && L >= 1 && n >= 1) {
        // byte-string tables (sgc_bytes.h): probe the library table for every guide, the children table for every stored child
        sgc_host_bytes hb;
        rc = sgc_build_bytes_tables(seqs, n, L, enable_1mm != 0, hb, err);
        if (rc != SGC_OK) return fail(rc, "sgc_check_host_tables: " + err);
        const uint32_t lmask = (1u << hb.lib_log2) - 1u;
        for (uint32_t g = 0; g < n; g++) {
            uint64_t hsh = sgc_bytes_hash_init();
            for (uint32_t i = 0; i < L; i++) hsh = sgc_bytes_hash_step(hsh, seqs[(size_t)g * L + i]);
            hsh = sgc_bytes_hash_fin(hsh);
            bool found = false;
            for (uint32_t s = sgc_bytes_slot(hsh, hb.lib_log2); hb.lib_tag[s] != SGC_BYTES_EMPTY && !found; s = (s + 1) & lmask)
                found = hb.lib_tag[s] == hsh && hb.lib_val[s] == g;
            if (!found) return fail(SGC_E_STATE, "sgc_check_host_tables: guide " + std::to_string(g) + " is not reachable in the byte-string table");
        }
        stats[0] = 2; stats[1] = (uint64_t)1 << hb.lib_log2; stats[2] = hb.perm_entries;
        return SGC_OK;
    }
    if (rc != SGC_OK) return fail(rc, "sgc_check_host_tables: " + err);
    stats[0] = 1; stats[1] = (uint64_t)1 << h.log2_slots;
    const uint32_t S = 1u << h.log2_slice;
    // the open-addressed array: scan from the home bucket inside the slice
    for (uint32_t g = 0; g < n; g++) {
        uint32_t b = sgc_home_bucket_ex(keys[g], h.log2_slots, h.log2_slice, h.core_cl);
        bool found = false;
        for (uint32_t step = 0; step < S && !found; step++) {
            for (uint32_t k = 0; k < 2; k++) {
                const uint64_t e = h.slots[2 * (size_t)b + k];
                if (e != SGC_EMPTY && h.gid_bits && (e >> h.gid_bits) == keys[g]) found = (uint32_t)(e & ((1ull << h.gid_bits) - 1ull)) == g;
                if (e != SGC_EMPTY && !h.gid_bits && e == keys[g]) found = h.vals[2 * (size_t)b + k] == g;
            }
            b = sgc_next_bucket(b, h.log2_slice);
        }
        if (!found) return fail(SGC_E_STATE, "sgc_check_host_tables: guide " + std::to_string(g) + " is not reachable in the library table");
    }
    // the two-choice image of the slices (k_count_slices): home slot or its alternate, nowhere else
    std::vector<uint64_t> ck;
    if (h.gid_bits && h.log2_slice <= SGC_LDS_LOG2_SLICE_BIG && sgc_build_slice_cuckoo(h, ck)) {
        stats[3] = 1;
        for (uint32_t g = 0; g < n; g++) {
            const uint32_t hs = sgc_home_slot_ex(keys[g], h.log2_slots, h.log2_slice, h.core_cl);
            const size_t base = (size_t)(hs >> h.log2_slice) * S;
            const uint32_t s1 = hs & (S - 1u), s2 = sgc_cuckoo_alt(keys[g], s1, h.log2_slice);
            const uint64_t want = (keys[g] << h.gid_bits) | g;
            if (ck[base + s1] != want && ck[base + s2] != want)
                return fail(SGC_E_STATE, "sgc_check_host_tables: guide " + std::to_string(g) + " is in neither of its two slots");
        }
        uint64_t occupied = 0;
        for (uint64_t e : ck) occupied += e != SGC_EMPTY;
        if (occupied != n) return fail(SGC_E_STATE, "sgc_check_host_tables: the two-choice image holds " + std::to_string(occupied) + " entries");
    }
    if (enable_1mm) {
        sgc_host_table hp;
        sgc_build_permute_table(keys, L, h, hp, nullptr, nullptr);
        stats[2] = hp.entries;
    }
    return SGC_OK;
}

int sgc_placement_info(sgc_ctx *c, uint64_t *out4) {
    if (!c || !out4) return fail(SGC_E_ARG, "sgc_placement_info: NULL");
    memcpy(out4, c->place_info, sizeof(c->place_info));
    return SGC_OK;
}

int sgc_timing_enable(sgc_ctx *c, int on) {
    if (!c) return fail(SGC_E_ARG, "sgc_timing_enable: NULL");
    c->timing = on != 0;
    return SGC_OK;
}

int sgc_timing_read(sgc_ctx *c, sgc_timing *out, int reset) {
    if (!c || !out) return fail(SGC_E_ARG, "sgc_timing_read: NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    timing_drain(c);
    *out = c->acc;
    if (reset) c->acc = sgc_timing{};
    return SGC_OK;
}

}  // extern "C"
