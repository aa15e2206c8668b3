// sgc_core.hip — single-mismatch resolution in LDS (count path variant 4), gfx950.
//
// k_resolve_miss (sgc_part.hip) answers "is this window one substitution away from exactly one guide?"
// (Permuter, reference src/permutes.rs:63-158) with gathers into a ~100 MB child -> parent table, and every
// such gather leaves the L2.  This file answers the same question inside LDS:
//
//   Span bases [2, L) lie inside all three windows (M = span[0, L), C = span[1, L+1), P = span[2, L+2)).
//   Cut them into two disjoint CORES, A = [2, 2+a) and B = [2+a, L).  A window that is within one
//   substitution of a guide agrees with it on core A or on core B (or both).  So: partition the unresolved
//   records by the hash of their core-A bases; a workgroup stages the guides whose core-A bases (at each of
//   the three alignments) hash to its partition — about a thousand 8-byte entries (core value | alignment,
//   the guide's other bases closed up into 32 bits), bucketed CSR-style — and every record meets, in one
//   short bucket scan, ALL guides that agree with any of its windows on core A.  XOR + popcount of the
//   32-bit "rest" gives the Hamming distance: 0 = exact (src/counter.rs:111), 1 = a parent of the window's
//   child (src/counter.rs:113-116).  What pass A cannot see (a substitution inside core A) pass B sees,
//   the same kernel over core B.
//
//   Uniqueness (src/permutes.rs:127-144: a child with two parents maps to nothing): the parents whose
//   substitution lies outside the pass's core all sit in the bucket and are counted; a parent with its
//   substitution inside the core is invisible, so for a single visible parent one bit of a per-guide mask
//   (amb, built by sgc_build.hip / sgc_tables.cpp: bit 4j+b = "child (j, b) of this guide has another
//   parent") decides.  A window with an 'N' at j needs no mask: it is resolved by the pass whose core does
//   not hold j, where every guide that agrees on all other positions is in the bucket.
//
//   Order (src/counter.rs:111-135): C-exact, C-1mm, P-exact, P-1mm, M-exact, M-1mm.  Pass A takes the first
//   level it can prove; if a higher single-mismatch level is still undecided (nothing seen in A) the
//   record goes on to pass B, which sees everything pass A could not, plus every exact level again.  The
//   rare "1mm found at P/M in pass A, C-1mm undecided" case is closed on the spot with the global tables.
//
// Kernel: k_core (the resolver).  Its input arrives as workgroup-private partitioned runs (sgc_runs.h): the misses of
// k_count_slices and the generic blocks for pass A, what pass A forwards for pass B — no counting or scatter kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "sgc_device.h"
#include "sgc_format.h"
#include "sgc_kernels.h"
#include "sgc_runs.h"

// s_memtime phase stamps + printf (dbg 512): compiled in only with -DSGC_STAMPS=1 (tools/evidence.sh builds such a library for
// profiles/*/stamps.txt); in the shipped kernels they would cost scalar registers the hot loops do not have
#ifndef SGC_STAMPS
#define SGC_STAMPS 0
#endif

#define CP_MAXP RUN_MAXP
#define KC_THREADS 1024u
#define KC_CHUNK 4096u           // records per unit of resolver work
#define KC_GRID 512u
#ifndef KC_ALTPRIO
#define KC_ALTPRIO 1
#endif

// tot[] (records per partition) -> ps[p] = first record of partition p (ps[P] = total) and cs[p] = first
// KC_CHUNK-record chunk of partition p (cs[P] = number of chunks); ps/cs: LDS arrays of P + 1 entries
__device__ __forceinline__ void starts_from_totals(const uint32_t *__restrict__ tot, uint32_t P, uint32_t *ps, uint32_t *cs,
                                                   uint32_t *wsum) {
    const uint32_t t = threadIdx.x;
    const uint32_t n = t < P ? tot[t] : 0;
    uint32_t total, ctotal;
    const uint32_t pre = wg_scan_1024(n, wsum, &total);
    const uint32_t cpre = wg_scan_1024((n + KC_CHUNK - 1) / KC_CHUNK, wsum, &ctotal);
    if (t < P) { ps[t] = pre; if (cs) cs[t] = cpre; }
    if (t == 0) { ps[P] = total; if (cs) cs[P] = ctotal; }
    __syncthreads();
}

#if SGC_STAMPS
static __device__ sgc_tl_row tl_core_a[SGC_TL_MAXWG], tl_core_b[SGC_TL_MAXWG];
#endif
void sgc_core_timeline_dump() {
#if SGC_STAMPS
    SGC_TIMELINE_DUMP(tl_core_a, "coreA"); SGC_TIMELINE_DUMP(tl_core_b, "coreB");
#endif
}

// ------------------------------------------------------------------------------------------------ resolver
// A partition's records (the concatenation of the segments the producers left for it, in.cnt / in.off row p) are cut
// into chunks of KC_CHUNK; workgroup w of KC_GRID takes the w-th equal share of the chunk list of all partitions — a
// run of consecutive chunks, so it stages a new partition's table only when its run crosses into the next partition,
// and a partition swollen by identical reads is shared by as many workgroups as it has chunks.
// FINAL = false (pass A): records with an undecided higher level are written to fwd[] (one run per workgroup, from the
// position of its first input record in the partition-ordered numbering — it forwards at most what it reads), counted
// by pass B's partitions on the way, and laid out as pass B's runs by the epilogue.  FINAL = true (pass B): undecided = no.
// EXACT (with FINAL): the -x mode (src/count.rs:103-107: no Permuter).  Only the exact levels exist, and a guide that equals a
// window agrees with it on BOTH cores, so ONE pass over core A settles everything k_count_slices left: Plus-exact and
// Minus-exact (src/counter.rs:123-130).  A window with an 'N' cannot equal an ACGT guide: invisible.
// LT: the guide length as a compile-time constant (20: what sgRNA libraries almost always are), or 0 = read it from the
// arguments.  With LT the shifts, masks and core geometry of the per-record code are immediates instead of two dozen scalar
// registers — the kernel is capped at 80 (more would halve the occupancy of its 1024-lane workgroups) and was restoring
// spilled scalars with v_readlane all through its record loop.
template <bool FINAL, bool EXACT, int LT>
__global__ void __launch_bounds__(KC_THREADS, 8) __attribute__((amdgpu_num_sgpr(80))) k_core(
    const sgc_runs in, sgc_core_view cv, const ulonglong2 *__restrict__ amb, uint32_t L_arg, sgc_table_view lib, sgc_table_view perm,
    uint64_t *__restrict__ fwd, const sgc_runs out, uint32_t *__restrict__ counts, unsigned long long *__restrict__ matched,
    uint32_t dbg) {
    const uint32_t L = LT ? (uint32_t)LT : L_arg;
    SGC_TIMELINE_BEGIN(dbg);
    if (LT) {
        // the cores are a function of L alone (sgc_api.cpp: A = [2, 2 + (L - 2) / 2), B = the rest of [2, L)); pass B is FINAL && !EXACT
        constexpr uint32_t ca = LT ? ((uint32_t)LT - 2u) / 2u : 0u;
        cv.cs = (FINAL && !EXACT) ? 2u + ca : 2u;
        cv.cl = (FINAL && !EXACT) ? (uint32_t)LT - 2u - ca : ca;
    }
    __shared__ uint64_t ent[SGC_CORE_EMAX];
    __shared__ uint32_t tgid[SGC_CORE_EMAX], cnt[SGC_CORE_EMAX];
    __shared__ uint16_t start[SGC_CORE_STARTS];
    __shared__ uint32_t ps[CP_MAXP + 1], cs_[CP_MAXP + 1], hn[FINAL ? 1 : CP_MAXP], rcur[FINAL ? 1 : CP_MAXP], wtmp[17];
    __shared__ uint32_t segpre[KC_THREADS], segoff[KC_THREADS];      // the segments of the current partition: prefix of counts, starts
    __shared__ uint32_t cmap[2][KC_CHUNK / 64u];
    __shared__ uint32_t n_fwd, rbase;
    __shared__ unsigned long long wsum;
    const uint32_t t = threadIdx.x, P = 1u << cv.log2_p;
    if (t == 0) { n_fwd = 0; wsum = 0; }
    if (!FINAL && t < CP_MAXP) hn[t] = 0;                  // what this workgroup forwards, by the next pass's partition
    starts_from_totals(in.tot, P, ps, cs_, wtmp);
    const uint32_t C = cs_[P];
    const uint32_t c_lo = (uint32_t)((uint64_t)blockIdx.x * C / gridDim.x), c_hi = (uint32_t)((uint64_t)(blockIdx.x + 1) * C / gridDim.x);
    const uint32_t K = L + 2, sh = 2 * K, cs2 = 2 * cv.cs, cl = cv.cl, kdiv = (1u << 20) / K + 1u;
    const uint64_t smask = (1ull << sh) - 1ull, kmask = sgc_key_mask(L), cmask = (1ull << (2 * cl)) - 1ull;
    // window a (0 = M, 1 = C, 2 = P) starts at span base a: its core sits at window positions [ll_a, ll_a + cl)
    const uint32_t ll0 = cv.cs, ll1 = cv.cs - 1, ll2 = cv.cs - 2;
    const uint32_t hs2 = 2 * (cv.cs + cl);                  // span bit where the bases above the core start
    const uint32_t hm = (uint32_t)((1ull << (2 * (L - cl))) - 1ull);   // a rest has L - cl bases
    uint32_t local = 0, cur_p = 0xFFFFFFFFu, wlo = 0;
    unsigned long long ts_unp = 0, ts_scan = 0, ts_dec = 0, ts_out = 0, ts_begin = __builtin_amdgcn_s_memtime();
    // phases of the timeline rows (dbg 1048576; thread 0's view): staging of partitions | chunk prologues (segment search, loads issued,
    // the barrier of the chunk) | record iterations | epilogue
    unsigned long long tp_stage = 0, tp_chunk = 0, tp_iter = 0, tp_x = ts_begin;
#define KC_PHASE(acc) if (SGC_STAMPS && (dbg & 1048576u)) { const unsigned long long x_ = __builtin_amdgcn_s_memtime(); acc += x_ - tp_x; tp_x = x_; }
    uint32_t n_it = 0;
    for (uint32_t ch = c_lo; ch < c_hi; ch++) {
#if KC_ALTPRIO
        // The two workgroups of a CU are not served alike: the issue arbiter prefers the older one, which then finishes a quarter of the
        // kernel's time before the younger (the timelines of round 3) and leaves it to run alone, at 0.75 of the pair's rate.  Taking
        // turns at the higher priority, chunk by chunk and in opposite phase, lets both advance alike: pass B 0.050 -> 0.046 ms, pass A
        // 0.197 -> 0.196 (per record iteration instead: pass A 0.200; the same in k_partition / k_count_slices: lost in their process-to-process spread).
        if ((((ch - c_lo) & 1u) ^ (blockIdx.x >= gridDim.x / 2u ? 1u : 0u)) != 0u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        const uint32_t p = find_extent<SGC_CORE_MAX_LOG2_P>(cs_, P, ch);
        if (p != cur_p) {                                    // uniform over the workgroup
            __syncthreads();
            if (cur_p != 0xFFFFFFFFu && !SGC_DBG(dbg, 8192u))
                for (uint32_t i = t; i < SGC_CORE_EMAX; i += KC_THREADS) {
                    const uint32_t v = cnt[i];
                    if (v) atomicAdd(&counts[tgid[i] & 0x7FFFFFFFu], v);
                }
            __syncthreads();
            const uint64_t *ge = cv.ents + (size_t)p * SGC_CORE_EMAX;
            const uint32_t *gg = cv.gids + (size_t)p * SGC_CORE_EMAX;
            const uint32_t *gs = reinterpret_cast<const uint32_t *>(cv.starts + (size_t)p * SGC_CORE_STARTS);
            // (every load of the staging first, into registers, then the stores: written load, store, load, store the loops run in
            // that order, each iteration a round trip to memory of its own)
            static_assert(SGC_CORE_EMAX == 2 * KC_THREADS && SGC_CORE_STARTS / 2 <= 3 * KC_THREADS, "the staging below is unrolled by hand");
            const uint64_t e0 = ge[t], e1 = ge[t + KC_THREADS];
            const uint32_t g0 = gg[t], g1 = gg[t + KC_THREADS];
            const uint32_t s0 = gs[t], s1 = gs[t + KC_THREADS], s2 = t + 2 * KC_THREADS < SGC_CORE_STARTS / 2 ? gs[t + 2 * KC_THREADS] : 0u;
            const uint32_t sc = t < in.W ? in.cnt[(size_t)p * in.W + t] : 0u;
            const uint32_t so = t < in.W ? in.off[(size_t)p * in.W + t] : 0u;
            ent[t] = e0; ent[t + KC_THREADS] = e1; tgid[t] = g0; tgid[t + KC_THREADS] = g1; cnt[t] = 0; cnt[t + KC_THREADS] = 0;
            reinterpret_cast<uint32_t *>(start)[t] = s0; reinterpret_cast<uint32_t *>(start)[t + KC_THREADS] = s1;
            if (t + 2 * KC_THREADS < SGC_CORE_STARTS / 2) reinterpret_cast<uint32_t *>(start)[t + 2 * KC_THREADS] = s2;
            // the partition's segments, one per producer workgroup (in.W <= KC_THREADS)
            uint32_t seg_total;
            const uint32_t sp = wg_scan_1024(sc, wtmp, &seg_total);
            segpre[t] = sp;
            segoff[t] = so;
            __syncthreads();
            cur_p = p;
            KC_PHASE(tp_stage)
        }
        // [lo, hi): the chunk in the partition-ordered numbering of all records; x0: its first record inside the partition
        const uint32_t x0 = (ch - cs_[p]) * KC_CHUNK;
        const uint32_t lo = ps[p] + x0, hi = lo + KC_CHUNK < ps[p + 1] ? lo + KC_CHUNK : ps[p + 1];
        if (ch == c_lo) wlo = lo;
        // the chunk's records first (KC_CHUNK / KC_THREADS loads in flight), then one at a time
        uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        static_assert(KC_CHUNK == 4 * KC_THREADS, "the register queue below holds four records");
        {
            // which segment holds record x of the partition: one search per 64 records (lane i of the first wave, for
            // record x0 + 64 i), then every lane walks on from there — segments are ~100 records long
            const uint32_t m = hi - lo;
            uint32_t *cm = cmap[(ch - c_lo) & 1u];
            if (t < KC_CHUNK / 64u) cm[t] = find_extent<10>(segpre, in.W, x0 + 64u * t < x0 + m ? x0 + 64u * t : x0 + m - 1u);
            __syncthreads();
            uint32_t x, w;
#define KC_LOAD(dst, k)                                                                                         \
            x = (k) * KC_THREADS + t;                                                                             \
            if (x < m) {                                                                                          \
                w = cm[x >> 6];                                                                                   \
                while (w + 1u < in.W && segpre[w + 1u] <= x0 + x) w++;                                           \
                dst = __builtin_nontemporal_load(&in.recs[segoff[w] + (x0 + x - segpre[w])]);                    \
            }
            KC_LOAD(r0, 0u) KC_LOAD(r1, 1u) KC_LOAD(r2, 2u) KC_LOAD(r3, 3u)
#undef KC_LOAD
        }
        KC_PHASE(tp_chunk)
#pragma unroll 1
        for (uint32_t i0 = lo; i0 < hi; i0 += KC_THREADS) {
            const bool valid = i0 + t < hi;
            unsigned long long tsa = 0;
            if (SGC_STAMPS && (dbg & 512)) { tsa = __builtin_amdgcn_s_memtime(); n_it++; }
            const uint64_t rec = r0;
            r0 = r1; r1 = r2; r2 = r3;
            // Most waves hold nothing but records without a status (the misses of k_count_slices: every window alive, no 'N'): for them the
            // per-window state (alive / 'N' position / undecidable) is a constant, and the body below is compiled a second time with it folded
            // away — the kernel is bound by vector issue.  The records with a status (the generic share: an 'N', a dead
            // window) sit together at the end of a partition, so few waves are mixed; a mixed wave takes the general body.
            auto body = [&](auto clean_tag) {
            constexpr bool CLEAN = decltype(clean_tag)::value;
            const uint64_t span = rec & smask;
            const uint32_t status = (uint32_t)(rec >> sh);
            const uint32_t corev = (uint32_t)((span >> cs2) & cmask);
            // the rests of the three windows: bases below the core | bases above it
            const uint32_t hi_bits = (uint32_t)(span >> hs2);
            const uint32_t R0 = ((uint32_t)span & ((1u << (2 * ll0)) - 1u)) | ((hi_bits << (2 * ll0)) & hm);
            const uint32_t R1 = ((uint32_t)(span >> 2) & ((1u << (2 * ll1)) - 1u)) | ((hi_bits << (2 * ll1)) & hm);
            const uint32_t R2 = ((uint32_t)(span >> 4) & ((1u << (2 * ll2)) - 1u)) | (hi_bits << (2 * ll2));
            // vis bit a: window a takes part in this pass; unk bit a: its single-mismatch level cannot be decided
            // here ('N' inside the core); nr_a: the 'N' position as a bit of the rest (0 = clean)
            uint32_t vis = valid ? 7u : 0u, unk = 0, nr0 = 0, nr1 = 0, nr2 = 0, st0 = 0, st1 = 0, st2 = 0;
            if (!CLEAN && status) {
                // status = sC + K (sP + K sM); n / K as a multiply-shift (exact for n < K^3 <= 2^14, K <= 25: checked
                // for every such n by tests/test_abi_cpu.py::test_status_division_magic)
                const uint32_t q1 = (status * kdiv) >> 20;
                st0 = (q1 * kdiv) >> 20; st1 = status - q1 * K; st2 = q1 - st0 * K;
                if (st0 == SGC_STATE_DEAD) vis &= ~1u;
                if (st1 == SGC_STATE_DEAD) vis &= ~2u;
                if (st2 == SGC_STATE_DEAD) vis &= ~4u;
                if (EXACT) {
                    if (st0 >= 2) vis &= ~1u;
                    if (st1 >= 2) vis &= ~2u;
                    if (st2 >= 2) vis &= ~4u;
                } else {
                    if (st0 >= 2) { const uint32_t j = st0 - 2; if (j >= ll0 && j < ll0 + cl) { vis &= ~1u; unk |= 1u; } else nr0 = 1u << (2 * (j < ll0 ? j : j - cl)); }
                    if (st1 >= 2) { const uint32_t j = st1 - 2; if (j >= ll1 && j < ll1 + cl) { vis &= ~2u; unk |= 2u; } else nr1 = 1u << (2 * (j < ll1 ? j : j - cl)); }
                    if (st2 >= 2) { const uint32_t j = st2 - 2; if (j >= ll2 && j < ll2 + cl) { vis &= ~4u; unk |= 4u; } else nr2 = 1u << (2 * (j < ll2 ? j : j - cl)); }
                }
            }
            if (SGC_STAMPS && (dbg & 512)) { const unsigned long long x = __builtin_amdgcn_s_memtime(); ts_unp += x - tsa; tsa = x; }
            uint32_t ex0 = SGC_NONE, ex1 = SGC_NONE, ex2 = SGC_NONE;      // entry of the exact guide
            uint32_t cc = 0;                                              // visible distance-1 guides, 8 bits per window
            uint32_t k0 = 0, k1 = 0, k2 = 0;                              // entry of the (last) one
            uint32_t d0 = 0, d1 = 0, d2 = 0;                              // its differing base (one bit of the rest)
            if (vis && !SGC_DBG(dbg, 1024u)) {
                const uint32_t b = sgc_core_home(sgc_core_hash(corev, cl), cv.log2_p);
                const uint32_t e_end = start[b + 1];
                for (uint32_t i = start[b]; i < e_end; i++) {
                    const uint64_t e = ent[i];
                    if (((uint32_t)e & 0x3FFFFFFFu) != corev) continue;          // another core in the same bucket
                    const uint32_t a = (uint32_t)e >> 30;
                    if (!CLEAN && !((vis >> a) & 1u)) continue;          // (a clean record inside this loop is valid: all three windows take part)
                    const uint32_t Ra = a == 0 ? R0 : (a == 1 ? R1 : R2), nr = a == 0 ? nr0 : (a == 1 ? nr1 : nr2);
                    const uint32_t x = Ra ^ (uint32_t)(e >> 32);
                    const uint32_t dm = ((x | (x >> 1)) & 0x55555555u) & ~nr;
                    const uint32_t d = (uint32_t)__popc(dm) + (nr ? 1u : 0u);
                    if (d == 0) { if (a == 0) ex0 = i; else if (a == 1) ex1 = i; else ex2 = i; }
                    else if (!EXACT && d == 1) {
                        cc += 1u << (8 * a);
                        if (a == 0) { k0 = i; d0 = dm; } else if (a == 1) { k1 = i; d1 = dm; } else { k2 = i; d2 = dm; }
                    }
                }
            }
            if (SGC_STAMPS && (dbg & 512)) { const unsigned long long x = __builtin_amdgcn_s_memtime(); ts_scan += x - tsa; tsa = x; }
            // Levels in the reference's order, as bits: 0 C-exact, 1 C-1mm, 2 P-exact, 3 P-1mm, 4 M-exact, 5 M-1mm.
            // lm: the level holds on what this pass sees (1mm: exactly one visible parent; invisible windows
            // found nothing).  um: the 1mm level cannot be decided here (a clean visible window with no
            // candidate — the substitution may sit inside the core — or an 'N' inside the core).
            const uint32_t cC = (cc >> 8) & 255u, cP = (cc >> 16) & 255u, cM = cc & 255u;
            uint32_t lm = (ex1 != SGC_NONE ? 1u : 0u) | (!EXACT && cC == 1 ? 2u : 0u) | (ex2 != SGC_NONE ? 4u : 0u) | (!EXACT && cP == 1 ? 8u : 0u) |
                          (ex0 != SGC_NONE ? 16u : 0u) | (!EXACT && cM == 1 ? 32u : 0u);
            const uint32_t um = EXACT ? 0u :
                                (((CLEAN ? valid : ((vis >> 1) & 1u) != 0) && cC == 0 && !nr1) || (unk & 2u) ? 2u : 0u) |
                                (((CLEAN ? valid : ((vis >> 2) & 1u) != 0) && cP == 0 && !nr2) || (unk & 4u) ? 8u : 0u) |
                                (((CLEAN ? valid : (vis & 1u) != 0) && cM == 0 && !nr0) || (unk & 1u) ? 32u : 0u);
            uint32_t lvl = 6, res = SGC_NONE;
            while (lm) {
                lvl = (uint32_t)__builtin_ctz(lm);
                if (!(lvl & 1u)) { res = lvl == 0 ? ex1 : (lvl == 2 ? ex2 : ex0); break; }
                // one visible parent: it is THE parent unless the child has another one inside the core (amb mask);
                // an 'N' window saw all of its candidates
                const uint32_t k = lvl == 1 ? k1 : (lvl == 3 ? k2 : k0), nr = lvl == 1 ? nr1 : (lvl == 3 ? nr2 : nr0);
                res = k;
                if (nr) break;
                const uint32_t dm = lvl == 1 ? d1 : (lvl == 3 ? d2 : d0), Ra = lvl == 1 ? R1 : (lvl == 3 ? R2 : R0);
                const uint32_t ll = lvl == 1 ? ll1 : (lvl == 3 ? ll2 : ll0);
                const uint32_t jr = (uint32_t)__builtin_ctz(dm) >> 1, j = jr < ll ? jr : jr + cl;
                const uint32_t bit = 4 * j + ((Ra >> (2 * jr)) & 3u);
                // bit 31 of the staged guide id: "some child of this guide has another parent" (sgc_flag_ambiguous: a couple of
                // hundred of 100k guides) — only then is the guide's mask worth a gather from global memory
                const uint32_t tg = tgid[k];
                if (!(tg >> 31) || SGC_DBG(dbg, 2048u)) break;
                const ulonglong2 mk = amb[tg & 0x7FFFFFFFu];
                if (!(((bit < 64 ? mk.x : mk.y) >> (bit & 63)) & 1ull)) break;
                lm &= lm - 1u;                   // ambiguous child: this level fails, on to the next
                lvl = 6; res = SGC_NONE;
            }
            uint32_t unk_above = um & ((1u << lvl) - 1u);
            if (!FINAL && unk_above && cv.filt && !SGC_DBG(dbg, 16384u)) {
                // a clean window that saw no candidate here: a clear bit of the rest filter proves that no parent hides
                // inside the core either ('N' inside the core — unk — stays undecided)
                const uint32_t cand = unk_above & ~(((unk & 2u) ? 2u : 0u) | ((unk & 4u) ? 8u : 0u) | ((unk & 1u) ? 32u : 0u));
                const uint32_t fw = 1u << (cv.filt_log2 - 5);
                // the (up to) three words are requested together — one round trip to the L2, not three in a row; a lane
                // without that candidate reads word 0
                const uint32_t x1 = sgc_rest_hash(R1, cv.filt_log2), x2 = sgc_rest_hash(R2, cv.filt_log2), x0 = sgc_rest_hash(R0, cv.filt_log2);
                const uint32_t w1 = cv.filt[(cand & 2u) ? fw + (x1 >> 5) : 0u];
                const uint32_t w2 = cv.filt[(cand & 8u) ? 2u * fw + (x2 >> 5) : 0u];
                const uint32_t w0 = cv.filt[(cand & 32u) ? (x0 >> 5) : 0u];
                if ((cand & 2u) && !((w1 >> (x1 & 31u)) & 1u)) unk_above &= ~2u;
                if ((cand & 8u) && !((w2 >> (x2 & 31u)) & 1u)) unk_above &= ~8u;
                if ((cand & 32u) && !((w0 >> (x0 & 31u)) & 1u)) unk_above &= ~32u;
            }
            if (SGC_STAMPS && (dbg & 512)) { const unsigned long long x = __builtin_amdgcn_s_memtime(); ts_dec += x - tsa; tsa = x; }
            bool fwd_it = false;
            if (valid) {
                if (FINAL || unk_above == 0) {
                    if (res != SGC_NONE) { atomicAdd(&cnt[res], 1u); local++; }
                } else if (!(lvl & 1u)) {
                    fwd_it = true;          // exact at P/M or nothing yet: pass B re-finds it and decides the levels above
                } else {
                    // single mismatch found at P or M while a higher single-mismatch level is undecided: settle those
                    // with the global tables (rare)
                    uint32_t g2 = SGC_NONE;
#pragma unroll
                    for (uint32_t w = 0; w < 2; w++) {
                        const uint32_t a = w == 0 ? 1u : 2u;
                        if (g2 != SGC_NONE || !(unk_above & (2u << (2 * w)))) continue;
                        const uint32_t sta = a == 1 ? st1 : st2;
                        const uint64_t Wa = (span >> (2 * a)) & kmask;
                        if (sta == SGC_STATE_CLEAN) g2 = table_find<true>(perm, Wa);
                        else {
                            uint32_t hit = SGC_NONE, nh = 0;
                            for (uint64_t bb = 0; bb < 4; bb++) {
                                const uint32_t x = table_find<true>(lib, Wa | (bb << (2 * (sta - 2u))));
                                if (x != SGC_NONE) { hit = x; nh++; }
                            }
                            if (nh == 1) g2 = hit;
                        }
                    }
                    if (g2 != SGC_NONE) atomicAdd(&counts[g2], 1u); else atomicAdd(&cnt[res], 1u);
                    local++;
                }
            }
            if (!FINAL) {
                const uint64_t bal = __ballot(fwd_it);
                if (bal) {
                    const uint32_t lane = t & 63u;
                    uint32_t b0 = 0;
                    if (lane == (uint32_t)__builtin_ctzll(bal)) b0 = atomicAdd(&n_fwd, (uint32_t)__popcll(bal));
                    b0 = __shfl(b0, __builtin_ctzll(bal), 64);
                    if (fwd_it && !SGC_DBG(dbg, 4096u)) {
                        fwd[(uint64_t)wlo + b0 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = rec;
                        const uint32_t q = run_part(out, rec);
                        if (q != RUN_DROP) atomicAdd(&hn[q], 1u);
                    }
                }
            }
            if (SGC_STAMPS && (dbg & 512)) { const unsigned long long x = __builtin_amdgcn_s_memtime(); ts_out += x - tsa; }
            };
            // (timing-only ablations, -DSGC_ABLATE=1: dbg 32768 the records are loaded and dropped — the kernel's skeleton: staging, chunk
            // prologues, loads, epilogue; 16384 no rest filter; 1024 no bucket scan; 2048 no ambiguity masks; 4096 no forwards; 8192 no flush)
            if (SGC_DBG(dbg, 32768u)) { if (rec == 0x123456789ABCDEFull) local++; continue; }
            if (__ballot((uint32_t)(rec >> sh) != 0) == 0ull) body(std::true_type{}); else body(std::false_type{});
        }
        KC_PHASE(tp_iter)
    }
    if ((SGC_STAMPS && (dbg & 512)) && (t & 63) == 0 && (blockIdx.x % 101) == 0 && (t >> 6) < 2)
        printf("k_core<%d> wg %u wave %u: %u iters, load+unpack %llu scan %llu decide %llu out %llu total %llu ticks\n", (int)FINAL,
               blockIdx.x, t >> 6, n_it, ts_unp, ts_scan, ts_dec, ts_out, (unsigned long long)(__builtin_amdgcn_s_memtime() - ts_begin));
    __syncthreads();
    if (cur_p != 0xFFFFFFFFu && !SGC_DBG(dbg, 8192u))
        for (uint32_t i = t; i < SGC_CORE_EMAX; i += KC_THREADS) {
            const uint32_t v = cnt[i];
            if (v) atomicAdd(&counts[tgid[i] & 0x7FFFFFFFu], v);
        }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((t & 63) == 0 && local) atomicAdd(&wsum, (unsigned long long)local);
    __syncthreads();
    if (t == 0 && wsum) atomicAdd(matched, wsum);
    if (!FINAL) {
        // epilogue (sgc_runs.h): the forwarded run of this workgroup, laid out by pass B's partitions in a region of its own
        __syncthreads();
        run_reserve(out, blockIdx.x, hn, rcur, wtmp, &rbase);
        const uint32_t nf = n_fwd;
        for (uint32_t j0 = 0; j0 < nf && !SGC_DBG(dbg, 4096u | 524288u); j0 += 4 * KC_THREADS) {
            uint64_t r[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) { const uint32_t j = j0 + k * KC_THREADS + t; if (j < nf) r[k] = fwd[(uint64_t)wlo + j]; }
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t j = j0 + k * KC_THREADS + t;
                if (j < nf) run_place(out, rcur, r[k]);
            }
        }
    }
    SGC_TIMELINE_END4(dbg, (FINAL ? tl_core_b : tl_core_a), c_hi - c_lo, tp_stage, tp_chunk, tp_iter, __builtin_amdgcn_s_memtime() - tp_x);
}

// ------------------------------------------------------------------------------------------------ host side
// Scratch of the two passes: runs_a_bytes (pass A's runs) and fwd_bytes (pass A's forwarded runs) — pass B's runs go to
// the slice pool, which is dead once k_count_slices has handed its leftovers on —, small_bytes for the four
// [P][W] matrices, zero_bytes of zeroed words (totals A | totals B | region cursors), which live in the zeroed tail of
// the descriptor buffer.
void sgc_core_plan(uint64_t n, const sgc_core_view &a, const sgc_core_view &b, uint32_t producers_a, sgc_core_geometry *g) {
    g->w = producers_a;
    g->grid_a = g->grid_b = KC_GRID;
    g->runs_a_bytes = n * 8;
    g->fwd_bytes = n * 8;
    g->zero_bytes = (size_t)(2 * CP_MAXP + 2) * 4;
    static_assert((2 * CP_MAXP + 2) * 4 <= SGC_DESC_TAIL, "the zeroed tail of the descriptor buffer holds the counters");
    g->mat_a = ((size_t)g->w << a.log2_p) * 4;                 // bytes of ONE matrix of pass A's runs
    g->mat_b = ((size_t)KC_GRID << b.log2_p) * 4;
    g->small_bytes = 2 * g->mat_a + 2 * g->mat_b;
}

static sgc_runs make_runs(uint64_t *recs, void *mats, size_t mat_bytes, uint32_t *tot, uint32_t *cursor, uint32_t W,
                          const sgc_core_view &cv, uint32_t L) {
    const uint32_t K = L + 2;
    sgc_runs r;
    r.recs = recs; r.cnt = (uint32_t *)mats; r.off = (uint32_t *)((char *)mats + mat_bytes); r.tot = tot; r.cursor = cursor; r.W = W;
    r.sub_bits = 0xFFu;
    r.cs2 = 2 * cv.cs; r.log2_p = cv.log2_p; r.sh = 2 * K; r.dead_all = SGC_STATE_DEAD * (1 + K + K * K);
    r.cmask = (1ull << (2 * cv.cl)) - 1ull; r.cl = cv.cl;
    return r;
}

// what k_count_slices' epilogue fills: the runs of pass A
sgc_runs sgc_core_runs_a(const sgc_core_geometry &g, const sgc_core_view &ca, uint32_t L, uint64_t *buf0, void *zeroed, void *small) {
    uint32_t *z = (uint32_t *)zeroed;
    return make_runs(buf0, small, g.mat_a, z, z + 2 * CP_MAXP, g.w, ca, L);
}

void sgc_launch_core(hipStream_t st, int pass, uint32_t L, const sgc_table_view &lib, const sgc_table_view &perm,
                     const sgc_core_view &ca, const sgc_core_view &cb, const uint64_t *amb, const sgc_core_geometry &g,
                     uint64_t *buf0, uint64_t *buf1, uint64_t *buf2, void *zeroed, void *small, uint32_t *counts,
                     unsigned long long *matched, uint32_t dbg) {
    uint32_t *z = (uint32_t *)zeroed;
    const ulonglong2 *am = reinterpret_cast<const ulonglong2 *>(amb);
    const sgc_runs ra = sgc_core_runs_a(g, ca, L, buf0, zeroed, small);
    const sgc_runs rb = make_runs(buf2, (char *)small + 2 * g.mat_a, g.mat_b, z + CP_MAXP, z + 2 * CP_MAXP + 1, KC_GRID, cb, L);
    // the specialisation for L = 20 needs the cores where it expects them (they are: sgc_set_library cuts them that way)
    const bool l20 = L == 20 && ca.cs == 2 && ca.cl == 9 && cb.cs == 11 && cb.cl == 9;
#define KC_LAUNCH(FINAL, EXACT, LT, IN, CV, FWD, OUT)                                                                        \
    hipLaunchKernelGGL((k_core<FINAL, EXACT, LT>), dim3(KC_GRID), dim3(KC_THREADS), sgc_extra_lds("CORE"), st, IN, CV, am, L, lib, perm, FWD, OUT, counts, matched, dbg)
    if (pass == 2) {    // -x: the one exact-only pass over the runs k_count_slices left in buf0
        if (l20) KC_LAUNCH(true, true, 20, ra, ca, (uint64_t *)nullptr, ra); else KC_LAUNCH(true, true, 0, ra, ca, (uint64_t *)nullptr, ra);
    } else if (pass == 0) { // pass A: the runs k_count_slices left in buf0; forwards go to buf1 run by run, then to buf2 as pass B's runs
        if (l20) KC_LAUNCH(false, false, 20, ra, ca, buf1, rb); else KC_LAUNCH(false, false, 0, ra, ca, buf1, rb);
    } else {            // pass B: what pass A forwarded
        if (l20) KC_LAUNCH(true, false, 20, rb, cb, (uint64_t *)nullptr, rb); else KC_LAUNCH(true, false, 0, rb, cb, (uint64_t *)nullptr, rb);
    }
#undef KC_LAUNCH
}

// diagnostic (sgc_set_option "print_occupancy"): resident workgroups per CU as the runtime computes them
void sgc_core_print_occupancy() {
    int a = -1, b = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_core<false, false, 20>, KC_THREADS, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_core<true, false, 20>, KC_THREADS, 0);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_core<false, false, 20>));
    fprintf(stderr, "occupancy (workgroups/CU): k_core<A> %d k_core<B> %d; k_core<A> lds %zu regs %d\n", a, b, fa.sharedSizeBytes, fa.numRegs);
}
