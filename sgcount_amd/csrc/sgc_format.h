// sgc_format.h — packed-record and key formats shared by host code and gfx950 kernels.
//
// A *record* is everything Counter::assign (reference src/counter.rs:96-140) needs from one read:
// the L+2 oriented bases covering the Minus / Centered / Plus windows, and one status code that
// folds in the bounds rule (src/counter.rs:158-180) and the non-ACGT bytes.
//
//   span  = oriented bases [o-1, o+L+1)      (oriented = forward read, or its reverse complement)
//   M = span[0..L)   C = span[1..L+1)   P = span[2..L+2)
//   base code: A=0 C=1 G=2 T=3, 2 bits each, span base w at bits [2w, 2w+2); non-ACGT bases store 0
//
//   per-window state s in [0, L+2):  0 = clean (all ACGT, in bounds)
//                                    1 = dead  (cannot match: out of bounds, >=2 non-ACGT bytes, or a
//                                               single non-ACGT byte that is not 'N')
//                                    2+j = exactly one non-ACGT byte, an 'N', at window position j
//   status = sC + K*(sP + K*sM),  K = L+2
//
// Bounds: a window that fails Counter::bounds makes the reference return None for the whole read
// (src/counter.rs:105-108), which is the same as that window and every later stage being dead, so
// the packer marks them dead (Centered fails => all three; Plus fails => Plus and Minus).
//
//   rec8  (L <= 23): one u64 = span bits [0, 2K) | status << 2K     (2K + 14 <= 64 since (L+2)^3 <= 2^14)
//   rec16 (L <= 30): u64 span, u64 status
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SGC_HD __host__ __device__ __forceinline__
#else
#define SGC_HD inline
#endif

#define SGC_REC8_MAXL 23
#define SGC_MAXL 30
#define SGC_STATE_CLEAN 0u
#define SGC_STATE_DEAD 1u
#define SGC_NONE 0xFFFFFFFFu
#define SGC_EMPTY 0xFFFFFFFFFFFFFFFFull

// byte -> code: 0..3 = ACGT, 4 = 'N', 5 = anything else
SGC_HD uint32_t sgc_base_code(uint8_t c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        case 'N': return 4; default: return 5;
    }
}

// Reverse strand: the reference slices Record::seq_rev_comp() (src/counter.rs:196-204).  fxread's
// complement is not pinned by any upstream test; it is restated (oracle/sgcount_oracle.c ctr_trim) as
// the byte trick `c & 2 ? c ^ 4 : c ^ 21` (A<->T, C<->G; 'N' -> 'J', and 'J' -> 'N'), and the
// complemented byte is then classified exactly like a forward byte.
SGC_HD uint32_t sgc_base_code_rc(uint8_t c) {
    return sgc_base_code((c & 2) ? (uint8_t)(c ^ 4) : (uint8_t)(c ^ 21));
}

SGC_HD uint64_t sgc_key_mask(uint32_t L) { return L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1ull); }

SGC_HD uint64_t sgc_hash(uint64_t key) {
    // multiplicative (Fibonacci) hash with a fold so that low key bits reach the top
    key ^= key >> 29;
    return key * 0x9E3779B97F4A7C15ull;
}

// Open-addressed table of 2-slot BUCKETS (16 bytes, read with one 16-byte load).  A key's home bucket is
// the top (log2_slots - 1) bits of its hash; an insert scans buckets from there and takes the first free
// slot, so a lookup scans the same buckets and stops at the first match or the first free slot.  The slot
// array is cut into 2^(log2_slots - log2_slice) *slices*; probing wraps INSIDE the home slice, so a slice
// is a self-contained hash table (the partitioned count path stages one slice per workgroup in LDS, the
// global paths probe the very same array).  An unpartitioned table has log2_slice == log2_slots.
struct sgc_table_view {
    const uint64_t *slots;   // packed: (key << gid_bits) | gid, SGC_EMPTY if free; split: key or SGC_EMPTY
    const uint32_t *vals;    // split layout only: gid per slot
    uint32_t log2_slots;
    uint32_t gid_bits;       // 0 => split layout
    uint32_t log2_slice;
    uint32_t core_cl;        // 0: the slice is the top of the full-key hash.  > 0: the slice is picked by the hash of key bases
                             // [1, 1 + core_cl) alone (sgc_home_bucket_ex): the same bases, hashed the same way, pick the
                             // partition of core pass A, so a slice's leftovers fall into a handful of that pass's partitions
};

SGC_HD uint32_t sgc_home_bucket(uint64_t key, uint32_t log2_slots) { return (uint32_t)(sgc_hash(key) >> (65 - log2_slots)); }
SGC_HD uint32_t sgc_next_bucket(uint32_t b, uint32_t log2_slice) {
    const uint32_t m = (1u << (log2_slice - 1)) - 1u;
    return (b & ~m) | ((b + 1) & m);
}
// slice (= partition of the partitioned count path) a key belongs to
SGC_HD uint32_t sgc_slice_of(uint64_t key, uint32_t log2_slots, uint32_t log2_slice) {
    return sgc_home_bucket(key, log2_slots) >> (log2_slice - 1);
}

// Blocked Bloom filter: one 64-bit word per key, two bits in it.  A clear bit proves the key is absent;
// the count path uses it to skip probes that would miss (most Plus/Minus/permute probes do).
SGC_HD uint64_t sgc_hash2(uint64_t key) {
    key ^= key >> 31;
    return key * 0xD6E8FEB86659FD93ull;
}
SGC_HD uint32_t sgc_bloom_word(uint64_t h2, uint32_t log2_words) { return (uint32_t)(h2 >> (64 - log2_words)); }
SGC_HD uint64_t sgc_bloom_mask(uint64_t h2) { return (1ull << (h2 & 63)) | (1ull << ((h2 >> 6) & 63)); }

// Core index (sgc_core.hip): the guides seen through one *core* — span bases [cs, cs + cl), a stretch
// that lies inside all three windows — at each of the three window alignments (0 = M, 1 = C, 2 = P; the
// window starts at span base a, so the core sits at window positions [cs - a, cs - a + cl)).
// 2^log2_p partitions by the hash of the core value; inside a partition the entries are bucketed (CSR) by
// further hash bits: bucket b holds entries [starts[b], starts[b + 1]).  An entry is one u64:
//   low 32 bits  = core value | alignment << 30
//   high 32 bits = the REST of the guide: its bases outside the core, closed up (<= 2 * 14 bits)
// and gids[] runs parallel to the entries.
#define SGC_CORE_LOG2_S 12u                                  // buckets per partition (log2)
#define SGC_CORE_EMAX 2048u                                  // entry capacity of a partition
#define SGC_CORE_STARTS ((1u << SGC_CORE_LOG2_S) + 2u)       // u16 per partition in starts[] (NB + 1, padded)
#define SGC_CORE_MAX_LOG2_P 9u                              // 512 partitions x 2048 entries: ~340k guides
struct sgc_core_view {
    const uint64_t *ents;
    const uint32_t *gids;
    const uint16_t *starts;
    uint32_t log2_p, cs, cl, filt_log2;
    // Rest filter (optional, core A only): three bit sets of 2^filt_log2 bits, one per alignment; bit
    // sgc_rest_hash(rest) of set a is on iff some guide has that rest at alignment a.  A clean window that found no
    // candidate through this core can only have a single-mismatch parent whose substitution lies INSIDE the core, and
    // such a parent agrees with the window on the whole rest: a clear bit proves there is none, so the pass can settle
    // the level instead of forwarding the record to the other core's pass (false positives only cost a forward).
    const uint32_t *filt;
};
SGC_HD uint32_t sgc_rest_hash(uint32_t rest, uint32_t log2_bits) { return (rest * 0x9E3779B1u) >> (32 - log2_bits); }
// Hash of a core value (cl bases = 2 cl bits, cl <= 14): its top 2 cl bits are a BIJECTION of the core value (an odd multiplier
// modulo 2^(2 cl): sgc_core_mix / sgc_core_unmix), the bits below are zero.  The library slice, the partition of a core pass and
// the bucket inside the partition are successive prefixes of it, so (a) a bucket of a core index holds the entries of as few
// distinct core values as the bits allow — for L = 20 exactly one: a bucket scan meets only true candidates — and (b) a record
// that sits in a slice's block need not store the slice bits: the five-byte slice records of k_partition keep the remaining
// 2 cl - (slice bits) bits of the mixed value in place of the core bases and k_count_slices rebuilds the core value with one
// multiply.  One multiply to evaluate (the partition kernel does it for every read, and 32-bit multiplies run at quarter rate).
// low 32 bits of (a mod 2^24) * (b mod 2^24): v_mul_u32_u24 on the device — full rate, where a 32-bit multiply runs at a quarter
SGC_HD uint32_t sgc_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
// the same with a compile-time multiplier, as an instruction the compiler cannot "improve": it folds a later shift or mask into the
// constant of a known-small product, the constant outgrows 24 bits and the multiply falls back to the quarter-rate 32-bit one
template <uint32_t C> SGC_HD uint32_t sgc_mul24c(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %2, %1" : "=v"(r) : "v"(a), "n"(C & 0xFFFFFFu));
    return r;
#else
    return (a & 0xFFFFFFu) * (C & 0xFFFFFFu);
#endif
}
#define SGC_CORE_M1 0x9E3779B1u
#define SGC_CORE_M1_INV 0x0E8B2F51u      // SGC_CORE_M1 * SGC_CORE_M1_INV == 1 (mod 2^32), hence modulo every 2^k
// (the low 2 cl bits of a product depend only on the low 2 cl bits of its factors: for 2 cl <= 24 — every one-word record,
// L <= 23 — the 24-bit multiply gives the same bits)
SGC_HD uint32_t sgc_core_mix(uint32_t corev, uint32_t cl) {
    const uint32_t m = (uint32_t)((1ull << (2 * cl)) - 1ull);
    return (cl <= 12 ? sgc_mul24c<SGC_CORE_M1>(corev) : corev * SGC_CORE_M1) & m;
}
SGC_HD uint32_t sgc_core_unmix(uint32_t h, uint32_t cl) {
    const uint32_t m = (uint32_t)((1ull << (2 * cl)) - 1ull);
    return (cl <= 12 ? sgc_mul24c<SGC_CORE_M1_INV>(h) : h * SGC_CORE_M1_INV) & m;
}
SGC_HD uint32_t sgc_core_hash(uint32_t corev, uint32_t cl) { return sgc_core_mix(corev, cl) << (32 - 2 * cl); }
// 32-bit hash of a whole key (<= 48 bits: one-word records), for the slot inside a core-hashed slice: two full-rate 24-bit
// multiplies, one per half of the key, instead of the 64-bit sgc_hash.  Its top bits pick the slot, the bits below them the
// displacement to the key's alternate slot (sgc_cuckoo_alt) — k_count_slices evaluates both for every read.
SGC_HD uint32_t sgc_hash32(uint64_t key) {
    return sgc_mul24c<0x9E3779u>((uint32_t)key) + sgc_mul24c<0xC2B2AFu>((uint32_t)(key >> 24));
}
SGC_HD uint32_t sgc_core_part(uint32_t h, uint32_t log2_p) { return log2_p ? h >> (32 - log2_p) : 0u; }
// the same from the mixed value alone (needs log2_p <= 2 cl)
SGC_HD uint32_t sgc_core_part_mix(uint32_t mix, uint32_t cl, uint32_t log2_p) { return mix >> (2 * cl - log2_p); }
// Home bucket of a key in a sliced table whose slices follow the core hash (sgc_table_view::core_cl): slice = the top
// bits of the hash of key bases [1, 1 + core_cl) (= span bases [2, 2 + core_cl): core A as the Centered window sees
// it), bucket inside the slice = the top bits of the full-key hash.  (The builder only makes such slices when their index
// fits the mixed value: log2_slots - log2_slice <= 2 core_cl.)
SGC_HD uint32_t sgc_home_bucket_ex(uint64_t key, uint32_t log2_slots, uint32_t log2_slice, uint32_t core_cl) {
    if (core_cl == 0 || log2_slice >= log2_slots) return sgc_home_bucket(key, log2_slots);
    const uint32_t hm = sgc_core_mix((uint32_t)((key >> 2) & ((1ull << (2 * core_cl)) - 1ull)), core_cl);
    return (sgc_core_part_mix(hm, core_cl, log2_slots - log2_slice) << (log2_slice - 1)) | (sgc_hash32(key) >> (33 - log2_slice));
}
// Home SLOT of a key (log2_slots bits); its upper bits are sgc_home_bucket_ex.  The slot inside the slice is what
// k_partition tags into a clean record and where the two-choice image below looks first.
SGC_HD uint32_t sgc_home_slot_ex(uint64_t key, uint32_t log2_slots, uint32_t log2_slice, uint32_t core_cl) {
    if (core_cl == 0 || log2_slice >= log2_slots) return (uint32_t)(sgc_hash(key) >> (64 - log2_slots));
    const uint32_t hm = sgc_core_mix((uint32_t)((key >> 2) & ((1ull << (2 * core_cl)) - 1ull)), core_cl);
    return (sgc_core_part_mix(hm, core_cl, log2_slots - log2_slice) << log2_slice) | (sgc_hash32(key) >> (32 - log2_slice));
}
// Two-choice image of a library slice for k_count_slices (sgc_part.hip): ONE-slot buckets, a key sits in its home slot s1
// (inside the slice) or in the alternate slot s1 ^ d(key), d != 0 — both are read at once (two 8-byte LDS reads: a random
// 16-byte gather costs twice the bank-conflict cycles, and those were a quarter of that kernel's busy time) and the probe has
// no loop at all (the open-addressed layout needs one for full buckets, and its exec-mask bookkeeping was most of that
// kernel's scalar instruction stream).  The table is built at load <= 0.4, below the 0.5 threshold of two-choice cuckoo
// placement with one-slot buckets.  ls = log2(slots per slice).
SGC_HD uint32_t sgc_cuckoo_alt_h(uint32_t h32 /* sgc_hash32(key) */, uint32_t s1, uint32_t ls) {
    if (ls == 0) return s1;
    uint32_t d = (h32 << ls) >> (32 - ls);          // the ls bits below the slot bits
    d |= (uint32_t)(d == 0);
    return s1 ^ d;
}
SGC_HD uint32_t sgc_cuckoo_alt(uint64_t key, uint32_t s1, uint32_t ls) { return sgc_cuckoo_alt_h(sgc_hash32(key), s1, ls); }
SGC_HD uint32_t sgc_core_home(uint32_t h, uint32_t log2_p) {
    return (h >> (32 - log2_p - SGC_CORE_LOG2_S)) & ((1u << SGC_CORE_LOG2_S) - 1u);
}
// the bases of a window (2-bit packed, L bases) outside core positions [lowlen, lowlen + cl), closed up
SGC_HD uint32_t sgc_core_rest(uint64_t window, uint32_t lowlen, uint32_t cl) {
    return (uint32_t)((window & ((1ull << (2 * lowlen)) - 1ull)) | ((window >> (2 * (lowlen + cl))) << (2 * lowlen)));
}

// Builds one record from a read.  `emit(span_bits, status)` style is avoided to keep this usable in
// kernels: returns span and status through references.
SGC_HD void sgc_pack_one(const uint8_t *seq, uint64_t n, uint32_t L, int reverse, uint32_t o, int recursion,
                         uint64_t &span, uint64_t &status) {
    const uint32_t K = L + 2;
    // window bounds (src/counter.rs:158-180); all arithmetic in u64 like the reference's usize
    const bool c_ok = (uint64_t)o + L <= n;
    const bool p_ok = c_ok && recursion && ((uint64_t)o + 1 + L <= n);
    const bool m_ok = p_ok && o >= 1;
    span = 0;
    uint32_t ninv[3] = {0, 0, 0};   // index 0 = M, 1 = C, 2 = P
    uint32_t st[3] = {0, 0, 0};
    bool bad[3] = {false, false, false};
    if (c_ok) {
        for (uint32_t w = 0; w < K; w++) {
            const int64_t p = (int64_t)o - 1 + (int64_t)w;
            if (p < 0 || (uint64_t)p >= n) continue;      // only reachable for windows already out of bounds
            const uint8_t b = reverse ? seq[n - 1 - (uint64_t)p] : seq[p];
            const uint32_t code = reverse ? sgc_base_code_rc(b) : sgc_base_code(b);
            if (code < 4) { span |= (uint64_t)code << (2 * w); continue; }
            // span base w sits at window position w (M), w-1 (C), w-2 (P)
            for (int k = 0; k < 3; k++) {
                const int32_t j = (int32_t)w - k;
                if (j < 0 || j >= (int32_t)L) continue;
                ninv[k]++;
                if (code == 4) st[k] = 2u + (uint32_t)j; else bad[k] = true;
            }
        }
    }
    const bool ok[3] = {m_ok, c_ok, p_ok};
    for (int k = 0; k < 3; k++) {
        if (!ok[k] || ninv[k] >= 2 || bad[k]) st[k] = SGC_STATE_DEAD;
        else if (ninv[k] == 0) st[k] = SGC_STATE_CLEAN;
    }
    status = (uint64_t)st[1] + (uint64_t)K * ((uint64_t)st[2] + (uint64_t)K * (uint64_t)st[0]);
}
