// sgc_tables.cpp — host-side construction of the library and single-mismatch tables.
#include "sgc_tables.h"

#include <string.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <utility>

#include "../../include/sgcount_hip.h"
#include "sgc_bytes.h"

bool sgc_pack_key(const uint8_t *seq, uint32_t L, uint64_t &key) {
    key = 0;
    for (uint32_t j = 0; j < L; j++) {
        const uint32_t c = sgc_base_code(seq[j]);
        if (c > 3) return false;
        key |= (uint64_t)c << (2 * j);
    }
    return true;
}

static uint32_t ceil_log2(uint64_t x) {
    uint32_t l = 0;
    while ((1ull << l) < x) l++;
    return l;
}

static void table_alloc(sgc_host_table &t, uint64_t entries, double max_load, uint32_t gid_bits,
                        uint32_t max_log2_slice = 0, uint32_t extra_log2 = 0) {
    uint64_t want = (uint64_t)((double)entries / max_load) + 1;
    t.log2_slots = std::max<uint32_t>(4, ceil_log2(want)) + extra_log2;
    if (max_log2_slice == 1) max_log2_slice = 2;
    t.log2_slice = (max_log2_slice && max_log2_slice < t.log2_slots) ? max_log2_slice : t.log2_slots;
    t.gid_bits = gid_bits;
    t.slots.assign(1ull << t.log2_slots, SGC_EMPTY);
    if (gid_bits == 0) t.vals.assign(1ull << t.log2_slots, SGC_NONE); else t.vals.clear();
    t.entries = 0;
}

static inline void table_insert(sgc_host_table &t, uint64_t key, uint32_t gid) {
    uint32_t b = sgc_home_bucket_ex(key, t.log2_slots, t.log2_slice, t.core_cl);
    for (;;) {
        for (uint32_t k = 0; k < 2; k++) {
            const uint64_t h = 2ull * b + k;
            if (t.slots[h] != SGC_EMPTY) continue;
            if (t.gid_bits) t.slots[h] = (key << t.gid_bits) | gid;
            else { t.slots[h] = key; t.vals[h] = gid; }
            t.entries++;
            return;
        }
        b = sgc_next_bucket(b, t.log2_slice);
    }
}

static inline uint32_t table_find_host(const sgc_host_table &t, uint64_t key) {
    uint32_t b = sgc_home_bucket_ex(key, t.log2_slots, t.log2_slice, t.core_cl);
    for (;;) {
        for (uint32_t k = 0; k < 2; k++) {
            const uint64_t h = 2ull * b + k;
            const uint64_t s = t.slots[h];
            if (s == SGC_EMPTY) return SGC_NONE;
            if (t.gid_bits) { if ((s >> t.gid_bits) == key) return (uint32_t)(s & ((1ull << t.gid_bits) - 1)); }
            else if (s == key) return t.vals[h];
        }
        b = sgc_next_bucket(b, t.log2_slice);
    }
}

// gid field width for the packed layout, or 0 if (key, gid) does not fit one u64
static uint32_t choose_gid_bits(uint32_t n, uint32_t L) {
    const uint32_t room = 64 - 2 * L;
    const uint32_t gb = std::min<uint32_t>(room, 32);
    if (gb == 0) return 0;
    // the all-ones gid is reserved so that a full slot can never equal SGC_EMPTY
    if (gb < 32 && (uint64_t)n > (1ull << gb) - 1) return 0;
    return gb;
}

// true if some slice is filled beyond `limit` of its slots (probing must always find a free slot)
static bool slice_overfull(const std::vector<uint64_t> &keys, uint32_t log2_slots, uint32_t log2_slice, uint32_t core_cl, double limit) {
    if (log2_slice >= log2_slots) return false;
    std::vector<uint32_t> fill(1ull << (log2_slots - log2_slice), 0);
    const uint32_t cap = (uint32_t)((double)(1u << log2_slice) * limit);
    for (uint64_t k : keys)
        if (++fill[sgc_home_bucket_ex(k, log2_slots, log2_slice, core_cl) >> (log2_slice - 1)] > cap) return true;
    return false;
}

int sgc_build_library_table(const uint8_t *seqs, uint32_t n, uint32_t L, uint32_t max_log2_slice, uint32_t core_cl,
                            std::vector<uint64_t> &keys, sgc_host_table &out, std::string &err) {
    if (L == 0 || L > SGC_MAXL) { err = "guide length " + std::to_string(L) + " outside 1.." + std::to_string(SGC_MAXL); return SGC_E_UNSUPPORTED; }
    if (n == 0) { err = "empty library"; return SGC_E_ARG; }
    keys.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        if (!sgc_pack_key(seqs + (size_t)i * L, L, keys[i])) {
            err = "library sequence " + std::to_string(i) + " has bytes outside ACGT";
            return SGC_E_UNSUPPORTED;
        }
    }
    // grow the table until no slice is more than 60 % full (hash imbalance between slices); slices that follow the core
    // hash are tried first (once, without growing: guides that share those bases do not spread however large the table)
    if (core_cl && 2 * (1 + core_cl) > 2 * L) core_cl = 0;
    out.core_cl = 0;
    if (core_cl) {
        table_alloc(out, n, 0.4, choose_gid_bits(n, L), max_log2_slice, 0);
        // (the slice index must be a prefix of the mixed core value: no more slice bits than core bits)
        if (out.log2_slice < out.log2_slots && out.log2_slots - out.log2_slice <= 2 * core_cl &&
            !slice_overfull(keys, out.log2_slots, out.log2_slice, core_cl, 0.6)) out.core_cl = core_cl;
    }
    if (!out.core_cl) {
        uint32_t extra = 0;
        for (;; extra++) {
            table_alloc(out, n, 0.4, choose_gid_bits(n, L), max_log2_slice, extra);
            if (extra >= 4 || !slice_overfull(keys, out.log2_slots, out.log2_slice, 0, 0.6)) break;
        }
        if (slice_overfull(keys, out.log2_slots, out.log2_slice, 0, 0.9)) {   // pathological key set: give up slicing
            table_alloc(out, n, 0.4, choose_gid_bits(n, L), 0, 0);
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        if (table_find_host(out, keys[i]) != SGC_NONE) {      // src/library.rs:91-96
            err = "Unexpected duplicate sequence in library found: " + std::string((const char *)seqs + (size_t)i * L, L);
            return SGC_E_DUPLICATE;
        }
        table_insert(out, keys[i], i);
    }
    return SGC_OK;
}

void sgc_build_bloom(const std::vector<uint64_t> &keys, uint32_t log2_words, std::vector<uint64_t> &out) {
    out.assign(1ull << log2_words, 0);
    for (uint64_t k : keys) {
        const uint64_t h = sgc_hash2(k);
        out[sgc_bloom_word(h, log2_words)] |= sgc_bloom_mask(h);
    }
}

uint32_t sgc_bloom_log2_words(uint64_t n_keys, uint32_t bits_per_key, uint32_t min_log2, uint32_t max_log2) {
    const uint32_t l = ceil_log2((n_keys * bits_per_key + 63) / 64 + 1);
    return std::min(std::max(l, min_log2), max_log2);
}

uint32_t sgc_permute_log2_slots(uint64_t n_children) {
    return std::max<uint32_t>(4, ceil_log2((uint64_t)((double)n_children / 0.5) + 1));
}

void sgc_build_permute_table(const std::vector<uint64_t> &keys, uint32_t L, const sgc_host_table &lib,
                             sgc_host_table &out, std::vector<uint64_t> *child_keys, std::vector<uint64_t> *amb) {
    const uint32_t n = (uint32_t)keys.size();
    const uint32_t gb = lib.gid_bits;
    // all (child, parent) pairs: 3 ACGT substitutions per position (src/permutes.rs:78-107 minus the 'N' column).
    // The table is sized for all of them (nearly all survive); children are ordered by their home bucket with a
    // two-pass radix sort, so that duplicate children meet in one bucket run and the inserts walk the table
    // sequentially instead of missing the cache 6 M times.
    struct Kid { uint64_t key; uint32_t bucket, gid; };
    const size_t total = (size_t)n * 3 * L;
    table_alloc(out, total, 0.5, gb);
    const uint32_t log2_buckets = out.log2_slots - 1;
    std::vector<Kid> a(total), b(total);
    {
        size_t w = 0;
        for (uint32_t g = 0; g < n; g++) {
            const uint64_t k = keys[g];
            for (uint32_t j = 0; j < L; j++)
                for (uint64_t d = 1; d < 4; d++) {
                    const uint64_t c = k ^ (d << (2 * j));
                    a[w++] = Kid{c, sgc_home_bucket(c, out.log2_slots), g};
                }
        }
    }
    const uint32_t lo_bits = log2_buckets / 2, hi_bits = log2_buckets - lo_bits;
    auto radix_pass = [&](std::vector<Kid> &src, std::vector<Kid> &dst, uint32_t shift, uint32_t bits) {
        std::vector<size_t> cnt((size_t)1 << bits, 0);
        const uint32_t mask = (1u << bits) - 1u;
        for (const Kid &x : src) cnt[(x.bucket >> shift) & mask]++;
        size_t run = 0;
        for (auto &c : cnt) { const size_t v = c; c = run; run += v; }
        for (const Kid &x : src) dst[cnt[(x.bucket >> shift) & mask]++] = x;
    };
    radix_pass(a, b, 0, lo_bits);
    radix_pass(b, a, lo_bits, hi_bits);
    // per bucket run: keep children with exactly one parent that are not library members (src/permutes.rs:127-144:
    // a second sighting moves the child to `null`; a parent is in `null` from the start)
    if (child_keys) { child_keys->clear(); child_keys->reserve(total); }
    if (amb) amb->assign((size_t)n * 2, 0);
    for (size_t i = 0; i < total;) {
        size_t j = i + 1;
        while (j < total && a[j].bucket == a[i].bucket) j++;
        if (j - i > 1) std::sort(a.begin() + i, a.begin() + j, [](const Kid &x, const Kid &y) { return x.key < y.key; });
        for (size_t u = i; u < j;) {
            size_t v = u + 1;
            while (v < j && a[v].key == a[u].key) v++;
            if (v - u == 1 && table_find_host(lib, a[u].key) == SGC_NONE) {
                table_insert(out, a[u].key, a[u].gid);
                if (child_keys) child_keys->push_back(a[u].key);
            } else if (amb) {
                for (size_t q = u; q < v; q++) {
                    const uint64_t x = a[q].key ^ keys[a[q].gid];            // one differing base
                    const uint32_t j = (uint32_t)__builtin_ctzll(x) / 2, bit = 4 * j + (uint32_t)((a[q].key >> (2 * j)) & 3);
                    (*amb)[(size_t)a[q].gid * 2 + (bit >> 6)] |= 1ull << (bit & 63);
                }
            }
            u = v;
        }
        i = j;
    }
}

bool sgc_build_core_index(const std::vector<uint64_t> &keys, uint32_t L, uint32_t cs, uint32_t cl, sgc_host_core &out) {
    const uint32_t n = (uint32_t)keys.size(), NB = 1u << SGC_CORE_LOG2_S;
    if (cs < 2 || cl == 0 || cl > 14 || cs + cl > L || L > SGC_REC8_MAXL || L - cl > 16) return false;
    const uint64_t cmask = (1ull << (2 * cl)) - 1ull;
    std::vector<uint32_t> h((size_t)n * 3);
    for (uint32_t g = 0; g < n; g++)
        for (uint32_t a = 0; a < 3; a++)               // window position of span base cs at alignment a: cs - a
            h[(size_t)g * 3 + a] = sgc_core_hash((uint32_t)((keys[g] >> (2 * (cs - a))) & cmask), cl);
    uint32_t lp = 0;
    std::vector<uint32_t> fill;
    for (;; lp++) {
        if (lp > SGC_CORE_MAX_LOG2_P) return false;
        fill.assign(1u << lp, 0);
        bool ok = true;
        for (uint32_t x : h)
            if (++fill[sgc_core_part(x, lp)] > SGC_CORE_EMAX) { ok = false; break; }
        if (ok) break;
    }
    const size_t P = (size_t)1 << lp;
    out.log2_p = lp; out.cs = cs; out.cl = cl;
    out.ents.assign(P * SGC_CORE_EMAX, 0);
    out.gids.assign(P * SGC_CORE_EMAX, SGC_NONE);
    out.starts.assign(P * SGC_CORE_STARTS, 0);
    // counting sort by (partition, bucket)
    std::vector<uint32_t> cnt(P * (NB + 1), 0);
    for (uint32_t x : h) cnt[(size_t)sgc_core_part(x, lp) * (NB + 1) + sgc_core_home(x, lp) + 1]++;
    for (size_t p = 0; p < P; p++) {
        uint32_t *c = &cnt[p * (NB + 1)];
        for (uint32_t b = 0; b < NB; b++) c[b + 1] += c[b];
        for (uint32_t b = 0; b <= NB; b++) out.starts[p * SGC_CORE_STARTS + b] = (uint16_t)c[b];
    }
    for (uint32_t g = 0; g < n; g++)
        for (uint32_t a = 0; a < 3; a++) {
            const uint32_t x = h[(size_t)g * 3 + a];
            const size_t p = sgc_core_part(x, lp);
            const uint32_t at = cnt[p * (NB + 1) + sgc_core_home(x, lp)]++;
            const uint32_t lowlen = cs - a;
            const uint64_t core = (keys[g] >> (2 * lowlen)) & cmask;
            out.ents[p * SGC_CORE_EMAX + at] = core | ((uint64_t)a << 30) | ((uint64_t)sgc_core_rest(keys[g], lowlen, cl) << 32);
            out.gids[p * SGC_CORE_EMAX + at] = g;
        }
    return true;
}

uint32_t sgc_rest_filter_log2(uint32_t n_guides) {
    uint32_t l = 16;
    while (l < 24 && (1ull << l) < (uint64_t)n_guides * 32) l++;       // <= ~3 % of the bits set per alignment
    return l;
}

void sgc_build_rest_filter(const std::vector<uint64_t> &keys, uint32_t cs, uint32_t cl, uint32_t log2_bits, std::vector<uint32_t> &out) {
    const size_t words = (size_t)1 << (log2_bits - 5);
    out.assign(3 * words, 0u);
    for (uint64_t k : keys)
        for (uint32_t a = 0; a < 3; a++) {
            const uint32_t idx = sgc_rest_hash(sgc_core_rest(k, cs - a, cl), log2_bits);
            out[a * words + (idx >> 5)] |= 1u << (idx & 31u);
        }
}

bool sgc_build_slice_cuckoo(const sgc_host_table &lib, std::vector<uint64_t> &out) {
    if (lib.gid_bits == 0 || lib.log2_slice < 1 || lib.log2_slice > lib.log2_slots) return false;
    const uint32_t S = 1u << lib.log2_slice, ls = lib.log2_slice;
    const size_t n_slices = (size_t)1 << (lib.log2_slots - lib.log2_slice);
    out.assign(lib.slots.size(), SGC_EMPTY);
    uint64_t rng = 0x243F6A8885A308D3ull;
    for (size_t s = 0; s < n_slices; s++) {
        uint64_t *t = &out[s * S];
        for (uint32_t i = 0; i < S; i++) {
            uint64_t cur = lib.slots[s * S + i];
            if (cur == SGC_EMPTY) continue;
            uint64_t key = cur >> lib.gid_bits;
            uint32_t a = sgc_home_slot_ex(key, lib.log2_slots, lib.log2_slice, lib.core_cl) & (S - 1);
            bool placed = false;
            for (int kick = 0; kick < 4000 && !placed; kick++) {
                const uint32_t a2 = sgc_cuckoo_alt(key, a, ls);
                if (t[a] == SGC_EMPTY) { t[a] = cur; placed = true; break; }
                if (t[a2] == SGC_EMPTY) { t[a2] = cur; placed = true; break; }
                // evict the occupant of one of the two slots and re-place it from its other slot
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                const uint32_t v = (rng >> 33) & 1 ? a2 : a;
                std::swap(cur, t[v]);
                key = cur >> lib.gid_bits;
                a = sgc_cuckoo_alt(key, v, ls);       // the evicted key's other slot (the relation is symmetric)
            }
            if (!placed) return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// generic byte-string tables (sgc_bytes.h)
// ---------------------------------------------------------------------------------------------------------------------
static uint64_t bytes_hash(const uint8_t *p, uint32_t L) {
    uint64_t h = sgc_bytes_hash_init();
    for (uint32_t i = 0; i < L; i++) h = sgc_bytes_hash_step(h, p[i]);
    return sgc_bytes_hash_fin(h);
}
int sgc_build_bytes_tables(const uint8_t *seqs, uint32_t n, uint32_t L, bool one_mm, sgc_host_bytes &out, std::string &err) {
    if (L == 0 || L > SGC_BYTES_MAXL) { err = "guide length " + std::to_string(L) + " outside 1.." + std::to_string(SGC_BYTES_MAXL); return SGC_E_UNSUPPORTED; }
    if (n == 0) { err = "empty library"; return SGC_E_ARG; }
    out.lib_log2 = std::max<uint32_t>(4, ceil_log2((uint64_t)n * 2 + 1));
    out.lib_tag.assign((size_t)1 << out.lib_log2, SGC_BYTES_EMPTY);
    out.lib_val.assign((size_t)1 << out.lib_log2, SGC_NONE);
    const uint32_t lmask = (1u << out.lib_log2) - 1u;
    auto lib_find = [&](uint64_t h, const uint8_t *w) -> uint32_t {          // Library::contains on bytes
        for (uint32_t s = sgc_bytes_slot(h, out.lib_log2);; s = (s + 1) & lmask) {
            if (out.lib_tag[s] == SGC_BYTES_EMPTY) return SGC_NONE;
            if (out.lib_tag[s] == h && memcmp(seqs + (size_t)out.lib_val[s] * L, w, L) == 0) return out.lib_val[s];
        }
    };
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *w = seqs + (size_t)i * L;
        const uint64_t h = bytes_hash(w, L);
        if (lib_find(h, w) != SGC_NONE) {                                      // src/library.rs:91-96
            err = "Unexpected duplicate sequence in library found: " + std::string((const char *)w, L);
            return SGC_E_DUPLICATE;
        }
        uint32_t s = sgc_bytes_slot(h, out.lib_log2);
        while (out.lib_tag[s] != SGC_BYTES_EMPTY) s = (s + 1) & lmask;
        out.lib_tag[s] = h; out.lib_val[s] = i;
    }
    out.perm_entries = 0;
    if (!one_mm) return SGC_OK;
    // children: every guide, every position, every letter of the lexicon that differs (src/permutes.rs:3,78-117).  8.0 M of them for
    // 100k guides of 20 bases: generated, sorted by hash and screened by a few threads — buckets by the top bits of the hash, so that a
    // run of equal hashes never straddles two buckets (a second was the whole of the CLI's start-up with such a library).
    static const uint8_t LEX[5] = {'A', 'C', 'G', 'T', 'N'};
    struct Kid { uint64_t h; uint32_t g, j; uint8_t b; };
    const unsigned T = std::max(1u, std::min(8u, std::min(std::thread::hardware_concurrency(), n / 2048u + 1u)));
    constexpr unsigned NB = 64;                                              // buckets: the top 6 bits of the hash
    std::vector<std::vector<std::vector<Kid>>> gen(T, std::vector<std::vector<Kid>>(NB));
    auto generate = [&](unsigned t) {
        const uint32_t g0 = (uint32_t)((uint64_t)n * t / T), g1 = (uint32_t)((uint64_t)n * (t + 1) / T);
        for (auto &v : gen[t]) v.reserve((size_t)(g1 - g0) * L * 4 / NB + 16);
        std::vector<uint64_t> prefix(L + 1);
        for (uint32_t g = g0; g < g1; g++) {
            const uint8_t *w = seqs + (size_t)g * L;
            prefix[0] = sgc_bytes_hash_init();
            for (uint32_t i = 0; i < L; i++) prefix[i + 1] = sgc_bytes_hash_step(prefix[i], w[i]);      // the hash state in front of every position
            for (uint32_t j = 0; j < L; j++)
                for (uint8_t bb : LEX) {
                    if (bb == w[j]) continue;
                    uint64_t h = sgc_bytes_hash_step(prefix[j], bb);
                    for (uint32_t i = j + 1; i < L; i++) h = sgc_bytes_hash_step(h, w[i]);
                    h = sgc_bytes_hash_fin(h);
                    gen[t][h >> 58].push_back(Kid{h, g, j, bb});
                }
        }
    };
    auto same_child = [&](const Kid &a, const Kid &b) {                       // do two (guide, position, letter) triples spell one string?
        const uint8_t *x = seqs + (size_t)a.g * L, *y = seqs + (size_t)b.g * L;
        for (uint32_t i = 0; i < L; i++) {
            const uint8_t cx = i == a.j ? a.b : x[i], cy = i == b.j ? b.b : y[i];
            if (cx != cy) return false;
        }
        return true;
    };
    std::vector<std::vector<Kid>> kept(NB);
    auto screen = [&](unsigned t) {
        std::vector<uint8_t> child(L);
        for (unsigned bk = t; bk < NB; bk += T) {
            std::vector<Kid> kids;
            size_t total = 0;
            for (unsigned u = 0; u < T; u++) total += gen[u][bk].size();
            kids.reserve(total);
            for (unsigned u = 0; u < T; u++) { kids.insert(kids.end(), gen[u][bk].begin(), gen[u][bk].end()); std::vector<Kid>().swap(gen[u][bk]); }
            // (guide, position, letter) breaks ties, so that the result does not depend on the number of threads
            std::sort(kids.begin(), kids.end(), [](const Kid &a, const Kid &b) { return a.h != b.h ? a.h < b.h : (a.g != b.g ? a.g < b.g : (a.j != b.j ? a.j < b.j : a.b < b.b)); });
            for (size_t i = 0; i < kids.size();) {
                size_t e = i;
                while (e < kids.size() && kids[e].h == kids[i].h) e++;
                // inside a run of equal hashes (almost always one string): a child survives iff no other triple spells the same string
                for (size_t a = i; a < e; a++) {
                    bool unique = true;
                    for (size_t b = i; b < e && unique; b++)
                        if (b != a && same_child(kids[a], kids[b])) unique = false;
                    if (!unique) continue;
                    const uint8_t *w = seqs + (size_t)kids[a].g * L;
                    for (uint32_t k = 0; k < L; k++) child[k] = k == kids[a].j ? kids[a].b : w[k];
                    if (lib_find(kids[a].h, child.data()) != SGC_NONE) continue;       // a library member is a parent: nulled (permutes.rs:149-152)
                    kept[bk].push_back(kids[a]);
                }
                i = e;
            }
        }
    };
    auto run_threads = [&](const std::function<void(unsigned)> &f) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(f, t);
        f(0);
        for (auto &x : th) x.join();
    };
    run_threads(generate);
    run_threads(screen);
    std::vector<Kid> kids;                                                   // the survivors, in hash order
    {
        size_t total = 0;
        for (auto &v : kept) total += v.size();
        kids.reserve(total);
        for (auto &v : kept) { kids.insert(kids.end(), v.begin(), v.end()); std::vector<Kid>().swap(v); }
    }
    std::vector<size_t> keep(kids.size());
    for (size_t i = 0; i < keep.size(); i++) keep[i] = i;
    out.perm_log2 = std::max<uint32_t>(4, ceil_log2((uint64_t)keep.size() * 2 + 1));
    out.perm_tag.assign((size_t)1 << out.perm_log2, SGC_BYTES_EMPTY);
    out.perm_val.assign((size_t)1 << out.perm_log2, SGC_NONE);
    out.perm_pl.assign((size_t)1 << out.perm_log2, 0);
    const uint32_t pmask = (1u << out.perm_log2) - 1u;
    for (size_t a : keep) {
        uint32_t s = sgc_bytes_slot(kids[a].h, out.perm_log2);
        while (out.perm_tag[s] != SGC_BYTES_EMPTY) s = (s + 1) & pmask;
        out.perm_tag[s] = kids[a].h; out.perm_val[s] = kids[a].g; out.perm_pl[s] = kids[a].j | (uint32_t)kids[a].b << 24;
    }
    out.perm_entries = keep.size();
    return SGC_OK;
}
