/*
 * sgcount_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see sgcount_oracle.h).
 *
 * A byte-string restatement of the reference's count path.  Every function
 * cites the reference file:line it follows (paths relative to the reference
 * checkout, noamteyssier/sgcount v0.1.35).  The structure deliberately mirrors
 * the reference: byte-keyed hash maps (hashbrown there, a small open-addressed
 * FNV map here), one pass over records, a window copy per probe, an id-keyed
 * result map.  It is single-threaded because the reference is single-threaded
 * within a sample (count.rs:117-118 parallelises across samples only).
 *
 * PINNED by the reference's own unit-test vectors (counter.rs:283-382,
 * permutes.rs:193-253, library.rs:119-136, offsetter.rs:249-362: tests/test_oracle_kat.py)
 * and by the reference's example/ files through their read names
 * (tests/test_oracle_fixtures.py).  NOT pinned (no upstream test, third-party
 * fxread ^0.2.5 — DESIGN.md §2): Offset::Reverse on non-ACGT bytes, and two
 * reader decisions taken here and shared with the product's readers: a '\r'
 * before the '\n' belongs to the line terminator, and a FASTQ header without
 * '@' / separator without '+' is a malformed record (ORC_E_FORMAT).
 */
#include "sgcount_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* byte-string hash map (stand-in for hashbrown::HashMap<Vec<u8>, _>)          */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t **key; /* not owned unless own_keys */
    uint32_t *klen;
    uint64_t *val;
    uint8_t *state; /* 0 empty, 1 full, 2 tombstone */
    size_t cap, n, used;
    int own_keys;
} bsmap;

static uint64_t bs_hash(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

static void bsmap_init(bsmap *m, size_t cap_hint, int own_keys) {
    size_t cap = 16;
    while (cap < cap_hint * 2) cap <<= 1;
    m->cap = cap; m->n = 0; m->used = 0; m->own_keys = own_keys;
    m->key = (const uint8_t **)calloc(cap, sizeof(*m->key));
    m->klen = (uint32_t *)calloc(cap, sizeof(*m->klen));
    m->val = (uint64_t *)calloc(cap, sizeof(*m->val));
    m->state = (uint8_t *)calloc(cap, 1);
}

static void bsmap_free(bsmap *m) {
    if (m->own_keys)
        for (size_t i = 0; i < m->cap; i++)
            if (m->state[i] == 1) free((void *)m->key[i]);
    free((void *)m->key); free(m->klen); free(m->val); free(m->state);
    memset(m, 0, sizeof(*m));
}

/* returns slot index of key or (size_t)-1 */
static size_t bsmap_find(const bsmap *m, const uint8_t *k, size_t n) {
    size_t mask = m->cap - 1, i = bs_hash(k, n) & mask;
    for (;;) {
        if (m->state[i] == 0) return (size_t)-1;
        if (m->state[i] == 1 && m->klen[i] == n && memcmp(m->key[i], k, n) == 0) return i;
        i = (i + 1) & mask;
    }
}

static void bsmap_grow(bsmap *m);

/* insert-or-get: returns slot; *existed tells whether the key was present */
static size_t bsmap_entry(bsmap *m, const uint8_t *k, size_t n, int *existed) {
    if ((m->used + 1) * 2 > m->cap) bsmap_grow(m);
    size_t mask = m->cap - 1, i = bs_hash(k, n) & mask, tomb = (size_t)-1;
    for (;;) {
        if (m->state[i] == 0) break;
        if (m->state[i] == 2) { if (tomb == (size_t)-1) tomb = i; }
        else if (m->klen[i] == n && memcmp(m->key[i], k, n) == 0) { *existed = 1; return i; }
        i = (i + 1) & mask;
    }
    *existed = 0;
    if (tomb != (size_t)-1) i = tomb; else m->used++;
    if (m->own_keys) {
        uint8_t *c = (uint8_t *)malloc(n ? n : 1);
        memcpy(c, k, n);
        m->key[i] = c;
    } else {
        m->key[i] = k;
    }
    m->klen[i] = (uint32_t)n; m->val[i] = 0; m->state[i] = 1; m->n++;
    return i;
}

static void bsmap_remove_slot(bsmap *m, size_t i) {
    if (m->own_keys) free((void *)m->key[i]);
    m->key[i] = NULL; m->state[i] = 2; m->n--;
}

static void bsmap_grow(bsmap *m) {
    bsmap old = *m;
    bsmap_init(m, old.cap, old.own_keys);
    for (size_t i = 0; i < old.cap; i++) {
        if (old.state[i] != 1) continue;
        size_t mask = m->cap - 1, j = bs_hash(old.key[i], old.klen[i]) & mask;
        while (m->state[j]) j = (j + 1) & mask;
        m->key[j] = old.key[i]; m->klen[j] = old.klen[i]; m->val[j] = old.val[i];
        m->state[j] = 1; m->n++; m->used++;
    }
    free((void *)old.key); free(old.klen); free(old.val); free(old.state);
}

/* ------------------------------------------------------------------------- */
/* FASTX text → records (stand-in for fxread ^0.2.5; call sites count.rs:24,  */
/* library.rs:89-92, offsetter.rs:38,59).  Pinned by upstream tests only for   */
/* single-line FASTA, id = header minus marker, seq without newline            */
/* (library.rs:119-130, counter.rs:261-272).  FASTQ = 4 lines per record.      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *id; size_t id_len;
    const uint8_t *seq; size_t seq_len;
} fx_record;

typedef struct {
    const uint8_t *p, *end;
    int fmt; /* 0 unknown, 1 fasta, 2 fastq */
} fx_iter;

static void fx_iter_init(fx_iter *it, const uint8_t *buf, size_t len) {
    it->p = buf; it->end = buf + len; it->fmt = 0;
}

static const uint8_t *fx_line(fx_iter *it, size_t *n) {
    if (it->p >= it->end) return NULL;
    const uint8_t *s = it->p;
    const uint8_t *nl = (const uint8_t *)memchr(s, '\n', (size_t)(it->end - s));
    if (nl) { *n = (size_t)(nl - s); it->p = nl + 1; }
    else { *n = (size_t)(it->end - s); it->p = it->end; }
    /* CRLF: a '\r' before the '\n' is taken as part of the terminator.  fxread's handling is NOT pinned by any upstream
       test; this is a decision shared with the product's readers, not a reference fact. */
    if (*n && s[*n - 1] == '\r') (*n)--;
    return s;
}

/* returns 1 record, 0 end, ORC_E_FORMAT on malformed input */
static int fx_next(fx_iter *it, fx_record *r) {
    size_t n;
    const uint8_t *h = fx_line(it, &n);
    if (!h) return 0;
    if (n == 0) {
        /* blank lines at the very end are not records (any number of them); a blank line with anything behind it is a malformed header */
        const uint8_t *q = it->p;
        while (q < it->end && (*q == '\n' || *q == '\r')) q++;
        if (q >= it->end) return 0;
        return ORC_E_FORMAT;
    }
    if (it->fmt == 0) {
        if (n && h[0] == '>') it->fmt = 1;
        else if (n && h[0] == '@') it->fmt = 2;
        else return ORC_E_FORMAT;
    }
    if (n == 0 || h[0] != (it->fmt == 1 ? '>' : '@')) return ORC_E_FORMAT;
    r->id = h + 1; r->id_len = n - 1;
    const uint8_t *s = fx_line(it, &n);
    if (!s) return ORC_E_FORMAT;
    r->seq = s; r->seq_len = n;
    if (it->fmt == 2) {
        size_t m;
        const uint8_t *plus = fx_line(it, &m);
        if (!plus || m == 0 || plus[0] != '+') return ORC_E_FORMAT;
        /* A stream that ends behind the separator line ends with a record whose quality line is empty: written with its terminator
           ("+\n\n"), without it ("+\n") or not at all ("+") — no reader can tell those apart from a cut-off file, and fxread's choice is
           not pinned here; decision #3 of DESIGN.md §2, shared with the product's readers (not a reference fact). */
        (void)fx_line(it, &m);
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Library — library.rs                                                       */
/* ------------------------------------------------------------------------- */

struct orc_library {
    bsmap table;        /* seq → index into recs (library.rs:10 HashMap<seq,id>) */
    uint8_t **seq; uint8_t **id; size_t *seq_len; size_t *id_len;
    size_t n, cap;
    size_t size;        /* library.rs:11 */
};

static void lib_push(orc_library *L, const fx_record *r) {
    if (L->n == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 1024;
        L->seq = (uint8_t **)realloc(L->seq, L->cap * sizeof(*L->seq));
        L->id = (uint8_t **)realloc(L->id, L->cap * sizeof(*L->id));
        L->seq_len = (size_t *)realloc(L->seq_len, L->cap * sizeof(size_t));
        L->id_len = (size_t *)realloc(L->id_len, L->cap * sizeof(size_t));
    }
    size_t i = L->n++;
    L->seq[i] = (uint8_t *)malloc(r->seq_len ? r->seq_len : 1); memcpy(L->seq[i], r->seq, r->seq_len);
    L->id[i] = (uint8_t *)malloc(r->id_len ? r->id_len : 1); memcpy(L->id[i], r->id, r->id_len);
    L->seq_len[i] = r->seq_len; L->id_len[i] = r->id_len;
}

void orc_library_free(orc_library *L) {
    if (!L) return;
    for (size_t i = 0; i < L->n; i++) { free(L->seq[i]); free(L->id[i]); }
    free(L->seq); free(L->id); free(L->seq_len); free(L->id_len);
    bsmap_free(&L->table);
    free(L);
}

/* library.rs:17-21 from_reader → :89-99 table_from_reader → :79-85 calculate_base_size */
orc_library *orc_library_from_text(const uint8_t *buf, size_t len, int *err) {
    orc_library *L = (orc_library *)calloc(1, sizeof(*L));
    fx_iter it; fx_iter_init(&it, buf, len);
    fx_record r; int rc;
    *err = ORC_OK;
    while ((rc = fx_next(&it, &r)) == 1) lib_push(L, &r);
    if (rc < 0) { *err = rc; orc_library_free(L); return NULL; }
    bsmap_init(&L->table, L->n + 1, 0);
    for (size_t i = 0; i < L->n; i++) {
        int existed;
        size_t s = bsmap_entry(&L->table, L->seq[i], L->seq_len[i], &existed);
        /* library.rs:91-96: map.insert returning Some(_) ⇒ panic "Unexpected duplicate sequence" */
        if (existed) { *err = ORC_E_DUPLICATE_SEQ; orc_library_free(L); return NULL; }
        L->table.val[s] = i;
    }
    /* library.rs:74 get_key_size unwraps keys().next() ⇒ panic on empty (validate_unique_size is
     * vacuously true for 0/1 keys, :65-70) */
    if (L->n == 0) { *err = ORC_E_EMPTY; orc_library_free(L); return NULL; }
    for (size_t i = 1; i < L->n; i++)
        if (L->seq_len[i] != L->seq_len[i - 1]) { *err = ORC_E_INCONSISTENT; orc_library_free(L); return NULL; }
    L->size = L->seq_len[0];
    return L;
}

size_t orc_library_size(const orc_library *L) { return L->size; }
size_t orc_library_n(const orc_library *L) { return L->n; }

/* library.rs:44-46 alias() */
static const uint8_t *lib_alias(const orc_library *L, const uint8_t *tok, size_t n, size_t *id_len) {
    size_t s = bsmap_find(&L->table, tok, n);
    if (s == (size_t)-1) return NULL;
    size_t i = (size_t)L->table.val[s];
    if (id_len) *id_len = L->id_len[i];
    return L->id[i];
}

/* library.rs:34-40 contains(): contains_key then alias */
const uint8_t *orc_library_contains(const orc_library *L, const uint8_t *tok, size_t n, size_t *id_len) {
    if (bsmap_find(&L->table, tok, n) != (size_t)-1) return lib_alias(L, tok, n, id_len);
    return NULL;
}

const uint8_t *orc_library_seq(const orc_library *L, size_t i) { return L->seq[i]; }
const uint8_t *orc_library_id(const orc_library *L, size_t i, size_t *id_len) {
    if (id_len) *id_len = L->id_len[i];
    return L->id[i];
}

/* ------------------------------------------------------------------------- */
/* Permuter — permutes.rs                                                     */
/* ------------------------------------------------------------------------- */

static const uint8_t LEXICON[5] = {'A', 'C', 'G', 'T', 'N'}; /* permutes.rs:3 */

struct orc_permuter {
    bsmap map;   /* child → parent sequence pointer (permutes.rs:5 PermuteMap) */
    bsmap null;  /* permutes.rs:4 NullSet */
    uint8_t *parents; size_t n, L; /* owned copy of the parent sequences */
};

/* permutes.rs:127-144 insert_sequence, with :149-152 insert_to_null and :156-158 insert_to_table */
static void perm_insert_sequence(orc_permuter *P, const uint8_t *sequence, const uint8_t *permutation, size_t L) {
    int existed;
    if (bsmap_find(&P->null, sequence, L) == (size_t)-1) bsmap_entry(&P->null, sequence, L, &existed);
    if (bsmap_find(&P->null, permutation, L) == (size_t)-1) {
        size_t s = bsmap_find(&P->map, permutation, L);
        if (s != (size_t)-1) {
            bsmap_remove_slot(&P->map, s);
            bsmap_entry(&P->null, permutation, L, &existed);
        } else {
            s = bsmap_entry(&P->map, permutation, L, &existed);
            P->map.val[s] = (uint64_t)(uintptr_t)sequence;
        }
    }
}

/* permutes.rs:63-75 build: for each sequence (library order here; the reference iterates
 * HashMap order, which SURVEY §8a shows is unobservable through Counter::assign), :78-85
 * permute_sequence = for idx in 0..len, :96-107 for y in LEXICON if y != seq[idx] */
static void perm_build(orc_permuter *P) {
    size_t L = P->L;
    uint8_t *child = (uint8_t *)malloc(L ? L : 1);
    for (size_t g = 0; g < P->n; g++) {
        const uint8_t *seq = P->parents + g * L;
        for (size_t idx = 0; idx < L; idx++) {
            for (int y = 0; y < 5; y++) {
                if (LEXICON[y] == seq[idx]) continue;
                memcpy(child, seq, L);          /* permutes.rs:111-117 build_permutation */
                child[idx] = LEXICON[y];
                perm_insert_sequence(P, seq, child, L);
            }
        }
    }
    free(child);
}

orc_permuter *orc_permuter_from_seqs(const uint8_t *seqs, size_t n, size_t L) {
    orc_permuter *P = (orc_permuter *)calloc(1, sizeof(*P));
    P->n = n; P->L = L;
    P->parents = (uint8_t *)malloc((n * L) != 0 ? n * L : 1);
    memcpy(P->parents, seqs, n * L);
    bsmap_init(&P->map, n * L * 4 + 16, 1);
    bsmap_init(&P->null, n + 16, 1);
    perm_build(P);
    return P;
}

/* count.rs:48-59 generate_permutations → Permuter::new(library.keys()) */
orc_permuter *orc_permuter_new(const orc_library *lib) {
    size_t n = lib->n, L = lib->size;
    uint8_t *flat = (uint8_t *)malloc((n * L) != 0 ? n * L : 1);
    for (size_t i = 0; i < n; i++) memcpy(flat + i * L, lib->seq[i], L);
    orc_permuter *P = orc_permuter_from_seqs(flat, n, L);
    free(flat);
    return P;
}

void orc_permuter_free(orc_permuter *P) {
    if (!P) return;
    bsmap_free(&P->map); bsmap_free(&P->null); free(P->parents); free(P);
}

/* permutes.rs:55-57 */
const uint8_t *orc_permuter_contains(const orc_permuter *P, const uint8_t *tok, size_t n) {
    size_t s = bsmap_find(&P->map, tok, n);
    return s == (size_t)-1 ? NULL : (const uint8_t *)(uintptr_t)P->map.val[s];
}
size_t orc_permuter_map_len(const orc_permuter *P) { return P->map.n; }
size_t orc_permuter_null_len(const orc_permuter *P) { return P->null.n; }
int orc_permuter_null_contains(const orc_permuter *P, const uint8_t *tok, size_t n) {
    return bsmap_find(&P->null, tok, n) != (size_t)-1;
}

/* ------------------------------------------------------------------------- */
/* Counter — counter.rs                                                       */
/* ------------------------------------------------------------------------- */

struct orc_counter {
    const orc_library *lib; const orc_permuter *perm;
    int reverse; size_t offset, size; int position;
    bsmap results;               /* id → count (counter.rs:18) */
    uint64_t total_reads, matched_reads;
    uint8_t *tok; size_t tok_cap;
};

/* counter.rs:158-180 */
int orc_bounds(size_t seq_len, size_t offset, size_t size, int position, size_t *min, size_t *max) {
    size_t lo, hi;
    if (position == ORC_POS_PLUS) { lo = offset + 1; hi = offset + 1 + size; }
    else if (position == ORC_POS_MINUS) {
        if (offset == 0) return 0;       /* checked_sub(1) == None */
        lo = offset - 1; hi = lo + size;
    } else { lo = offset; hi = offset + size; }
    if (hi > seq_len) return 0;
    *min = lo; *max = hi;
    return 1;
}

/* counter.rs:144-154 apply_trim; :184-192 trim_forward_sequence; :196-204 trim_reverse_sequence.
 * Reverse: the reference materialises record.seq_rev_comp() and slices [min..max] of it, with
 * bounds computed on the forward length.  fxread's complement is UNPINNED by any upstream test;
 * restated as the bit trick `c & 2 ? c ^ 4 : c ^ 21` over the reversed bytes (A<->T, C<->G;
 * 'N' becomes 'J', i.e. a non-ACGT byte never matches after reverse-complementing). */
static int ctr_trim(orc_counter *C, const uint8_t *seq, size_t n, int position) {
    size_t lo, hi;
    if (!orc_bounds(n, C->offset, C->size, position, &lo, &hi)) return 0;
    if (!C->reverse) {
        memcpy(C->tok, seq + lo, C->size);
    } else {
        for (size_t k = lo; k < hi; k++) {
            uint8_t c = seq[n - 1 - k];
            C->tok[k - lo] = (c & 2) ? (uint8_t)(c ^ 4) : (uint8_t)(c ^ 21);
        }
    }
    return 1;
}

/* counter.rs:96-140 assign (recursion unrolled into a loop over Centered→Plus→Minus) */
static const uint8_t *ctr_assign(orc_counter *C, const uint8_t *seq, size_t n, size_t *id_len) {
    int position = C->position;
    for (;;) {
        if (!ctr_trim(C, seq, n, position)) return NULL;      /* :105-108 None ⇒ return None */
        const uint8_t *alias = orc_library_contains(C->lib, C->tok, C->size, id_len); /* :111 */
        if (!alias && C->perm) {                               /* :113-116 */
            const uint8_t *parent = orc_permuter_contains(C->perm, C->tok, C->size);
            if (parent) alias = lib_alias(C->lib, parent, C->size, id_len);
        }
        if (alias) return alias;
        if (position == ORC_POS_CENTERED) position = ORC_POS_PLUS;        /* :123-125 */
        else if (position == ORC_POS_PLUS) position = ORC_POS_MINUS;      /* :128-130 */
        else return NULL;                                                 /* :133 */
    }
}

orc_counter *orc_counter_new(const orc_library *lib, const orc_permuter *perm, int reverse, size_t offset,
                             size_t size, int position_recursion) {
    orc_counter *C = (orc_counter *)calloc(1, sizeof(*C));
    C->lib = lib; C->perm = perm; C->reverse = reverse; C->offset = offset; C->size = size;
    C->position = position_recursion ? ORC_POS_CENTERED : ORC_POS_NULL;   /* counter.rs:44-48 */
    bsmap_init(&C->results, lib->n + 16, 1);
    C->tok_cap = size ? size : 1;
    C->tok = (uint8_t *)malloc(C->tok_cap);
    return C;
}

/* counter.rs:211-236 count(): total += 1 for every record, matched += 1 and results[id] += 1 per hit */
void orc_counter_feed_seq(orc_counter *C, const uint8_t *seq, size_t n) {
    C->total_reads++;
    size_t id_len = 0;
    const uint8_t *id = ctr_assign(C, seq, n, &id_len);
    if (!id) return;
    C->matched_reads++;
    int existed;
    size_t s = bsmap_entry(&C->results, id, id_len, &existed);
    C->results.val[s] += 1;
}

int orc_counter_feed_text(orc_counter *C, const uint8_t *buf, size_t len) {
    fx_iter it; fx_iter_init(&it, buf, len);
    fx_record r; int rc;
    while ((rc = fx_next(&it, &r)) == 1) orc_counter_feed_seq(C, r.seq, r.seq_len);
    return rc < 0 ? rc : ORC_OK;
}

void orc_counter_free(orc_counter *C) {
    if (!C) return;
    bsmap_free(&C->results); free(C->tok); free(C);
}

uint64_t orc_counter_get_value(const orc_counter *C, const uint8_t *id, size_t id_len) {
    size_t s = bsmap_find(&C->results, id, id_len);
    return s == (size_t)-1 ? 0 : C->results.val[s];
}
uint64_t orc_counter_total_reads(const orc_counter *C) { return C->total_reads; }
uint64_t orc_counter_matched_reads(const orc_counter *C) { return C->matched_reads; }
double orc_counter_fraction_mapped(const orc_counter *C) {
    return (double)C->matched_reads / (double)C->total_reads;
}

void orc_counter_table(const orc_counter *C, const orc_library *lib, uint64_t *counts_out) {
    for (size_t i = 0; i < lib->n; i++) counts_out[i] = orc_counter_get_value(C, lib->id[i], lib->id_len[i]);
}

/* count.rs:74-148 reduced to one sample: library → (permuter unless exact) → Counter::new */
int orc_count_text(const uint8_t *lib_buf, size_t lib_len, const uint8_t *reads_buf, size_t reads_len,
                   int reverse, size_t offset, int exact, int position_recursion,
                   uint64_t *counts_out, size_t n_counts, uint64_t *total, uint64_t *matched) {
    int err;
    orc_library *lib = orc_library_from_text(lib_buf, lib_len, &err);
    if (!lib) return err;
    if (n_counts != lib->n) { orc_library_free(lib); return ORC_E_ARG; }
    orc_permuter *perm = exact ? NULL : orc_permuter_new(lib);            /* count.rs:103-107 */
    orc_counter *C = orc_counter_new(lib, perm, reverse, offset, lib->size, position_recursion);
    err = orc_counter_feed_text(C, reads_buf, reads_len);
    if (err == ORC_OK) {
        orc_counter_table(C, lib, counts_out);
        *total = C->total_reads; *matched = C->matched_reads;
    }
    orc_counter_free(C); orc_permuter_free(perm); orc_library_free(lib);
    return err;
}

/* ------------------------------------------------------------------------- */
/* Offsetter — offsetter.rs                                                   */
/* ------------------------------------------------------------------------- */

/* offsetter.rs:42-50 */
static int base_map(uint8_t c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

/* offsetter.rs:55-79 position_counts: (size, 4) matrix, row-major in out[]; size = length of the
 * first record, which is consumed and NOT counted (offsetter.rs:37-39,57). */
int orc_position_counts(const uint8_t *buf, size_t len, size_t take, double *out, size_t cap_rows, size_t *n_out) {
    fx_iter it; fx_iter_init(&it, buf, len);
    fx_record r; int rc;
    if (take == 0) return ORC_E_EMPTY;           /* .take(0) ⇒ next() == None ⇒ expect("empty reader") */
    rc = fx_next(&it, &r);
    if (rc == 0) return ORC_E_EMPTY;
    if (rc < 0) return rc;
    size_t size = r.seq_len, seen = 1;
    if (size > cap_rows) return ORC_E_ARG;
    memset(out, 0, size * 4 * sizeof(double));
    while (seen < take && (rc = fx_next(&it, &r)) == 1) {
        seen++;
        size_t lim = r.seq_len < size ? r.seq_len : size;   /* .take(size) */
        for (size_t i = 0; i < lim; i++) {
            int j = base_map(r.seq[i]);
            if (j >= 0) out[i * 4 + j] += 1.0;
            else { out[i * 4] += 1.0; out[i * 4 + 1] += 1.0; out[i * 4 + 2] += 1.0; out[i * 4 + 3] += 1.0; }
        }
    }
    if (rc < 0) return rc;
    *n_out = size;
    return ORC_OK;
}

/* offsetter.rs:82-87 normalize_counts → :90-95 row entropy
 * (ndarray-stats EntropyExt::entropy = -Σ p ln p with 0·ln 0 = 0). */
int orc_positional_entropy(const uint8_t *buf, size_t len, size_t take, double *out, size_t cap, size_t *n_out) {
    double *m = (double *)calloc(cap * 4 + 4, sizeof(double));
    size_t size = 0;
    int rc = orc_position_counts(buf, len, take, m, cap, &size);
    if (rc != ORC_OK) { free(m); return rc; }
    for (size_t i = 0; i < size; i++) {
        double s = 0.0;
        for (int j = 0; j < 4; j++) s += m[i * 4 + j];   /* sum_axis(Axis(1)) */
        double h = 0.0;
        for (int j = 0; j < 4; j++) {
            double p = m[i * 4 + j] / s;                  /* 0/0 = NaN propagates, as in the reference */
            if (p == 0.0) continue;
            h -= p * log(p);
        }
        out[i] = h;
    }
    free(m);
    *n_out = size;
    return ORC_OK;
}

/* offsetter.rs:109-120 windowed_mse (ndarray-stats mean_sq_err = Σ(a-b)² / n) */
static void windowed_mse(const double *a, size_t na, const double *b, size_t nb, double *out) {
    size_t size = nb - na + 1;
    for (size_t x = 0; x < size; x++) {
        double s = 0.0;
        for (size_t k = 0; k < na; k++) { double d = a[k] - b[x + k]; s += d * d; }
        out[x] = s / (double)na;
    }
}

/* ndarray-stats argmin/min: first minimum; any NaN ⇒ Err ⇒ the reference panics (offsetter.rs:123-141) */
static int argmin_first(const double *v, size_t n, size_t *arg) {
    if (n == 0) return ORC_E_EMPTY;
    size_t a = 0;
    for (size_t i = 0; i < n; i++) {
        if (isnan(v[i])) return ORC_E_NAN;
        if (v[i] < v[a]) a = i;
    }
    *arg = a;
    return ORC_OK;
}

/* offsetter.rs:153-163 minimize_mse + :122-150 assign_offset */
int orc_minimize_mse(const double *ref, size_t n_ref, const double *cmp, size_t n_cmp, int *reverse, size_t *index) {
    if (n_cmp < n_ref) return ORC_E_SHORT;                  /* :154-156 bail */
    size_t size = n_cmp - n_ref + 1;
    double *rev = (double *)malloc(n_cmp * sizeof(double));
    double *mf = (double *)malloc(size * sizeof(double));
    double *mr = (double *)malloc(size * sizeof(double));
    for (size_t i = 0; i < n_cmp; i++) rev[i] = cmp[n_cmp - 1 - i];
    windowed_mse(ref, n_ref, cmp, n_cmp, mf);
    windowed_mse(ref, n_ref, rev, n_cmp, mr);
    size_t af, ar; int rc = argmin_first(mf, size, &af);
    if (rc == ORC_OK) rc = argmin_first(mr, size, &ar);
    if (rc == ORC_OK) {
        if (mf[af] < mr[ar]) { *reverse = 0; *index = af; }   /* :143 strict < */
        else { *reverse = 1; *index = ar; }
    }
    free(rev); free(mf); free(mr);
    return rc;
}

/* offsetter.rs:165-183 entropy_offset (one input; :185-210 entropy_offset_group maps this over inputs) */
int orc_entropy_offset(const uint8_t *lib_buf, size_t lib_len, const uint8_t *reads_buf, size_t reads_len,
                       size_t subsample, int *reverse, size_t *index) {
    size_t cap = 1 << 16, n_ref = 0, n_cmp = 0;
    double *ref = (double *)malloc(cap * sizeof(double));
    double *cmp = (double *)malloc(cap * sizeof(double));
    int rc = orc_positional_entropy(lib_buf, lib_len, (size_t)-1, ref, cap, &n_ref);
    if (rc == ORC_OK) rc = orc_positional_entropy(reads_buf, reads_len, subsample, cmp, cap, &n_cmp);
    if (rc == ORC_OK) rc = orc_minimize_mse(ref, n_ref, cmp, n_cmp, reverse, index);
    free(ref); free(cmp);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* utils.rs — generate_sample_names                                          */
/* ------------------------------------------------------------------------- */

/* Rust str::trim_end_matches(pat): strips the suffix repeatedly */
static size_t trim_end_matches(const char *s, size_t n, const char *pat) {
    size_t m = strlen(pat);
    while (m && n >= m && memcmp(s + n - m, pat, m) == 0) n -= m;
    return n;
}

int orc_generate_sample_names(const char *paths, size_t n, char *out, size_t cap, int *fell_back) {
    /* utils.rs:20-30: basename = split('/').last(); trim ".gz", ".fasta", ".fastq", ".fa", ".fq" in that order */
    const char **base = (const char **)calloc(n ? n : 1, sizeof(*base));
    size_t *blen = (size_t *)calloc(n ? n : 1, sizeof(*blen));
    const char *p = paths;
    for (size_t i = 0; i < n; i++) {
        size_t len = strlen(p);
        const char *slash = NULL;
        for (size_t k = 0; k < len; k++) if (p[k] == '/') slash = p + k;
        const char *b = slash ? slash + 1 : p;
        size_t bl = len - (size_t)(b - p);
        bl = trim_end_matches(b, bl, ".gz");
        bl = trim_end_matches(b, bl, ".fasta");
        bl = trim_end_matches(b, bl, ".fastq");
        bl = trim_end_matches(b, bl, ".fa");
        bl = trim_end_matches(b, bl, ".fq");
        base[i] = b; blen[i] = bl;
        p += len + 1;
    }
    int dup = 0;                                   /* utils.rs:38-43: HashSet size != len */
    for (size_t i = 0; i < n && !dup; i++)
        for (size_t j = i + 1; j < n; j++)
            if (blen[i] == blen[j] && memcmp(base[i], base[j], blen[i]) == 0) { dup = 1; break; }
    if (fell_back) *fell_back = dup;
    size_t w = 0;
    int rc = ORC_OK;
    for (size_t i = 0; i < n; i++) {
        char tmp[32];
        const char *src = base[i]; size_t sl = blen[i];
        if (dup) { sl = (size_t)snprintf(tmp, sizeof(tmp), "Sample.%zu", i); src = tmp; }   /* utils.rs:32-36 */
        if (w + sl + 2 > cap) { rc = ORC_E_ARG; break; }
        if (i) out[w++] = '\n';
        memcpy(out + w, src, sl); w += sl;
    }
    if (rc == ORC_OK) out[w] = 0;
    free((void *)base); free(blen);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* genemap.rs                                                                 */
/* ------------------------------------------------------------------------- */

struct orc_genemap {
    bsmap map;          /* sgrna → index into genes */
    uint8_t **gene; size_t *gene_len; size_t n, cap;
};

void orc_genemap_free(orc_genemap *g) {
    if (!g) return;
    for (size_t i = 0; i < g->n; i++) free(g->gene[i]);
    free(g->gene); free(g->gene_len);
    bsmap_free(&g->map);
    free(g);
}

/* genemap.rs:53-72: for_byte_line (terminator \n or \r\n stripped); split at the first tab */
orc_genemap *orc_genemap_from_text(const uint8_t *buf, size_t len, int *err) {
    orc_genemap *g = (orc_genemap *)calloc(1, sizeof(*g));
    bsmap_init(&g->map, 64, 1);
    *err = ORC_OK;
    size_t p = 0;
    while (p < len) {
        const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', len - p);
        size_t e = nl ? (size_t)(nl - buf) : len, le = e;
        if (le > p && buf[le - 1] == '\r') le--;
        const uint8_t *tab = (const uint8_t *)memchr(buf + p, '\t', le - p);
        if (!tab) { *err = ORC_E_NOTAB; orc_genemap_free(g); return NULL; }
        size_t gl = (size_t)(tab - (buf + p));
        const uint8_t *sg = tab + 1; size_t sl = le - (size_t)(sg - buf);
        int existed;
        size_t slot = bsmap_entry(&g->map, sg, sl, &existed);
        if (existed) { *err = ORC_E_DUPKEY; orc_genemap_free(g); return NULL; }
        if (g->n == g->cap) {
            g->cap = g->cap ? g->cap * 2 : 256;
            g->gene = (uint8_t **)realloc(g->gene, g->cap * sizeof(*g->gene));
            g->gene_len = (size_t *)realloc(g->gene_len, g->cap * sizeof(size_t));
        }
        g->gene[g->n] = (uint8_t *)malloc(gl ? gl : 1); memcpy(g->gene[g->n], buf + p, gl);
        g->gene_len[g->n] = gl;
        g->map.val[slot] = g->n++;
        p = e + 1;
    }
    return g;
}

const uint8_t *orc_genemap_get(const orc_genemap *g, const uint8_t *sgrna, size_t n, size_t *gene_len) {
    size_t s = bsmap_find(&g->map, sgrna, n);
    if (s == (size_t)-1) return NULL;
    if (gene_len) *gene_len = g->gene_len[g->map.val[s]];
    return g->gene[g->map.val[s]];
}

long orc_genemap_missing(const orc_genemap *g, const orc_library *lib) {
    for (size_t i = 0; i < lib->n; i++)
        if (!orc_genemap_get(g, lib->id[i], lib->id_len[i], NULL)) return (long)i;
    return -1;
}

/* ------------------------------------------------------------------------- */
/* results.rs                                                                 */
/* ------------------------------------------------------------------------- */

int orc_generate_columns(const char *names, size_t n, int with_genemap, char *out, size_t cap) {
    size_t w = (size_t)snprintf(out, cap, "Guide");                /* results.rs:36 */
    const char *p = names;
    for (size_t i = 0; i < n; i++) {
        if (i == 0 && with_genemap) w += (size_t)snprintf(out + w, w < cap ? cap - w : 0, "\tGene");   /* :37-39 */
        w += (size_t)snprintf(out + w, w < cap ? cap - w : 0, "\t%s", p);                               /* :40 */
        p += strlen(p) + 1;
        if (w >= cap) return ORC_E_ARG;
    }
    return ORC_OK;
}

long orc_format_results(const orc_library *lib, const uint64_t *counts, size_t n_samples, const char *names,
                        const orc_genemap *genemap, int include_zero, char *out, size_t cap) {
    /* Counter per sample: results keyed by id (counter.rs:232-235) */
    bsmap *res = (bsmap *)calloc(n_samples ? n_samples : 1, sizeof(bsmap));
    for (size_t s = 0; s < n_samples; s++) {
        bsmap_init(&res[s], lib->n + 16, 0);
        for (size_t i = 0; i < lib->n; i++) {
            uint64_t c = counts[s * lib->n + i];
            if (!c) continue;
            int existed;
            size_t slot = bsmap_entry(&res[s], lib->id[i], lib->id_len[i], &existed);
            res[s].val[slot] += c;
        }
    }
    long rc = 0;
    if (orc_generate_columns(names, n_samples, genemap != NULL, out, cap) != ORC_OK) { rc = ORC_E_ARG; goto done; }
    {
        size_t w = strlen(out);
        if (w + 2 > cap) { rc = ORC_E_ARG; goto done; }
        out[w++] = '\n';
        for (size_t i = 0; i < lib->n; i++) {                      /* results.rs:79 library.values() */
            size_t row0 = w;
            uint64_t total = 0;
            if (w + lib->id_len[i] + 2 > cap) { rc = ORC_E_ARG; goto done; }
            memcpy(out + w, lib->id[i], lib->id_len[i]); w += lib->id_len[i];
            for (size_t s = 0; s < n_samples; s++) {
                if (s == 0 && genemap) {                            /* results.rs:46-62 append_gene */
                    size_t gl = 0;
                    const uint8_t *gene = orc_genemap_get(genemap, lib->id[i], lib->id_len[i], &gl);
                    if (!gene) { rc = ORC_E_NOGENE; goto done; }
                    if (w + gl + 2 > cap) { rc = ORC_E_ARG; goto done; }
                    out[w++] = '\t'; memcpy(out + w, gene, gl); w += gl;
                }
                size_t slot = bsmap_find(&res[s], lib->id[i], lib->id_len[i]);
                uint64_t c = slot == (size_t)-1 ? 0 : res[s].val[slot];          /* counter.rs:71-76 */
                if (w + 24 > cap) { rc = ORC_E_ARG; goto done; }
                w += (size_t)snprintf(out + w, cap - w, "\t%llu", (unsigned long long)c);   /* results.rs:65-67 */
                total += c;
            }
            if (include_zero || total > 0) out[w++] = '\n';        /* results.rs:90-94 */
            else w = row0;
        }
        out[w] = 0;
        rc = (long)w;
    }
done:
    for (size_t s = 0; s < n_samples; s++) bsmap_free(&res[s]);
    free(res);
    return rc;
}
