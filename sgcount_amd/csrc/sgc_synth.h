// sgc_synth.h — the synthetic read model, shared verbatim by the host and gfx950 generators so that
// both produce the same bytes.  See include/sgcount_synth.h for the workload definition.
#pragma once
#include <stdint.h>

#include "sgc_format.h"   // SGC_HD

#define SGS_LEN 150u
#define SGS_P0 30u
#define SGS_SCAF_LEN 128u

// constant 5' adapter/stagger (30 bp) and 3' scaffold (128 bp; tracrRNA-like, repeated)
#define SGS_PREFIX_STR "TCTTGTGGAAAGGACGAAACACCGGTACCG"
#define SGS_SCAFFOLD_STR                                                                                     \
    "GTTTTAGAGCTAGAAATAGCAAGTTAAAATAAGGCTAGTCCGTTATCAACTTGAAAAAGTGGCACCGAGTCGGTGCTTTTTTGAATTCGCTAGCTAGGTCTTGA" \
    "AAGGAGTGGGAATTGGCTCCGGTGC"

SGC_HD uint8_t sgs_prefix_at(uint32_t k) {
    const char *s = SGS_PREFIX_STR;
    return (uint8_t)s[k];
}
SGC_HD uint8_t sgs_scaffold_at(uint32_t k) {
    const char *s = SGS_SCAFFOLD_STR;
    return (uint8_t)s[k];
}

SGC_HD uint64_t sgs_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct sgs_spec {
    uint32_t cls;        // 0 exact 1 sub 2 N 3 ins 4 del 5 junk 6 trunc
    uint32_t gid;        // guide drawn (meaningless for junk)
    uint32_t pos;        // mutated position (sub / N)
    uint32_t alt;        // 1..3 rotation of the base (sub)
    uint32_t P;          // guide start = prefix length
    uint32_t len;        // read length
    uint64_t junk;       // random L-mer bits (junk) / random leading stagger bases
    uint64_t lead;       // random bases in front of the constant prefix when P > 30
};

SGC_HD uint32_t sgs_pick_guide(uint64_t r, uint32_t n) {
    // every 100th guide (index % 100 == 0) carries 50x weight
    const uint64_t n_hot = ((uint64_t)n + 99) / 100, n_cold = n - n_hot;
    const uint64_t W = n_hot * 50 + n_cold;
    const uint64_t x = r % W;
    if (x < n_hot * 50) return (uint32_t)((x / 50) * 100);
    const uint64_t c = x - n_hot * 50;
    return (uint32_t)(c + c / 99 + 1);
}

SGC_HD void sgs_make_spec(uint64_t seed, uint64_t i, uint32_t n_guides, uint32_t L, uint32_t mode, sgs_spec &sp) {
    uint64_t s = sgs_mix(seed + 0x632BE59BD9B4E019ull * (i + 1));
    const uint64_t d0 = sgs_mix(s += 0x9E3779B97F4A7C15ull);
    const uint64_t d1 = sgs_mix(s += 0x9E3779B97F4A7C15ull);
    const uint64_t d2 = sgs_mix(s += 0x9E3779B97F4A7C15ull);
    const uint64_t d3 = sgs_mix(s += 0x9E3779B97F4A7C15ull);
    const uint64_t d4 = sgs_mix(s += 0x9E3779B97F4A7C15ull);
    const uint32_t u = (uint32_t)(d0 % 100);
    sp.cls = u < 85 ? 0 : u < 90 ? 1 : u < 91 ? 2 : u < 93 ? 3 : u < 95 ? 4 : u < 99 ? 5 : 6;
    sp.gid = n_guides ? sgs_pick_guide(d1, n_guides) : 0;
    // mode bits 8..14 (SGS_MODE_DOMINANT(pct)): that percentage of the reads takes ONE guide (index 7 % n) instead — a sample a single
    // guide dominates, the worst case for anything that shares the reads out by guide
    if (n_guides && ((mode >> 8) & 127u) && (uint32_t)((d1 >> 40) % 100) < ((mode >> 8) & 127u)) sp.gid = 7u % n_guides;
    sp.pos = (uint32_t)(d2 % L);
    sp.alt = 1 + (uint32_t)((d2 >> 32) % 3);
    sp.junk = d3;
    sp.lead = d4;
    uint32_t P = SGS_P0;
    if ((mode & 255u) == 1) {  // stagger: 28..32 with weights 5/10/70/10/5
        const uint32_t v = (uint32_t)((d0 >> 32) % 100);
        P = v < 5 ? 28 : v < 15 ? 29 : v < 85 ? 30 : v < 95 ? 31 : 32;
    }
    if (sp.cls == 3) P += 1;
    if (sp.cls == 4) P -= 1;
    sp.P = P;
    sp.len = sp.cls == 6 ? P + L / 2 : SGS_LEN;
}

SGC_HD uint8_t sgs_acgt(uint32_t c) { return (uint8_t)("ACGT"[c & 3]); }

// byte j (< sp.len) of the read; lib = n_guides x L ASCII
SGC_HD uint8_t sgs_byte(const sgs_spec &sp, const uint8_t *lib, uint32_t L, uint32_t j) {
    if (j < sp.P) {
        if (sp.P > SGS_P0) {
            const uint32_t extra = sp.P - SGS_P0;
            if (j < extra) return sgs_acgt((uint32_t)(sp.lead >> (2 * j)));
            return sgs_prefix_at(j - extra);
        }
        return sgs_prefix_at(SGS_P0 - sp.P + j);
    }
    const uint32_t k = j - sp.P;
    if (k < L) {
        if (sp.cls == 5) return sgs_acgt((uint32_t)(sp.junk >> (2 * k)));
        const uint8_t b = lib[(uint64_t)sp.gid * L + k];
        if (k == sp.pos) {
            if (sp.cls == 2) return (uint8_t)'N';
            if (sp.cls == 1) return sgs_acgt(sgc_base_code(b) + sp.alt);
        }
        return b;
    }
    return sgs_scaffold_at((k - L) % SGS_SCAF_LEN);
}

SGC_HD uint32_t sgs_digits(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}
// FASTQ record of read i: "@r<i>\n" seq "\n+\n" qual "\n"
SGC_HD uint32_t sgs_fastq_record_len(uint64_t i, uint32_t len) { return 2 + sgs_digits(i) + 1 + len + 3 + len + 1; }
