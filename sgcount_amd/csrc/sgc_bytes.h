// sgc_bytes.h — the generic byte-string count path: libraries the 2-bit path cannot represent.
//
// The reference compares raw bytes (src/library.rs:34-46, src/permutes.rs:3,78-117): any byte is a legal library symbol,
// 'N' is the fifth letter of the permute lexicon, and a guide may have any length.  The packed path (sgc_format.h) needs an
// ACGT-only library of at most 30 bases; everything else is served here, by the same rules on bytes:
//   * library table   hash64(sequence) -> guide, verified byte for byte against the library strings resident in HBM
//   * children table  hash64(child) -> (parent guide, position), one entry per child string that exactly ONE guide generates
//                     (substituting one position by one of A C G T N, src/permutes.rs:78-117) and that is not itself a library
//                     sequence (src/permutes.rs:127-144: parents and multiply generated children are nulled) — the same
//                     observable rule as SURVEY §8a; a hit is verified: the window equals the parent except at that position, where it holds the child's letter
// Both tables are open-addressed on the 64-bit hash with linear probing; equal hashes of different strings simply occupy
// different slots and are told apart by the verification, so the outcome is exact, not probabilistic.
// One lane per read walks Counter::assign's chain (src/counter.rs:96-140) on the read's bytes.  This is a fallback: global
// probes and global atomics, no partitioning — correctness first (tests/test_generic_gpu.py), ~10x slower than the packed path.
#pragma once
#include <stdint.h>

#include "sgc_format.h"

#define SGC_BYTES_MAXL 65535u
#define SGC_BYTES_EMPTY 0xFFFFFFFFFFFFFFFFull

struct sgc_bytes_view {
    const uint8_t *seqs;        // n x L library strings
    const uint64_t *lib_tag;    // 2^lib_log2 hashes (SGC_BYTES_EMPTY = free)
    const uint32_t *lib_val;    // guide index per slot
    const uint64_t *perm_tag;   // 2^perm_log2 hashes of the unambiguous children (NULL in exact mode)
    const uint32_t *perm_val;   // parent guide per slot
    const uint32_t *perm_pl;    // position of the substitution | substituted letter << 24, per slot
    uint32_t n, L, lib_log2, perm_log2;
};

// FNV-1a over the bytes, with a final avalanche so that the top bits (the slot) depend on every byte
SGC_HD uint64_t sgc_bytes_hash_step(uint64_t h, uint8_t b) { return (h ^ b) * 0x100000001b3ull; }
SGC_HD uint64_t sgc_bytes_hash_init() { return 0xcbf29ce484222325ull; }
SGC_HD uint64_t sgc_bytes_hash_fin(uint64_t h) {
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h == SGC_BYTES_EMPTY ? 0 : h;
}
SGC_HD uint32_t sgc_bytes_slot(uint64_t h, uint32_t log2) { return (uint32_t)(h >> (64 - log2)); }
