// sgc_tables.h — host-side construction of the device hash tables.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "sgc_format.h"

struct sgc_host_table {
    std::vector<uint64_t> slots;
    std::vector<uint32_t> vals;   // split layout only
    uint32_t log2_slots = 0;
    uint32_t gid_bits = 0;        // 0 => split layout
    uint32_t log2_slice = 0;      // probing wraps inside slices of 2^log2_slice slots
    uint32_t core_cl = 0;         // slices follow the core hash (sgc_format.h sgc_home_bucket_ex); 0 = full-key hash
    uint64_t entries = 0;
};

// Packs an ASCII guide (ACGT only) into a 2-bit key, base j at bits [2j, 2j+2).  Returns false on other bytes.
bool sgc_pack_key(const uint8_t *seq, uint32_t L, uint64_t &key);

// Library::table_from_reader (reference src/library.rs:89-99).  Returns 0, or SGC_E_DUPLICATE /
// SGC_E_UNSUPPORTED (include/sgcount_hip.h) with `err` set.
// max_log2_slice bounds the slice size (the LDS budget of the partitioned path); 0 = unpartitioned.
// core_cl > 0 asks for slices that follow the core hash; a key set that does not spread that way (guides sharing
// those bases) gets full-key-hash slices instead (out.core_cl says which).
int sgc_build_library_table(const uint8_t *seqs, uint32_t n, uint32_t L, uint32_t max_log2_slice, uint32_t core_cl,
                            std::vector<uint64_t> &keys, sgc_host_table &out, std::string &err);

// Permuter::build (reference src/permutes.rs:63-75,127-158) restricted to what Counter::assign can
// observe: child -> the UNIQUE guide at Hamming distance 1; children that are library members or have
// two or more parents are dropped.  Only ACGT substitutions are stored; the 'N' children are resolved
// in-kernel by probing the library (sgc_kernels.hip window_assign).
// `amb` (optional): 2 u64 per guide, bit 4j + b set iff the child "guide with base b at position j" has two
// or more parents (or is itself a library member), i.e. is NOT a key of the table.
void sgc_build_permute_table(const std::vector<uint64_t> &keys, uint32_t L, const sgc_host_table &lib,
                             sgc_host_table &out, std::vector<uint64_t> *child_keys = nullptr,
                             std::vector<uint64_t> *amb = nullptr);

// Core index over span bases [cs, cs + cl) (sgc_format.h sgc_core_view).  Returns false when the guides do
// not spread over <= 2^SGC_CORE_MAX_LOG2_P partitions of at most SGC_CORE_EMAX entries (the caller then keeps the
// probing resolver).
struct sgc_host_core {
    std::vector<uint64_t> ents;     // 2^log2_p x SGC_CORE_EMAX
    std::vector<uint32_t> gids;     // parallel to ents
    std::vector<uint16_t> starts;   // 2^log2_p x SGC_CORE_STARTS
    uint32_t log2_p = 0, cs = 0, cl = 0;
};
bool sgc_build_core_index(const std::vector<uint64_t> &keys, uint32_t L, uint32_t cs, uint32_t cl, sgc_host_core &out);

// Two-choice (cuckoo) image of every slice of a packed, sliced library table (sgc_format.h sgc_cuckoo_alt): same size and slot
// format as lib.slots.  Returns false if some slice cannot be placed (the caller then keeps the open-addressed probe).
bool sgc_build_slice_cuckoo(const sgc_host_table &lib, std::vector<uint64_t> &out);

// Generic byte-string tables (sgc_bytes.h) for a library of n sequences of L arbitrary bytes.  Returns 0 or SGC_E_DUPLICATE
// (src/library.rs:91-96) with `err` set.  perm_* stay empty unless one_mm.
struct sgc_host_bytes {
    std::vector<uint64_t> lib_tag, perm_tag;
    std::vector<uint32_t> lib_val, perm_val;
    std::vector<uint32_t> perm_pl;
    uint32_t lib_log2 = 0, perm_log2 = 0;
    uint64_t perm_entries = 0;
};
int sgc_build_bytes_tables(const uint8_t *seqs, uint32_t n, uint32_t L, bool one_mm, sgc_host_bytes &out, std::string &err);

// Rest filter of a core (sgc_format.h sgc_core_view::filt): 3 x 2^log2_bits bits, as 32-bit words.
uint32_t sgc_rest_filter_log2(uint32_t n_guides);
void sgc_build_rest_filter(const std::vector<uint64_t> &keys, uint32_t cs, uint32_t cl, uint32_t log2_bits, std::vector<uint32_t> &out);

// log2 of the slot count of a single-mismatch table sized for n_children (all 3 L children per guide; load <= 0.5)
uint32_t sgc_permute_log2_slots(uint64_t n_children);

// Blocked Bloom filter over `keys` with 2^log2_words 64-bit words (sgc_format.h sgc_bloom_*).
void sgc_build_bloom(const std::vector<uint64_t> &keys, uint32_t log2_words, std::vector<uint64_t> &out);
uint32_t sgc_bloom_log2_words(uint64_t n_keys, uint32_t bits_per_key, uint32_t min_log2, uint32_t max_log2);
