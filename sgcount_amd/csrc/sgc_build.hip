// sgc_build.hip — device-side construction of the single-mismatch table (Permuter::build, reference
// src/permutes.rs:63-158, restricted to what Counter::assign can observe), its Bloom filter and the per-guide
// ambiguity masks of the core resolver.  gfx950.
//
// The host version (sgc_tables.cpp sgc_build_permute_table) spends ~0.6 s on 6 M children of a 100k-guide
// library — the largest fixed cost of a short run.  Here: one kernel writes every (child, parent) pair, rocPRIM
// sorts the pairs by child, and a second kernel looks at each child's neighbours in the sorted order:
//   alone and not a library member  -> the child has exactly one parent: insert (child -> parent) into the
//                                      open-addressed table (atomicCAS on the slots), set its Bloom bits
//   otherwise                        -> two or more parents (src/permutes.rs:127-144 moves it to `null`), or a
//                                      library member: set the ambiguity bit (4 j + b) of every parent
// Insertion order is whatever the race gives; a lookup scans from the home bucket to the first free slot, which
// is independent of that order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "sgc_bytes.h"
#include "sgc_device.h"
#include "sgc_format.h"
#include "sgc_kernels.h"

__global__ void __launch_bounds__(256) k_gen_children(const uint64_t *__restrict__ keys, uint32_t n, uint32_t L,
                                                      uint64_t *__restrict__ ck, uint32_t *__restrict__ cg) {
    const uint64_t total = (uint64_t)n * 3 * L;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = (uint32_t)(i / (3 * L)), r = (uint32_t)(i % (3 * L)), j = r / 3;
        const uint64_t d = r % 3 + 1;                       // 3 ACGT substitutions per position
        ck[i] = keys[g] ^ (d << (2 * j));
        cg[i] = g;
    }
}

__global__ void __launch_bounds__(256) k_finish_children(const uint64_t *__restrict__ ck, const uint32_t *__restrict__ cg,
                                                         uint64_t total, const uint64_t *__restrict__ keys,
                                                         sgc_table_view lib, uint64_t *__restrict__ slots,
                                                         uint32_t log2_slots, uint32_t gid_bits, uint64_t *__restrict__ bloom,
                                                         uint32_t bloom_log2, unsigned long long *__restrict__ amb,
                                                         unsigned long long *__restrict__ n_entries) {
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = ck[i];
        const uint32_t g = cg[i];
        const bool dup = (i > 0 && ck[i - 1] == k) || (i + 1 < total && ck[i + 1] == k);
        if (!dup && table_find<true>(lib, k) == SGC_NONE) {
            const uint64_t packed = (k << gid_bits) | g;
            uint32_t b = sgc_home_bucket(k, log2_slots);
            for (;;) {
                if (atomicCAS((unsigned long long *)&slots[2ull * b], (unsigned long long)SGC_EMPTY, (unsigned long long)packed) == SGC_EMPTY) break;
                if (atomicCAS((unsigned long long *)&slots[2ull * b + 1], (unsigned long long)SGC_EMPTY, (unsigned long long)packed) == SGC_EMPTY) break;
                b = sgc_next_bucket(b, log2_slots);
            }
            const uint64_t h = sgc_hash2(k);
            atomicOr((unsigned long long *)&bloom[sgc_bloom_word(h, bloom_log2)], (unsigned long long)sgc_bloom_mask(h));
            mine++;
        } else {
            const uint64_t x = k ^ keys[g];                 // one differing base
            const uint32_t j = (uint32_t)__builtin_ctzll(x) >> 1, bit = 4 * j + (uint32_t)((k >> (2 * j)) & 3);
            atomicOr(&amb[2ull * g + (bit >> 6)], 1ull << (bit & 63));
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_entries, (unsigned long long)mine);
}

// d_slots: 2^log2_slots u64 preset to SGC_EMPTY; d_bloom: 2^bloom_log2 u64 of zeros; d_amb: 2 n u64 of zeros;
// d_scratch: sgc_device_build_scratch_bytes(n, L) bytes.  Asynchronous on `st`; *d_entries (a device counter,
// zeroed by the caller) receives the number of table entries.
size_t sgc_device_build_scratch_bytes(uint32_t n, uint32_t L) {
    const size_t total = (size_t)n * 3 * L;
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    total, 0, 2 * L, (hipStream_t)0);
    return total * (8 + 8 + 4 + 4) + ((tmp + 255) & ~(size_t)255) + 1024;
}

int sgc_device_build_permute(hipStream_t st, const uint64_t *d_keys, uint32_t n, uint32_t L, const sgc_table_view &lib,
                             uint64_t *d_slots, uint32_t log2_slots, uint32_t gid_bits, uint64_t *d_bloom,
                             uint32_t bloom_log2, uint64_t *d_amb, unsigned long long *d_entries, void *d_scratch) {
    const size_t total = (size_t)n * 3 * L;
    uint64_t *ck_in = (uint64_t *)d_scratch, *ck_out = ck_in + total;
    uint32_t *cg_in = (uint32_t *)(ck_out + total), *cg_out = cg_in + total;
    void *tmp = (void *)(((uintptr_t)(cg_out + total) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, ck_in, ck_out, cg_in, cg_out, total, 0, 2 * L, st) != hipSuccess) return -1;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_gen_children, dim3(grid), dim3(256), 0, st, d_keys, n, L, ck_in, cg_in);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, ck_in, ck_out, cg_in, cg_out, total, 0, 2 * L, st) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_finish_children, dim3(grid), dim3(256), 0, st, ck_out, cg_out, (uint64_t)total, d_keys, lib, d_slots,
                       log2_slots, gid_bits, d_bloom, bloom_log2, (unsigned long long *)d_amb, d_entries);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Marks, in a core index's guide ids, the guides that have an ambiguous child (any bit of their mask set): bit 31 of the id.
// k_core reads the id from LDS anyway and gathers a guide's mask from global memory only when that bit is set — a few
// hundred guides of 100k have a neighbour within two substitutions.
__global__ void __launch_bounds__(256) k_flag_ambiguous(uint32_t *__restrict__ gids, uint64_t n_entries, const ulonglong2 *__restrict__ amb) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_entries) return;
    const uint32_t g = gids[i];
    if (g == SGC_NONE) return;
    const ulonglong2 m = amb[g];
    if (m.x | m.y) gids[i] = g | 0x80000000u;
}

void sgc_flag_ambiguous(hipStream_t st, uint32_t *d_gids, uint64_t n_entries, const uint64_t *d_amb) {
    if (!n_entries) return;
    hipLaunchKernelGGL(k_flag_ambiguous, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, st, d_gids, n_entries,
                       reinterpret_cast<const ulonglong2 *>(d_amb));
}

// ---- the children table of the byte-string path (sgc_bytes.h), built on the device ------------------------------------------
// The host version (sgc_tables.cpp sgc_build_bytes_tables) hashes, sorts and screens 8.0 M child strings for a 100k-guide library of
// 20 bases: 0.5 s on eight threads — most of the start-up of a run with a library that has guides outside ACGT.  Here, as for the
// packed single-mismatch table above: one kernel writes (hash of the child string, parent | position | letter) for every child,
// rocPRIM sorts the pairs by hash, and a second kernel keeps a child iff no other pair in its run of equal hashes spells the same
// string (src/permutes.rs:127-144: a child of two parents is nulled) and the string is no library member (:149-152), and inserts it
// (atomicCAS on the tag array; linear probing — a lookup verifies the string, so the order of equal hashes does not matter).
#define BY_G_BITS 22u
#define BY_J_BITS 7u
__device__ __forceinline__ uint8_t by_lex(uint32_t i) { return i == 0 ? 'A' : i == 1 ? 'C' : i == 2 ? 'G' : i == 3 ? 'T' : 'N'; }

__global__ void __launch_bounds__(256) k_bytes_children(const uint8_t *__restrict__ seqs, uint32_t n, uint32_t L, uint64_t *__restrict__ hs,
                                                        uint32_t *__restrict__ pay) {
    const uint64_t total = (uint64_t)n * L * 5u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = (uint32_t)(i / (5u * L)), r = (uint32_t)(i % (5u * L)), j = r / 5u, bi = r % 5u;
        const uint8_t *w = seqs + (size_t)g * L;
        const uint8_t b = by_lex(bi);
        uint64_t h = SGC_BYTES_EMPTY;                           // the letter the guide already has there: no child (sorts to the end)
        if (b != w[j]) {
            h = sgc_bytes_hash_init();
            for (uint32_t k = 0; k < L; k++) h = sgc_bytes_hash_step(h, k == j ? b : w[k]);
            h = sgc_bytes_hash_fin(h);
        }
        hs[i] = h;
        pay[i] = g | (j << BY_G_BITS) | (bi << (BY_G_BITS + BY_J_BITS));
    }
}

__device__ __forceinline__ bool by_same_child(const uint8_t *seqs, uint32_t L, uint32_t pa, uint32_t pb) {
    const uint32_t ga = pa & ((1u << BY_G_BITS) - 1u), ja = (pa >> BY_G_BITS) & ((1u << BY_J_BITS) - 1u), gb = pb & ((1u << BY_G_BITS) - 1u),
                   jb = (pb >> BY_G_BITS) & ((1u << BY_J_BITS) - 1u);
    const uint8_t ba = by_lex(pa >> (BY_G_BITS + BY_J_BITS)), bb = by_lex(pb >> (BY_G_BITS + BY_J_BITS));
    const uint8_t *x = seqs + (size_t)ga * L, *y = seqs + (size_t)gb * L;
    for (uint32_t k = 0; k < L; k++)
        if ((k == ja ? ba : x[k]) != (k == jb ? bb : y[k])) return false;
    return true;
}

__global__ void __launch_bounds__(256) k_bytes_children_finish(const uint64_t *__restrict__ hs, const uint32_t *__restrict__ pay, uint64_t total,
                                                               sgc_bytes_view v, uint64_t *__restrict__ perm_tag, uint32_t *__restrict__ perm_val,
                                                               uint32_t *__restrict__ perm_pl, uint32_t perm_log2,
                                                               unsigned long long *__restrict__ n_entries) {
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hs[i];
        if (h == SGC_BYTES_EMPTY) continue;
        const uint32_t p = pay[i];
        bool unique = true;
        for (uint64_t k = i; k > 0 && hs[k - 1] == h && unique; k--) unique = !by_same_child(v.seqs, v.L, pay[k - 1], p);
        for (uint64_t k = i + 1; k < total && hs[k] == h && unique; k++) unique = !by_same_child(v.seqs, v.L, pay[k], p);
        if (!unique) continue;
        const uint32_t g = p & ((1u << BY_G_BITS) - 1u), j = (p >> BY_G_BITS) & ((1u << BY_J_BITS) - 1u);
        const uint8_t b = by_lex(p >> (BY_G_BITS + BY_J_BITS));
        const uint8_t *w = v.seqs + (size_t)g * v.L;
        // a library member? (Library::contains on the child string)
        bool member = false;
        const uint32_t lmask = (1u << v.lib_log2) - 1u;
        for (uint32_t s = sgc_bytes_slot(h, v.lib_log2); !member; s = (s + 1u) & lmask) {
            const uint64_t tag = v.lib_tag[s];
            if (tag == SGC_BYTES_EMPTY) break;
            if (tag != h) continue;
            const uint8_t *q = v.seqs + (size_t)v.lib_val[s] * v.L;
            bool same = true;
            for (uint32_t k = 0; k < v.L && same; k++) same = q[k] == (k == j ? b : w[k]);
            member = same;
        }
        if (member) continue;
        const uint32_t pmask = (1u << perm_log2) - 1u;
        for (uint32_t s = sgc_bytes_slot(h, perm_log2);; s = (s + 1u) & pmask) {
            if (atomicCAS((unsigned long long *)&perm_tag[s], (unsigned long long)SGC_BYTES_EMPTY, (unsigned long long)h) == SGC_BYTES_EMPTY) {
                perm_val[s] = g; perm_pl[s] = j | ((uint32_t)b << 24);
                break;
            }
        }
        mine++;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_entries, (unsigned long long)mine);
}

bool sgc_device_bytes_children_supported(uint32_t n, uint32_t L) { return n < (1u << BY_G_BITS) && L < (1u << BY_J_BITS) && (uint64_t)n * L * 5u < (1ull << 31); }

size_t sgc_device_bytes_children_scratch(uint32_t n, uint32_t L) {
    const size_t total = (size_t)n * L * 5u;
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, total, 0, 64,
                                    (hipStream_t)0);
    return total * (8 + 8 + 4 + 4) + ((tmp + 255) & ~(size_t)255) + 1024;
}

// v: the library strings and the library table on the device (perm fields unused); perm_*: 2^perm_log2 slots, tags preset to
// SGC_BYTES_EMPTY; *d_entries (zeroed) receives the number of children kept.  Asynchronous on st.
int sgc_device_bytes_children(hipStream_t st, const sgc_bytes_view &v, uint64_t *perm_tag, uint32_t *perm_val, uint32_t *perm_pl, uint32_t perm_log2,
                              unsigned long long *d_entries, void *d_scratch) {
    const size_t total = (size_t)v.n * v.L * 5u;
    uint64_t *h_in = (uint64_t *)d_scratch, *h_out = h_in + total;
    uint32_t *p_in = (uint32_t *)(h_out + total), *p_out = p_in + total;
    void *tmp = (void *)(((uintptr_t)(p_out + total) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, h_in, h_out, p_in, p_out, total, 0, 64, st) != hipSuccess) return -1;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_bytes_children, dim3(grid), dim3(256), 0, st, v.seqs, v.n, v.L, h_in, p_in);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, h_in, h_out, p_in, p_out, total, 0, 64, st) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_bytes_children_finish, dim3(grid), dim3(256), 0, st, h_out, p_out, (uint64_t)total, v, perm_tag, perm_val, perm_pl, perm_log2,
                       d_entries);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
