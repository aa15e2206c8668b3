// sgc_synth.hip — synthetic workload generator, host and gfx950 (include/sgcount_synth.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/sgcount_synth.h"
#include "sgc_synth.h"

static_assert(sizeof(SGS_PREFIX_STR) - 1 == SGS_P0, "prefix must be 30 bp");
static_assert(sizeof(SGS_SCAFFOLD_STR) - 1 >= SGS_SCAF_LEN, "scaffold must be >= 128 bp");

static thread_local std::string g_serr;
static int sfail(const std::string &m) { g_serr = m; return -1; }

// ---- xoshiro256** (library generation, host only) ---------------------------------------------
struct xoshiro {
    uint64_t s[4];
    explicit xoshiro(uint64_t seed) {
        for (int k = 0; k < 4; k++) s[k] = sgs_mix(seed += 0x9E3779B97F4A7C15ull);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
};

static void key_to_ascii(uint64_t key, uint32_t L, uint8_t *out) {
    for (uint32_t j = 0; j < L; j++) out[j] = sgs_acgt((uint32_t)(key >> (2 * j)));
}

// ---- kernels --------------------------------------------------------------------------------------
__global__ void k_read_lens(uint64_t seed, uint64_t first, uint64_t n, uint32_t L, uint32_t mode,
                            uint32_t *__restrict__ lens, int fastq) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    sgs_spec sp;
    sgs_make_spec(seed, first + t, 0, L, mode, sp);
    lens[t] = fastq ? sgs_fastq_record_len(first + t, sp.len) : sp.len;
}

// one lane per output byte: lane (i, j) writes byte j of read i; 160 lanes per read
#define SGS_LANES_PER_READ 160u
__global__ void k_reads_fill(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *__restrict__ lib,
                             uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *__restrict__ offsets,
                             uint8_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / SGS_LANES_PER_READ;
    const uint32_t j = (uint32_t)(t % SGS_LANES_PER_READ);
    if (i >= n) return;
    sgs_spec sp;
    sgs_make_spec(seed, first + i, n_guides, L, mode, sp);
    if (j < sp.len) out[offsets[i] + j] = sgs_byte(sp, lib, L, j);
}

#define SGS_LANES_PER_REC 384u
__device__ __host__ inline uint8_t sgs_fastq_byte(const sgs_spec &sp, uint64_t idx, const uint8_t *lib, uint32_t L,
                                                  uint32_t j) {
    const uint32_t nd = sgs_digits(idx);
    if (j == 0) return '@';
    if (j == 1) return 'r';
    if (j < 2 + nd) {
        uint64_t v = idx;
        for (uint32_t k = 0; k < nd - 1 - (j - 2); k++) v /= 10;
        return (uint8_t)('0' + v % 10);
    }
    uint32_t k = j - (2 + nd);
    if (k == 0) return '\n';
    k -= 1;
    if (k < sp.len) return sgs_byte(sp, lib, L, k);
    k -= sp.len;
    if (k == 0) return '\n';
    if (k == 1) return '+';
    if (k == 2) return '\n';
    k -= 3;
    if (k < sp.len) return 'I';
    return '\n';
}

__global__ void k_fastq_fill(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *__restrict__ lib,
                             uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *__restrict__ offsets,
                             uint8_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / SGS_LANES_PER_REC;
    const uint32_t j = (uint32_t)(t % SGS_LANES_PER_REC);
    if (i >= n) return;
    sgs_spec sp;
    sgs_make_spec(seed, first + i, n_guides, L, mode, sp);
    const uint32_t rl = sgs_fastq_record_len(first + i, sp.len);
    if (j < rl) out[offsets[i] + j] = sgs_fastq_byte(sp, first + i, lib, L, j);
}

// ---- host ---------------------------------------------------------------------------------------
static uint32_t fill_read_host(const sgs_spec &sp, const uint8_t *lib, uint32_t L, uint8_t *out) {
    for (uint32_t j = 0; j < sp.len; j++) out[j] = sgs_byte(sp, lib, L, j);
    return sp.len;
}

extern "C" {

const char *sgs_last_error(void) { return g_serr.c_str(); }

int sgs_library(uint64_t seed, uint32_t n, uint32_t L, uint8_t *seqs_out) {
    if (!seqs_out || L == 0 || L > 32) return sfail("sgs_library: bad argument");
    const uint64_t mask = L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1);
    if (L < 16 && (uint64_t)n * 4 > (mask + 1)) return sfail("sgs_library: too many guides for this length");
    xoshiro rng(seed);
    std::unordered_set<uint64_t> seen;
    seen.reserve((size_t)n * 2);
    std::vector<uint64_t> keys(n);
    uint32_t n_pairs = n >= 400 ? 100 : n / 4;           // planted block: the last 2*n_pairs guides
    const uint32_t n_plain = n - 2 * n_pairs;
    for (uint32_t i = 0; i < n_plain; i++) {
        uint64_t k;
        do { k = rng.next() & mask; } while (!seen.insert(k).second);
        keys[i] = k;
    }
    for (uint32_t p = 0; p < n_pairs; p++) {
        uint64_t a, b;
        for (;;) {
            a = rng.next() & mask;
            const uint64_t r = rng.next();
            const uint32_t j1 = (uint32_t)(r % L), j2 = (uint32_t)((r >> 16) % L);
            b = a ^ ((1 + (r >> 32) % 3) << (2 * j1));                       // Hamming 1
            if (p & 1) {                                                       // Hamming 2
                if (j2 == j1) continue;
                b ^= (1 + (r >> 40) % 3) << (2 * j2);
            }
            if (seen.count(a) || seen.count(b) || a == b) continue;
            seen.insert(a); seen.insert(b);
            break;
        }
        keys[n_plain + 2 * p] = a;
        keys[n_plain + 2 * p + 1] = b;
    }
    for (uint32_t i = 0; i < n; i++) key_to_ascii(keys[i], L, seqs_out + (size_t)i * L);
    return 0;
}

size_t sgs_library_fasta(const uint8_t *seqs, uint32_t n, uint32_t L, uint8_t *out, size_t cap) {
    const size_t need = (size_t)n * (1 + 8 + 1 + L + 1);
    if (!out || cap < need) return need;
    uint8_t *p = out;
    for (uint32_t i = 0; i < n; i++) {
        char hdr[16];
        snprintf(hdr, sizeof(hdr), ">sg%06u\n", i % 1000000u);
        memcpy(p, hdr, 10); p += 10;
        memcpy(p, seqs + (size_t)i * L, L); p += L;
        *p++ = '\n';
    }
    return (size_t)(p - out);
}

uint32_t sgs_read_len(uint64_t seed, uint64_t i, uint32_t L, uint32_t mode) {
    sgs_spec sp;
    sgs_make_spec(seed, i, 0, L, mode, sp);
    return sp.len;
}

uint32_t sgs_read_class(uint64_t seed, uint64_t i, uint32_t n_guides, uint32_t L, uint32_t mode, uint32_t *gid_out) {
    sgs_spec sp;
    sgs_make_spec(seed, i, n_guides, L, mode, sp);
    if (gid_out) *gid_out = sp.gid;
    return sp.cls;
}

int sgs_reads_host(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs, uint32_t n_guides, uint32_t L,
                   uint32_t mode, uint8_t *seqs_out, uint64_t *offsets_out) {
    if (!lib_seqs || !seqs_out || !offsets_out || !n_guides) return sfail("sgs_reads_host: bad argument");
    uint64_t off = 0;
    for (uint64_t t = 0; t < n; t++) {
        sgs_spec sp;
        sgs_make_spec(seed, first + t, n_guides, L, mode, sp);
        offsets_out[t] = off;
        off += fill_read_host(sp, lib_seqs, L, seqs_out + off);
    }
    offsets_out[n] = off;
    return 0;
}

size_t sgs_fastq_host(uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs, uint32_t n_guides,
                      uint32_t L, uint32_t mode, uint8_t *out, size_t cap) {
    size_t need = 0;
    for (uint64_t t = 0; t < n; t++) {
        sgs_spec sp;
        sgs_make_spec(seed, first + t, n_guides, L, mode, sp);
        need += sgs_fastq_record_len(first + t, sp.len);
    }
    if (!out || cap < need || !lib_seqs) return need;
    uint8_t *p = out;
    for (uint64_t t = 0; t < n; t++) {
        sgs_spec sp;
        sgs_make_spec(seed, first + t, n_guides, L, mode, sp);
        p += snprintf((char *)p, 24, "@r%llu\n", (unsigned long long)(first + t));
        p += fill_read_host(sp, lib_seqs, L, p);
        *p++ = '\n'; *p++ = '+'; *p++ = '\n';
        memset(p, 'I', sp.len); p += sp.len;
        *p++ = '\n';
    }
    return (size_t)(p - out);
}

#define SGS_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return sfail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

int sgs_read_lens_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, uint32_t L, uint32_t mode,
                         uint32_t *lens_out) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_read_lens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, first, n,
                       L, mode, lens_out, 0);
    SGS_HIP(hipGetLastError());
    return 0;
}

int sgs_fastq_lens_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, uint32_t L, uint32_t mode,
                          uint32_t *rec_lens_out) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_read_lens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, first, n,
                       L, mode, rec_lens_out, 1);
    SGS_HIP(hipGetLastError());
    return 0;
}

int sgs_reads_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs,
                     uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *offsets, uint8_t *seqs_out) {
    // launches are split so that the grid stays below 2^31 blocks
    const uint64_t per = (1ull << 22);
    for (uint64_t done = 0; done < n; done += per) {
        const uint64_t m = n - done < per ? n - done : per;
        const uint64_t threads = m * SGS_LANES_PER_READ;
        hipLaunchKernelGGL(k_reads_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                           first + done, m, lib_seqs, n_guides, L, mode, offsets + done, seqs_out);
        SGS_HIP(hipGetLastError());
    }
    return 0;
}

int sgs_fastq_device(void *stream, uint64_t seed, uint64_t first, uint64_t n, const uint8_t *lib_seqs,
                     uint32_t n_guides, uint32_t L, uint32_t mode, const uint64_t *rec_offsets, uint8_t *text_out) {
    const uint64_t per = (1ull << 21);
    for (uint64_t done = 0; done < n; done += per) {
        const uint64_t m = n - done < per ? n - done : per;
        const uint64_t threads = m * SGS_LANES_PER_REC;
        hipLaunchKernelGGL(k_fastq_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                           first + done, m, lib_seqs, n_guides, L, mode, rec_offsets + done, text_out);
        SGS_HIP(hipGetLastError());
    }
    return 0;
}

}  // extern "C"
