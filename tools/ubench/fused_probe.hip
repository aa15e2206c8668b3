// Would ONE kernel that probes the whole library table in the XCD L2 and counts in per-CU byte counters in LDS beat
// k_partition + k_count_slices (0.52 ms per 100M reads)?  The experiment: 100M records, 86 % of them keys of a 100k-entry table of
// 2-slot buckets (2 MB), per record one 16-byte gather (+ a second for the few full buckets), a hit adds to one of 100k u8 counters
// in LDS (overflow -> global atomic), a miss is appended to a per-workgroup run.
//   hipcc --offload-arch=gfx950 -O3 -o fused_probe fused_probe.hip && ./fused_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NG 100000u
#define LOG2_SLOTS 18u
#define EMPTY 0xFFFFFFFFFFFFFFFFull
__host__ __device__ inline uint32_t bucket_of(uint64_t key) { return (uint32_t)(((key ^ (key >> 29)) * 0x9E3779B97F4A7C15ull) >> (65 - LOG2_SLOTS)); }

template <int R, int MODE>
__global__ void __launch_bounds__(1024) k_fused(const uint64_t *__restrict__ recs, uint64_t n, const ulonglong2 *__restrict__ table, uint32_t *__restrict__ counts,
                                                uint64_t *__restrict__ miss, uint32_t *__restrict__ miss_n, uint64_t per_wg) {
    __shared__ uint32_t cnt[(NG + 3) / 4];          // u8 counters, four to a word
    __shared__ uint32_t nm;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < (NG + 3) / 4; i += 1024) cnt[i] = 0;
    if (t == 0) nm = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_wg, hi = lo + per_wg < n ? lo + per_wg : n;
    uint64_t *mout = miss + lo;
    for (uint64_t base = lo; base < hi; base += 1024ull * R) {
        uint64_t r[R];
        ulonglong2 b[R];
#pragma unroll
        for (int k = 0; k < R; k++) { const uint64_t i = base + (uint64_t)k * 1024 + t; r[k] = i < hi ? __builtin_nontemporal_load(&recs[i]) : EMPTY; }
#pragma unroll
        for (int k = 0; k < R; k++) { if (MODE == 2) { b[k].x = (r[k] & 0xFFFFFFFFFFull) << 20 | (uint32_t)(r[k] % NG); b[k].y = EMPTY; if ((r[k] >> 3 & 7) == 0) b[k].x = 0; } else b[k] = table[bucket_of(r[k] & 0xFFFFFFFFFFull)]; }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint64_t key = r[k] & 0xFFFFFFFFFFull;
            if (r[k] == EMPTY) continue;
            uint32_t g = 0xFFFFFFFFu;
            if ((b[k].x >> 20) == key) g = (uint32_t)(b[k].x & 0xFFFFF);
            else if ((b[k].y >> 20) == key) g = (uint32_t)(b[k].y & 0xFFFFF);
            else if (b[k].y != EMPTY) {                      // full bucket: the next one (rare)
                const ulonglong2 c = table[(bucket_of(key) + 1) & ((1u << (LOG2_SLOTS - 1)) - 1)];
                if ((c.x >> 20) == key) g = (uint32_t)(c.x & 0xFFFFF); else if ((c.y >> 20) == key) g = (uint32_t)(c.y & 0xFFFFF);
            }
            if (g != 0xFFFFFFFFu) {
                if (MODE == 1) continue;
                if (MODE == 3) { atomicAdd(&cnt[g >> 2], 1u); continue; }
                const uint32_t sh = 8u * (g & 3u);
                const uint32_t old = atomicAdd(&cnt[g >> 2], 1u << sh);
                if (((old >> sh) & 0xFFu) == 0xFFu) {          // the byte wrapped and carried into its neighbour: take the carry back, bank 256
                    if (sh < 24u) atomicSub(&cnt[g >> 2], 1u << (sh + 8u));
                    atomicAdd(&counts[g], 256u);
                }
            } else {
                mout[atomicAdd(&nm, 1u)] = r[k];
            }
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < NG; i += 1024) { const uint32_t v = (cnt[i >> 2] >> (8u * (i & 3u))) & 0xFFu; if (v) atomicAdd(&counts[i], v); }
    if (t == 0) miss_n[blockIdx.x] = nm;
}

int main() {
    const uint64_t n = 100000000ull;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> keys(NG), slots(1u << LOG2_SLOTS, EMPTY);
    for (uint32_t g = 0; g < NG; g++) {
        keys[g] = rng() & 0xFFFFFFFFFFull;
        uint32_t b = bucket_of(keys[g]);
        for (;;) { if (slots[2 * b] == EMPTY) { slots[2 * b] = keys[g] << 20 | g; break; } if (slots[2 * b + 1] == EMPTY) { slots[2 * b + 1] = keys[g] << 20 | g; break; } b = (b + 1) & ((1u << (LOG2_SLOTS - 1)) - 1); }
    }
    std::vector<uint64_t> recs(n);
    for (uint64_t i = 0; i < n; i++) { const uint64_t x = rng(); recs[i] = (x % 100 < 86) ? keys[(x >> 8) % NG] : (x & 0xFFFFFFFFFFull); }
    uint64_t *d_recs, *d_miss; ulonglong2 *d_tab; uint32_t *d_counts, *d_mn;
    CK(hipMalloc(&d_recs, n * 8)); CK(hipMalloc(&d_miss, n * 8)); CK(hipMalloc(&d_tab, slots.size() * 8)); CK(hipMalloc(&d_counts, NG * 4)); CK(hipMalloc(&d_mn, 4096 * 4));
    CK(hipMemcpy(d_recs, recs.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, slots.data(), slots.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; mode++) for (int wgs : {256, 512}) {
        const uint64_t per = ((n + wgs - 1) / wgs + 8191) / 8192 * 8192;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(d_counts, 0, NG * 4));
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL((k_fused<8, 0>), dim3(wgs), dim3(1024), 0, 0, d_recs, n, d_tab, d_counts, d_miss, d_mn, per);
            if (mode == 1) hipLaunchKernelGGL((k_fused<8, 1>), dim3(wgs), dim3(1024), 0, 0, d_recs, n, d_tab, d_counts, d_miss, d_mn, per);
            if (mode == 2) hipLaunchKernelGGL((k_fused<8, 2>), dim3(wgs), dim3(1024), 0, 0, d_recs, n, d_tab, d_counts, d_miss, d_mn, per);
            if (mode == 3) hipLaunchKernelGGL((k_fused<8, 3>), dim3(wgs), dim3(1024), 0, 0, d_recs, n, d_tab, d_counts, d_miss, d_mn, per);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) {
                std::vector<uint32_t> c(NG); CK(hipMemcpy(c.data(), d_counts, NG * 4, hipMemcpyDeviceToHost));
                uint64_t tot = 0; for (uint32_t v : c) tot += v;
                std::vector<uint32_t> mn(wgs); CK(hipMemcpy(mn.data(), d_mn, wgs * 4, hipMemcpyDeviceToHost));
                uint64_t m = 0; for (uint32_t v : mn) m += v;
                printf("mode %d (0 full, 1 no counters, 2 no gather, 3 no-return atomics) %4d workgroups: %.3f ms  (hits %llu + misses %llu = %llu)\n", mode, wgs, ms, (unsigned long long)tot, (unsigned long long)m, (unsigned long long)(tot + m));
            }
        }
    }
    return 0;
}
