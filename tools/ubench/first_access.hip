// What does the FIRST memory access of a kernel wait for?  Every kernel of the pass sees its first load come back after ~12 us whatever it
// asks for (DESIGN.md §4 "Where a workgroup's time goes", round 3) — four launches per pass: 5-6 % of it.  Here: a producer kernel dirties
// `mb` MB with plain (or write-through / nontemporal) stores, then a consumer kernel (512 workgroups x 1024 lanes, like the pass's) stamps how long
// its first load takes, from a buffer the producer wrote, from one it did not touch, and from one nobody touched since the start.
//   hipcc --offload-arch=gfx950 -O3 -o first_access first_access.hip && ./first_access
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>      // 0 plain, 1 nontemporal, 2 write-through (sc1)
__global__ void __launch_bounds__(1024) k_dirty(uint4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) {
        const uint4 v = make_uint4((uint32_t)i, 1, 2, 3);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        if (MODE == 1) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(&dst[i]));
        else if (MODE == 2) { const u32x4 vv = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&dst[i]), "v"(vv) : "memory"); }
        else dst[i] = v;
    }
}

__global__ void __launch_bounds__(1024) k_first(const uint32_t *src, size_t stride_words, uint32_t *out_ticks, uint32_t *sink) {
    // lane 0 of every wave: one load, timed on the constant 100 MHz clock
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t v = 0;
    if ((threadIdx.x & 63) == 0) v = src[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * stride_words];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) out_ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = (uint32_t)(t1 - t0);
    // a second, dependent access right behind it (what a warm access costs)
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    uint32_t w = 0;
    if ((threadIdx.x & 63) == 0) w = src[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * stride_words + 1024 + (v & 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { out_ticks[8192 + blockIdx.x * 16 + (threadIdx.x >> 6)] = (uint32_t)(t3 - t2); if (w == 0x12345678u) sink[0] = w; }
}

static void report(const char *what, uint32_t *d_ticks) {
    std::vector<uint32_t> h(16384);
    CK(hipMemcpy(h.data(), d_ticks, 16384 * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> a(h.begin(), h.begin() + 8192), b(h.begin() + 8192, h.end());
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-64s first load: p10 %.2f p50 %.2f p90 %.2f us   second: p50 %.2f p90 %.2f us\n", what, a[819] / 100.0, a[4096] / 100.0, a[7372] / 100.0,
           b[4096] / 100.0, b[7372] / 100.0);
}

int main() {
    const size_t big = 1ull << 30;
    uint4 *pool; uint32_t *other, *cold, *ticks, *sink;
    CK(hipMalloc(&pool, big)); CK(hipMalloc(&other, big)); CK(hipMalloc(&cold, big)); CK(hipMalloc(&ticks, 16384 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(pool, 0, big)); CK(hipMemset(other, 0, big)); CK(hipMemset(cold, 0, big));
    CK(hipDeviceSynchronize());
    const size_t stride = big / 4 / 8192;          // one word per wave, spread over the whole GB
    for (int rep = 0; rep < 2; rep++) {
        // nothing before
        hipLaunchKernelGGL(k_first, dim3(512), dim3(1024), 0, 0, other, stride, ticks, sink);
        CK(hipDeviceSynchronize());
        report(rep ? "idle device, buffer read before" : "idle device, buffer never read", ticks);
    }
    for (int mode = 0; mode < 3; mode++)
        for (size_t mb : {0, 8, 64, 512}) {
            for (int which = 0; which < 2; which++) {
                if (mb) {
                    if (mode == 0) hipLaunchKernelGGL(k_dirty<0>, dim3(512), dim3(1024), 0, 0, pool, (mb << 20) / 16);
                    else if (mode == 1) hipLaunchKernelGGL(k_dirty<1>, dim3(512), dim3(1024), 0, 0, pool, (mb << 20) / 16);
                    else hipLaunchKernelGGL(k_dirty<2>, dim3(512), dim3(1024), 0, 0, pool, (mb << 20) / 16);
                } else if (mode) continue;
                hipLaunchKernelGGL(k_first, dim3(512), dim3(1024), 0, 0, which ? other : (const uint32_t *)pool, stride, ticks, sink);
                CK(hipDeviceSynchronize());
                char what[128];
                snprintf(what, sizeof what, "behind %4zu MB of %s stores, reads %s", mb, mode == 0 ? "plain" : mode == 1 ? "nontemporal" : "write-through",
                         which ? "ANOTHER buffer" : "the written buffer");
                report(what, ticks);
            }
        }
    // the same consumer twice in a row (what a launch boundary alone costs)
    hipLaunchKernelGGL(k_first, dim3(512), dim3(1024), 0, 0, other, stride, ticks, sink);
    hipLaunchKernelGGL(k_first, dim3(512), dim3(1024), 0, 0, other, stride, ticks, sink);
    CK(hipDeviceSynchronize());
    report("behind a kernel that only read", ticks);
    hipLaunchKernelGGL(k_first, dim3(512), dim3(1024), 0, 0, cold, stride, ticks, sink);
    CK(hipDeviceSynchronize());
    report("a buffer nobody read since its memset", ticks);
    return 0;
}
