// Does the PHYSICAL layout of a device buffer change what a plain streaming kernel gets out of HBM?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/contig_bw tools/ubench/contig_bw.hip && /tmp/contig_bw
// Buffers of 1 GiB: hipMalloc, hipExtMallocWithFlags(hipDeviceMallocContiguous), and a virtual range backed by 2 MiB (or
// granularity-sized) physical chunks created one by one and mapped in SHUFFLED order (hipMemCreate / hipMemMap).
// Kernels: fill (write only), sum (read only), copy src -> dst, and a strided-block writer that imitates k_partition
// (each workgroup appends ~0.5 KB runs to 64 open 4 KiB blocks of its own range).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) k_fill(u32x4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) dst[i] = u32x4{1, 2, 3, (uint32_t)i};
}
__global__ void __launch_bounds__(1024) k_fill_nt(u32x4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) __builtin_nontemporal_store(u32x4{1, 2, 3, (uint32_t)i}, &dst[i]);
}
__global__ void __launch_bounds__(1024) k_copy_nt(const u32x4 *src, u32x4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
__global__ void __launch_bounds__(1024) k_sum(const u32x4 *src, size_t n16, uint32_t *out) {
    uint32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) { u32x4 v = __builtin_nontemporal_load(&src[i]); s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) *out = s;
}
__global__ void __launch_bounds__(1024) k_copy(const u32x4 *src, u32x4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) dst[i] = __builtin_nontemporal_load(&src[i]);
}
// workgroup w owns bytes [w * per, (w + 1) * per) of dst; it walks its range in "rounds": in a round every one of 64 open 4 KiB
// blocks receives a 512-byte run (16 lanes x 16 B... 32 lanes x 16 B) at the current fill level; 8 rounds fill the 64 blocks
__global__ void __launch_bounds__(1024) k_blocks(u32x4 *dst, size_t per_wg_bytes, int rot) {
    const size_t base = (size_t)blockIdx.x * per_wg_bytes;
    const uint32_t t = threadIdx.x, blk = t >> 4, lane = t & 15;          // 64 blocks x 16 lanes x 16 B = 256 B per block per step
    for (size_t g = 0; g + 64 * 4096 <= per_wg_bytes; g += 64 * 4096)      // a group of 64 blocks
        for (uint32_t lvl = 0; lvl < 16; lvl++) {                          // 16 steps of 256 B fill a 4 KiB block
            const uint32_t r = rot ? ((uint32_t)((g >> 12) + blk) * 2654435761u >> 28) : 0u;
            const size_t off = base + g + (size_t)blk * 4096 + (((lvl + r) & 15u) << 8) + lane * 16;
            dst[off >> 4] = u32x4{t, lvl, 3, 4};
        }
}

static float time_it(hipStream_t st, int reps, const std::function<void()> &f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t N = 1ull << 30;
    hipStream_t st; CK(hipStreamCreate(&st));
    void *src; CK(hipMalloc(&src, N)); CK(hipMemset(src, 1, N));
    uint32_t *flag; CK(hipMalloc((void **)&flag, 4));
    struct Buf { const char *name; void *p; };
    std::vector<Buf> bufs;
    void *a; CK(hipMalloc(&a, N)); bufs.push_back({"hipMalloc", a});
    void *c = nullptr;
    if (hipExtMallocWithFlags(&c, N, hipDeviceMallocContiguous) == hipSuccess) bufs.push_back({"contiguous", c}); else { (void)hipGetLastError(); printf("no contiguous allocation\n"); }
    // VMM: chunks mapped in shuffled order
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) == hipSuccess && gran) {
        for (size_t chunk : {(size_t)(2u << 20), (size_t)(32u << 20)}) {
            if (chunk % gran) continue;
            const size_t n_chunks = N / chunk;
            void *va = nullptr;
            if (hipMemAddressReserve(&va, N, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            std::vector<hipMemGenericAllocationHandle_t> hs(n_chunks);
            bool ok = true;
            for (size_t i = 0; i < n_chunks && ok; i++) ok = hipMemCreate(&hs[i], chunk, &prop, 0) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); printf("hipMemCreate failed (chunk %zu)\n", chunk); continue; }
            std::vector<size_t> order(n_chunks);
            for (size_t i = 0; i < n_chunks; i++) order[i] = i;
            std::mt19937_64 rng(7); std::shuffle(order.begin(), order.end(), rng);
            for (size_t i = 0; i < n_chunks && ok; i++) ok = hipMemMap((char *)va + i * chunk, chunk, 0, hs[order[i]], 0) == hipSuccess;
            hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            ok = ok && hipMemSetAccess(va, N, &ad, 1) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); printf("hipMemMap failed (chunk %zu)\n", chunk); continue; }
            char *nm = (char *)malloc(64); snprintf(nm, 64, "vmm shuffled %zu KiB", chunk >> 10);
            bufs.push_back({nm, va});
        }
        printf("VMM granularity %zu\n", gran);
    }
    const size_t n16 = N / 16;
    for (auto &b : bufs) {
        const float f = time_it(st, 10, [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, st, (u32x4 *)b.p, n16); });
        const float fnt = time_it(st, 10, [&] { hipLaunchKernelGGL(k_fill_nt, dim3(2048), dim3(1024), 0, st, (u32x4 *)b.p, n16); });
        const float cnt = time_it(st, 10, [&] { hipLaunchKernelGGL(k_copy_nt, dim3(2048), dim3(1024), 0, st, (const u32x4 *)src, (u32x4 *)b.p, n16); });
        printf("%-26s fill with nontemporal stores %.2f TB/s  copy-into with nontemporal stores %.2f TB/s (r+w)\n", b.name, N / fnt / 1e9, 2.0 * N / cnt / 1e9);
        const float s = time_it(st, 10, [&] { hipLaunchKernelGGL(k_sum, dim3(2048), dim3(1024), 0, st, (const u32x4 *)b.p, n16, flag); });
        const float cp = time_it(st, 10, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(1024), 0, st, (const u32x4 *)src, (u32x4 *)b.p, n16); });
        const float cr = time_it(st, 10, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(1024), 0, st, (const u32x4 *)b.p, (u32x4 *)src, n16); });
        const float b0 = time_it(st, 10, [&] { hipLaunchKernelGGL(k_blocks, dim3(512), dim3(1024), 0, st, (u32x4 *)b.p, N / 512, 0); });
        const float b1 = time_it(st, 10, [&] { hipLaunchKernelGGL(k_blocks, dim3(512), dim3(1024), 0, st, (u32x4 *)b.p, N / 512, 1); });
        printf("%-26s fill %.2f TB/s  read %.2f TB/s  copy-into %.2f TB/s (r+w)  copy-from %.2f TB/s  block-writer %.2f TB/s  rotated %.2f TB/s\n", b.name,
               N / f / 1e9, N / s / 1e9, 2.0 * N / cp / 1e9, 2.0 * N / cr / 1e9, N / b0 / 1e9, N / b1 / 1e9);
    }
    return 0;
}
