// Microbenchmark: random 8-byte gathers from a table of varying size, with an optional coalesced
// stream read next to it (what the lookup kernel does).  Prints G gathers/s.
//   hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int R, bool STREAM, bool NT>
__global__ void __launch_bounds__(256) k_gather(const uint64_t *__restrict__ table, uint32_t log2n,
                                                const uint64_t *__restrict__ stream, uint64_t n,
                                                uint64_t *__restrict__ out) {
    uint64_t acc = 0;
    const uint64_t groups = n / R;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k[R], v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (STREAM) k[r] = NT ? __builtin_nontemporal_load(&stream[g * R + r]) : stream[g * R + r];
            else k[r] = mix(g * R + r);
        }
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = table[k[r] >> (64 - log2n)];
#pragma unroll
        for (int r = 0; r < R; r++) acc ^= v[r];
    }
    if (acc == 0x1234567) out[0] = acc;
}

__global__ void k_fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = mix(i + 77);
}

template <int R, bool STREAM, bool NT>
float run(const uint64_t *table, uint32_t log2n, const uint64_t *stream, uint64_t n, uint64_t *out, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_gather<R, STREAM, NT>), dim3(blocks), dim3(256), 0, 0, table, log2n, stream, n, out);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_gather<R, STREAM, NT>), dim3(blocks), dim3(256), 0, 0, table, log2n, stream, n, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const uint64_t n = 100000000ull;
    uint64_t *table, *stream, *out;
    hipMalloc(&table, 1ull << 28 << 3 >> 3);   // placeholder, realloc below
    hipFree(table);
    hipMalloc(&table, (1ull << 25) * 8);       // up to 256 MB
    hipMalloc(&stream, n * 8);
    hipMalloc(&out, 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, 1ull << 25);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, stream, n);
    hipDeviceSynchronize();
    printf("%-10s %-28s %10s %12s\n", "table", "mode", "ms", "Ggathers/s");
    for (uint32_t log2n : {15u, 17u, 18u, 19u, 20u, 22u, 24u, 25u}) {
        const double mb = (double)(8ull << log2n) / 1048576.0;
        for (int blocks : {2048, 8192}) {
            float t;
            char name[64];
            t = run<1, false, false>(table, log2n, stream, n, out, blocks);
            snprintf(name, 64, "R1 nostream b%d", blocks); printf("%8.2fMB %-28s %10.3f %12.1f\n", mb, name, t, n / t / 1e6);
            t = run<4, false, false>(table, log2n, stream, n, out, blocks);
            snprintf(name, 64, "R4 nostream b%d", blocks); printf("%8.2fMB %-28s %10.3f %12.1f\n", mb, name, t, n / t / 1e6);
            t = run<4, true, false>(table, log2n, stream, n, out, blocks);
            snprintf(name, 64, "R4 stream b%d", blocks); printf("%8.2fMB %-28s %10.3f %12.1f\n", mb, name, t, n / t / 1e6);
            t = run<4, true, true>(table, log2n, stream, n, out, blocks);
            snprintf(name, 64, "R4 stream-nt b%d", blocks); printf("%8.2fMB %-28s %10.3f %12.1f\n", mb, name, t, n / t / 1e6);
        }
    }
    return 0;
}
